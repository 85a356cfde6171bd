// Host-side helpers shared by the C-ABI translation units (api.hip, rl_api.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/medaka_amd.h"

inline thread_local std::string g_mdk_err;

inline int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_mdk_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            int code_ = (e_ == hipErrorOutOfMemory) ? MDK_ERR_OOM : MDK_ERR_DEVICE;            \
            return fail(code_, "%s failed: %s (%s:%d)%s", #expr, hipGetErrorString(e_),        \
                        __FILE__, __LINE__,                                                    \
                        code_ == MDK_ERR_OOM ? " -- lower the batch size (-b)" : "");          \
        }                                                                                      \
    } while (0)

inline void free_dev(void *p) { if (p) (void)hipFree(p); }

// power-of-two scale s.t. max|w| * scale <= 2^14 (fp16 max 65504), clamped
inline float pick_scale_max(float mx) {
    if (!(mx > 0.f) || !std::isfinite(mx)) return 1.0f;
    int e = 0;
    (void)std::frexp(mx, &e);         // mx = f * 2^e, f in [0.5, 1)
    int sh = 14 - e;                  // mx * 2^sh in [2^13, 2^14)
    sh = std::max(-10, std::min(14, sh));
    return std::ldexp(1.0f, sh);
}
inline float pick_scale(const float *w, size_t n) {
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) mx = std::max(mx, std::fabs(w[i]));
    return pick_scale_max(mx);
}

inline void split_host(float v, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

template <typename T>
inline int upload(T **dst, const std::vector<T> &src) {
    HIP_TRY(hipMalloc((void **)dst, src.size() * sizeof(T)));
    HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return MDK_OK;
}
