// C ABI of the MI355X consensus-inference engine (include/medaka_amd.h).
// Host side: weight packing, workspace management, launch sequencing, hipEvent timing.
// No CPU fallback exists here: every compute entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/medaka_amd.h"
#include "common.hpp"
#include "exact.hpp"
#include "gi_proj.hpp"
#include "layout.hpp"
#include "head.hpp"
#include "rec_mfma.hpp"
#include "rec_fused.hpp"
#include "scan_split.hpp"

using namespace mdk;

#ifndef MDK_PF
#define MDK_PF 5   // gi / x prefetch ring depth (effective look-ahead PF-1 steps)
#endif


#include "host_common.hpp"

extern "C" const char *mdk_last_error(void) { return g_mdk_err.c_str(); }
extern "C" const char *mdk_version(void) { return "medaka_amd 0.1 (gfx950)"; }

// The GRU engine proper, in four parts of this translation unit (each builds on the one before):
#include "gru_model.hpp"     // model object, contexts, create / destroy / options
#include "gru_pass.hpp"      // PassPlan + Pass: one pass of the network over a batch
#include "gru_split.hpp"     // split scan: plan, enqueue / finish, run_forward, start_call
#include "gru_entries.hpp"   // mdk_gru_forward_dev / _stage_input / _forward_pipelined / _forward / counts, decoded

// ------------------------------------------------------------------------------------------
// majority-vote model
extern "C" int mdk_majority_forward_dev(const float *x_dev, long n_cols, float *probs_dev, int device,
                                        void *stream) {
    if (n_cols < 0) return fail(MDK_ERR_ARG, "negative n_cols");
    if (n_cols == 0) return MDK_OK;
    if (!x_dev || !probs_dev) return fail(MDK_ERR_ARG, "null buffer");
    HIP_TRY(hipSetDevice(device));
    hipLaunchKernelGGL(k_majority, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x_dev, probs_dev, n_cols);
    HIP_TRY(hipGetLastError());
    return MDK_OK;
}

// (the host entry's device buffers are kept per device, grow-only: a hipMalloc / hipFree pair per call costs milliseconds, and
// freed device memory is wiped by the driver on the DMA engines -- behind which any engine's strided result copies wait)
namespace {
struct MajorityBuffers { std::mutex mu; float *x = nullptr, *p = nullptr; size_t cols = 0; };
MajorityBuffers g_majority[16];
}

extern "C" int mdk_majority_forward(const float *x_host, long n_cols, float *probs_host, int device) {
    if (n_cols < 0) return fail(MDK_ERR_ARG, "negative n_cols");
    if (n_cols == 0) return MDK_OK;
    if (!x_host || !probs_host) return fail(MDK_ERR_ARG, "null buffer");
    if (device < 0 || device >= 16) return fail(MDK_ERR_ARG, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));
    MajorityBuffers &b = g_majority[device];
    std::lock_guard<std::mutex> lock(b.mu);
    if ((size_t)n_cols > b.cols) {
        free_dev(b.x); free_dev(b.p); b.x = b.p = nullptr; b.cols = 0;
        HIP_TRY(hipMalloc((void **)&b.x, (size_t)n_cols * 10 * sizeof(float)));
        hipError_t e = hipMalloc((void **)&b.p, (size_t)n_cols * 5 * sizeof(float));
        if (e != hipSuccess) { free_dev(b.x); b.x = nullptr; return fail(MDK_ERR_OOM, "hipMalloc failed: %s", hipGetErrorString(e)); }
        b.cols = (size_t)n_cols;
    }
    if (hipMemcpy(b.x, x_host, (size_t)n_cols * 10 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return fail(MDK_ERR_DEVICE, "H2D copy failed");
    int rc = mdk_majority_forward_dev(b.x, n_cols, b.p, device, nullptr);
    if (!rc && hipMemcpy(probs_host, b.p, (size_t)n_cols * 5 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(MDK_ERR_DEVICE, "D2H copy failed");
    return rc;
}

// ------------------------------------------------------------------------------------------
// raw device helpers
extern "C" int mdk_device_count(int *count) {
    if (!count) return fail(MDK_ERR_ARG, "null argument");
    *count = 0;
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) { *count = 0; return fail(MDK_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    return MDK_OK;
}
extern "C" int mdk_device_name(int device, char *buf, size_t buflen) {
    if (!buf || buflen == 0) return fail(MDK_ERR_ARG, "null argument");
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return MDK_OK;
}
extern "C" int mdk_dev_alloc(int device, size_t bytes, void **ptr) {
    if (!ptr) return fail(MDK_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMalloc(ptr, bytes));
    return MDK_OK;
}
extern "C" int mdk_dev_free(int device, void *ptr) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(ptr));
    return MDK_OK;
}
// page-locked host memory: buffers handed to mdk_gru_forward / mdk_rl_forward from here are copied by
// DMA without a staging pass and are never page-faulted in by the copy (a fresh 40 MB malloc costs 3.6 ms
// of first-touch faults as a copy target: profiles/r2_host_path_probe.txt)
extern "C" int mdk_host_alloc(size_t bytes, void **ptr) {
    if (!ptr) return fail(MDK_ERR_ARG, "null argument");
    HIP_TRY(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return MDK_OK;
}
extern "C" int mdk_host_free(void *ptr) {
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return MDK_OK;
}
// Batch assembly (reference Batch.collate -> torch.stack, torch_ext.py:147-148) with a few host threads.
extern "C" int mdk_gather_rows(void *dst, const void *const *rows, int n_rows, size_t row_bytes, int n_threads) {
    if (n_rows < 0) return fail(MDK_ERR_ARG, "negative n_rows");
    if (n_rows == 0 || row_bytes == 0) return MDK_OK;
    if (!dst || !rows) return fail(MDK_ERR_ARG, "null buffer");
    for (int i = 0; i < n_rows; ++i)
        if (!rows[i]) return fail(MDK_ERR_ARG, "row %d is null", i);
    n_threads = std::max(1, std::min(std::min(n_threads, 64), n_rows));
    auto work = [=](int k) {
        const int lo = (int)((long)n_rows * k / n_threads), hi = (int)((long)n_rows * (k + 1) / n_threads);
        for (int i = lo; i < hi; ++i) memcpy(static_cast<char *>(dst) + (size_t)i * row_bytes, rows[i], row_bytes);
    };
    if (n_threads == 1) { work(0); return MDK_OK; }
    std::vector<std::thread> pool;
    try {
        for (int k = 1; k < n_threads; ++k) pool.emplace_back(work, k);
    } catch (...) {            // thread creation failed: finish what was not handed out on this thread
        const int started = (int)pool.size();
        for (int k = started + 1; k < n_threads; ++k) work(k);
    }
    work(0);
    for (auto &t : pool) t.join();
    return MDK_OK;
}
extern "C" int mdk_memcpy_h2d(int device, void *dst_dev, const void *src_host, size_t bytes) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    return MDK_OK;
}
extern "C" int mdk_memcpy_d2h(int device, void *dst_host, const void *src_dev, size_t bytes) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return MDK_OK;
}
extern "C" int mdk_device_synchronize(int device) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    return MDK_OK;
}

#ifdef MDK_DEBUG_HOOKS
// ------------------------------------------------------------------------------------------
// Test hook: keep `blocks` CUs busy with a compute-bound loop (profiles/soak_wide.py uses it as the competing
// tenant of the LSTM(384) cluster recurrence).  Synchronous.
__global__ __launch_bounds__(512, 1) void k_burn(float *out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; ++i) { a = fmaf(a, b, c); c = fmaf(c, b, a); }
    if (a + c == 12345.678f) out[0] = a;
}
extern "C" int mdk_selftest_burn(int device, int blocks, int iters) {
    if (blocks < 1 || iters < 0) return fail(MDK_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(device));
    float *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 4));
    hipLaunchKernelGGL(k_burn, dim3(blocks), dim3(512), 0, nullptr, d, iters);
    hipError_t e = hipDeviceSynchronize();
    (void)hipFree(d);
    if (e != hipSuccess) return fail(MDK_ERR_DEVICE, "burn kernel failed: %s", hipGetErrorString(e));
    return MDK_OK;
}

// Test hook: hold `blocks` CUs EXCLUSIVELY (one 512-thread work-group each with `lds_bytes` of LDS, e.g. 140 KB, so
// that nothing else fits beside it) for `milliseconds` of wall clock.  Synchronous; tests/test_parity_gpu.py uses it from
// a second thread as the tenant that leaves the LSTM(384) cluster recurrence fewer CUs than it needs.
__global__ __launch_bounds__(512, 1) void k_hold(unsigned long long ticks) {
    extern __shared__ unsigned char hold_lds[];
    hold_lds[threadIdx.x] = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
extern "C" int mdk_selftest_hold(int device, int blocks, int milliseconds, int lds_bytes) {
    if (blocks < 1 || milliseconds < 0 || milliseconds > 10000 || lds_bytes < 512 || lds_bytes > 160 * 1024)
        return fail(MDK_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hold), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipLaunchKernelGGL(k_hold, dim3(blocks), dim3(512), (size_t)lds_bytes, st, (unsigned long long)milliseconds * 100000ull);   // 100 MHz
    hipError_t e = hipStreamSynchronize(st);
    (void)hipStreamDestroy(st);
    if (e != hipSuccess) return fail(MDK_ERR_DEVICE, "hold kernel failed: %s", hipGetErrorString(e));
    return MDK_OK;
}
#endif   // MDK_DEBUG_HOOKS


// ------------------------------------------------------------------------------------------
// MFMA self-test: D = A(16x32) B(32x16) with the fragment layout the kernels assume, on
// asymmetric integer data (exact in fp16/fp32), plus an fp16-subnormal operand probe.
__global__ void k_selftest(const _Float16 *A /*[16][32]*/, const _Float16 *Bm /*[32][16]*/,
                           float *Dm /*[16][16]*/) {
    const int lane = threadIdx.x;
    half8 a, b;
    for (int i = 0; i < 8; ++i) {
        const int k = (lane >> 4) * 8 + i;
        a[i] = A[(lane & 15) * 32 + k];
        b[i] = Bm[k * 16 + (lane & 15)];
    }
    floatx4 c = {0.f, 0.f, 0.f, 0.f};
    c = mfma16(a, b, c);
    for (int r = 0; r < 4; ++r) Dm[((lane >> 4) * 4 + r) * 16 + (lane & 15)] = c[r];
}

extern "C" int mdk_selftest_mfma(int device, float *max_abs_err, int *subnormal_preserved) {
    if (!max_abs_err || !subnormal_preserved) return fail(MDK_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(device));
    std::vector<_Float16> A(16 * 32), Bm(32 * 16);
    std::vector<float> ref(256, 0.f), got(256);
    for (int pass = 0; pass < 2; ++pass) {
        for (int m_ = 0; m_ < 16; ++m_)
            for (int k = 0; k < 32; ++k) A[m_ * 32 + k] = (_Float16)(float)((m_ * 7 + k * 3) % 11 - 5);
        for (int k = 0; k < 32; ++k)
            for (int n = 0; n < 16; ++n) Bm[k * 16 + n] = (_Float16)(float)((k * 5 + n * 2 + k * n) % 13 - 6);
        if (pass == 1) {
            // subnormal probe: A[0][0] = 2^-20 (fp16 subnormal), B[0][0] = 1024, rest of row/col 0 zero
            for (int k = 0; k < 32; ++k) { A[k] = (_Float16)0.f; Bm[k * 16] = (_Float16)0.f; }
            A[0] = (_Float16)9.5367431640625e-07f;
            Bm[0] = (_Float16)1024.f;
        }
        for (int m_ = 0; m_ < 16; ++m_)
            for (int n = 0; n < 16; ++n) {
                float acc = 0.f;
                for (int k = 0; k < 32; ++k) acc += (float)A[m_ * 32 + k] * (float)Bm[k * 16 + n];
                ref[m_ * 16 + n] = acc;
            }
        _Float16 *dA = nullptr, *dB = nullptr;
        float *dD = nullptr;
        HIP_TRY(hipMalloc((void **)&dA, A.size() * 2));
        HIP_TRY(hipMalloc((void **)&dB, Bm.size() * 2));
        HIP_TRY(hipMalloc((void **)&dD, 256 * 4));
        HIP_TRY(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dB, Bm.data(), Bm.size() * 2, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_selftest, dim3(1), dim3(64), 0, nullptr, dA, dB, dD);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(got.data(), dD, 256 * 4, hipMemcpyDeviceToHost));
        (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dD);
        if (pass == 0) {
            float e = 0.f;
            for (int i = 0; i < 256; ++i) e = std::max(e, std::fabs(got[i] - ref[i]));
            *max_abs_err = e;
        } else {
            *subnormal_preserved = (std::fabs(got[0] - ref[0]) <= 1e-6f * std::fabs(ref[0])) ? 1 : 0;
        }
    }
    return MDK_OK;
}
