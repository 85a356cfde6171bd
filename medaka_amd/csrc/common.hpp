// Shared device helpers for the medaka_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kH = 128;        // GRU hidden size per direction (gru.py:16, every bundled model)
constexpr int kG = 3 * kH;     // gate rows r,z,n
constexpr int kWave = 64;      // CDNA wavefront

// Operand pre-scaling of the fp16 hi+lo split (DESIGN.md "fp16x2 split"): activations are
// multiplied by 2^10 before the split so that the low halves stay in fp16's normal range;
// weights by a per-matrix power of two chosen at load time.
constexpr float kActScale = 1024.0f;

// D = A(16x32) * B(32x16) + C on the matrix core, fp16 operands, fp32 accumulate.
// Lane l holds A[row = l&15][k-slot (l>>4)*8 + i], B[k-slot (l>>4)*8 + i][col = l&15],
// D[row = 4*(l>>4) + r][col = l&15].  Which physical k a (lane-group, i) slot stands for is
// free as long as A and B agree -- the kernels exploit that (see rec_mfma.hpp).
__device__ __forceinline__ floatx4 mfma16(half8 a, half8 b, floatx4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// exp2/rcp based logistic and tanh: ~2 ulp, error << 1e-6 absolute on (0,1)/(-1,1)
__device__ __forceinline__ float fast_exp(float x) {
    return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
}
__device__ __forceinline__ float sigmoid_f(float x) {
    return __builtin_amdgcn_rcpf(1.0f + fast_exp(-x));
}
__device__ __forceinline__ float tanh_f(float x) {
    // 1 - 2/(e^{2x}+1): saturates correctly for |x| large (exp -> inf / 0)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + fast_exp(2.0f * x));
}

// split x (already multiplied by the operand scale) into fp16 hi + fp16 lo, hi+lo ~ x to 2^-22
__device__ __forceinline__ void split_f16(float xs, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)xs;
    lo = (_Float16)(xs - (float)hi);
}

// N consecutive floats at a 4-byte aligned address as ONE memory instruction (global_load/store_dwordx3/x4
// only need dword alignment on gfx950).  The vector type is declared with 4-byte alignment so that the
// compiler neither assumes more nor splits the access.
template <int N>
struct FloatRun {
    typedef float vec_t __attribute__((ext_vector_type(N)));
    typedef vec_t unaligned_t __attribute__((aligned(4)));
};
template <int N>
__device__ __forceinline__ typename FloatRun<N>::vec_t load_run(const float *p) {
    return *reinterpret_cast<const typename FloatRun<N>::unaligned_t *>(p);
}
template <int N>
__device__ __forceinline__ void store_run(float *p, typename FloatRun<N>::vec_t v) {
    *reinterpret_cast<typename FloatRun<N>::unaligned_t *>(p) = v;
}

// Buffer addressing: a wave-uniform base in four SGPRs + ONE 32-bit lane offset + a scalar (or immediate) offset per access.
// Where a kernel's addresses differ by compile-time or wave-uniform amounts -- a scan's per-step blocks -- the 64-bit
// per-lane pointer form costs vector instructions per access (64-bit adds, selects) in an in-order instruction stream whose
// every cycle between two MFMAs is on the recurrence's critical path; in this form the stepping is scalar arithmetic.
// The window is 2 GB from the base (num_records, raw buffer; out-of-range reads return 0, writes are dropped); word 3 is
// gfx9's 32-bit-data descriptor.
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ floatx4 buf_load_floatx4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float buf_load_float(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
// aux: 0 default cache policy, 2 = nt (non-temporal: streaming data nobody re-reads soon)
template <int AUX = 0>
__device__ __forceinline__ void buf_store_float(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, AUX);
}

// LDS-only workgroup barrier: waits for this wave's LDS traffic, NOT for global loads/stores
// in flight (the gi prefetch ring and the h stores must stay in flight across steps).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

}  // namespace mdk
