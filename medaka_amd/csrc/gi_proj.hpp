// Input projections  gi = x W_ih^T + b   for every column and both directions
// (the time-parallel half of nn.GRU, called from reference medaka/architectures/gru.py:66).
//
//  * k_gi_small : layer 0, K = num_features (10).  Exact fp32 FMA on the VALU: 10 FMAs per
//                 output make this a pure HBM-write kernel (1.5 KB out per 40 B in).
//  * k_gi_gemm  : layers >= 1, K = 256 (or 128).  49 % of the network's FLOPs.  fp16x2-split
//                 MFMA GEMM (three products hi*hi + lo*hi + hi*lo into one fp32 accumulator,
//                 operands pre-scaled by powers of two), fp32 in / fp32 out.
// The folded bias is b_ih + b_hh for the r and z gates and b_ih for n (b_hn must stay inside
// the r * (.) product, see rec_mfma.hpp).
#pragma once
#include "common.hpp"

namespace mdk {

// ------------------------------------------------------------------------------------------
// layer 0: each thread owns 4 consecutive gate columns for a strip of rows; its 4 x K weights
// stay in registers.
template <int KMAX>
__global__ __launch_bounds__(192) void k_gi_small(
    const float *__restrict__ x,      // [M][K]
    const float *__restrict__ w_ih_t, // [D][K][384]  (transposed at load time)
    const float *__restrict__ bias,   // [D][384]     folded bias
    float *__restrict__ gi,           // [D][M][384]
    long M, int K, size_t gi_dir_stride, int rows_per_block)
{
    const int d = blockIdx.y;
    const int c4 = threadIdx.x % 96;   // float4 column
    const int rsel = threadIdx.x / 96; // 0/1: even / odd rows of the strip
    float4 wreg[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        wreg[k] = (k < K)
            ? *reinterpret_cast<const float4 *>(w_ih_t + ((size_t)d * K + k) * kG + 4 * c4)
            : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float4 b = *reinterpret_cast<const float4 *>(bias + (size_t)d * kG + 4 * c4);
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    float *gout = gi + (size_t)d * gi_dir_stride;
    for (long r = r0 + rsel; r < r1; r += 2) {
        const float *xr = x + r * K;
        float4 acc = b;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k < K) {
                const float xv = xr[k];
                acc.x = fmaf(xv, wreg[k].x, acc.x);
                acc.y = fmaf(xv, wreg[k].y, acc.y);
                acc.z = fmaf(xv, wreg[k].z, acc.z);
                acc.w = fmaf(xv, wreg[k].w, acc.w);
            }
        }
        *reinterpret_cast<float4 *>(gout + (size_t)r * kG + 4 * c4) = acc;
    }
}

// ------------------------------------------------------------------------------------------
// layers >= 1.  Work-group = 4 waves, tile = 128 rows x 384 columns (one direction at a time,
// both directions from the same LDS-resident x tile).  The x tile is converted once to fp16
// hi/lo A-fragments in LDS (128 KB at K = 256); W_ih B-fragments stream from L2 (pre-packed so
// that every lane issues one 16-byte load per fragment).  Per (direction, column-half) pass wave
// w owns 48 columns (gemm_col()).
constexpr int kGemmRows = 128;
// gate column owned by (column half, wave, column tile, lane&15)
__host__ __device__ inline int gemm_col(int nhalf, int w, int nt, int n) {
    return nhalf * 192 + 48 * w + 16 * nt + n;
}

template <int KSTEPS>   // K = 32 * KSTEPS
__global__ __launch_bounds__(256, 1) void k_gi_gemm(
    const float *__restrict__ x,       // [M][K] fp32 (|x| < 1: GRU outputs)
    const half8 *__restrict__ wfrag,   // [D][2 halves][4 waves][KSTEPS][3 tiles][2 hi/lo][64 lanes]
    const float *__restrict__ bias,    // [D][384]
    float *__restrict__ gi,            // [D][M][384]
    long M, int D, size_t gi_dir_stride, const float *__restrict__ inv_scale_p)
{
    constexpr int K = 32 * KSTEPS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half8 *xs = reinterpret_cast<half8 *>(smem);   // [split 2][mt 8][KSTEPS][64 lanes]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long m0 = (long)blockIdx.x * kGemmRows;

    // ---- stage x tile: wave-item = 8 rows x 8 k-octets; lane = (k-octet hi3, row lo3)
    {
        const int r_lo = lane & 7, k_lo = lane >> 3;
        constexpr int KB = K / 64;               // k-octet blocks per row
        constexpr int ITEMS = 16 * KB;           // 16 row blocks
        for (int it = w; it < ITEMS; it += 4) {
            const int rb = it / KB, kb = it % KB;
            const int r = rb * 8 + r_lo;         // 0..127
            const int k8 = kb * 8 + k_lo;        // k-octet 0..K/8-1
            const long m = m0 + r;
            float v[8];
            if (m < M) {
                const float4 v0 = *reinterpret_cast<const float4 *>(x + m * K + k8 * 8);
                const float4 v1 = *reinterpret_cast<const float4 *>(x + m * K + k8 * 8 + 4);
                v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w;
                v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = 0.f;
            }
            half8 hi, lo;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                _Float16 a, b;
                split_f16(v[i] * kActScale, a, b);
                hi[i] = a; lo[i] = b;
            }
            const int mt = r >> 4, ks = k8 >> 2;
            const int slot = (k8 & 3) * 16 + (r & 15);   // A-fragment lane that consumes it
            xs[((0 * 8 + mt) * KSTEPS + ks) * 64 + slot] = hi;
            xs[((1 * 8 + mt) * KSTEPS + ks) * 64 + slot] = lo;
        }
    }
    __syncthreads();

    // 2*D passes: (direction, column half); wave w owns 48 columns = 3 MFMA column tiles per pass
    for (int pass = 0; pass < 2 * D; ++pass) {
        const int d = pass >> 1, nhalf = pass & 1;
        floatx4 acc[8][3];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) acc[mt][nt] = floatx4{0.f, 0.f, 0.f, 0.f};

        const half8 *wp = wfrag + ((size_t)(pass * 4 + w) * KSTEPS) * 6 * 64 + lane;
#pragma unroll 1
        for (int ks = 0; ks < KSTEPS; ++ks) {
            half8 bh[3], bl[3];
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                bh[nt] = wp[(size_t)((ks * 3 + nt) * 2 + 0) * 64];
                bl[nt] = wp[(size_t)((ks * 3 + nt) * 2 + 1) * 64];
            }
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                const half8 ah = xs[((0 * 8 + mt) * KSTEPS + ks) * 64 + lane];
                const half8 al = xs[((1 * 8 + mt) * KSTEPS + ks) * 64 + lane];
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) {
                    acc[mt][nt] = mfma16(ah, bh[nt], acc[mt][nt]);
                    acc[mt][nt] = mfma16(al, bh[nt], acc[mt][nt]);
                    acc[mt][nt] = mfma16(ah, bl[nt], acc[mt][nt]);
                }
            }
        }

        // ---- epilogue: scale back, add folded bias, store fp32
        const float inv_scale = inv_scale_p[d];
        float *gout = gi + (size_t)d * gi_dir_stride;
        const int col0 = gemm_col(nhalf, w, 0, lane & 15);
        const int rg = (lane >> 4) * 4;
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int col = col0 + 16 * nt;
            const float b = bias[(size_t)d * kG + col];
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long m = m0 + mt * 16 + rg + r;
                    if (m < M) gout[(size_t)m * kG + col] = fmaf(acc[mt][nt][r], inv_scale, b);
                }
            }
        }
    }
}

}  // namespace mdk
