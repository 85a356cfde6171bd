// MDK_VARIANT_EXACT: plain fp32 FMA kernels for the same three stages.  No matrix core, no
// precision tricks, weights re-read from L2 every step -- slow on purpose, bit-simple, used by
// the gpu tests to cross-check the MFMA kernels on the device itself and as the generic path
// for shapes the MFMA kernels do not cover.  Same math as the oracle (oracle/gru_oracle.c).
#pragma once
#include "common.hpp"

namespace mdk {

// gi[d][m][n] = sum_k x[m][k] * w_ih_t[d][k][n] + bias[d][n];  one thread per (m, n)
static __global__ __launch_bounds__(128) void k_gi_exact(
    const float *__restrict__ x, const float *__restrict__ w_ih_t, const float *__restrict__ bias,
    float *__restrict__ gi, long M, int K, size_t gi_dir_stride, const float *__restrict__ out_scale_p)
{
    const int d = blockIdx.y;
    const long m = blockIdx.x / 3;
    const int n = (blockIdx.x % 3) * 128 + threadIdx.x;
    const float *xr = x + m * K;
    const float *wt = w_ih_t + (size_t)d * K * kG + n;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(xr[k], wt[(size_t)k * kG], acc);
    gi[(size_t)d * gi_dir_stride + (size_t)m * kG + n] = (acc + bias[(size_t)d * kG + n]) * out_scale_p[d];
}

// one 128-thread work-group per (window, direction); thread j owns hidden unit j
static __global__ __launch_bounds__(128) void k_rec_exact(
    const float *__restrict__ gi,      // [D][M][384], folded bias
    const float *__restrict__ w_hh_t,  // [D][128][384]
    const float *__restrict__ b_hn,    // [D][128]
    float *__restrict__ out, int B, int T, int out_stride, size_t gi_dir_stride, int reverse_mask)
{
    __shared__ float hs[2][kH];
    const int j = threadIdx.x;
    const int d = blockIdx.y;
    const int seq = blockIdx.x;
    const bool reverse = (reverse_mask >> d) & 1;
    const float *wt = w_hh_t + (size_t)d * kH * kG;
    const float *gi_d = gi + (size_t)d * gi_dir_stride + (size_t)seq * T * kG;
    const float bn = b_hn[d * kH + j];
    hs[0][j] = 0.f;
    float hprev = 0.f;
    __syncthreads();
    for (int step = 0; step < T; ++step) {
        const int t = reverse ? T - 1 - step : step;
        const float *hc = hs[step & 1];
        float ar = 0.f, az = 0.f, an = 0.f;
        for (int k = 0; k < kH; ++k) {
            const float hk = hc[k];
            ar = fmaf(wt[(size_t)k * kG + j], hk, ar);
            az = fmaf(wt[(size_t)k * kG + kH + j], hk, az);
            an = fmaf(wt[(size_t)k * kG + 2 * kH + j], hk, an);
        }
        const float *gr = gi_d + (size_t)t * kG;
        const float r = 1.0f / (1.0f + expf(-(gr[j] + ar)));
        const float z = 1.0f / (1.0f + expf(-(gr[kH + j] + az)));
        const float n = tanhf(gr[2 * kH + j] + r * (an + bn));
        const float h = (1.0f - z) * n + z * hprev;
        hprev = h;
        hs[(step + 1) & 1][j] = h;
        out[((size_t)seq * T + t) * out_stride + d * kH + j] = h;
        __syncthreads();
    }
}

}  // namespace mdk
