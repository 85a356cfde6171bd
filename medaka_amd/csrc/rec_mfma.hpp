// GRU recurrence on the matrix cores -- the dominant kernel of the consensus forward pass.
//
// Computes, for one layer and one direction, the T dependent steps of
//     gh = W_hh h + b_hh ;  r = s(gi_r + gh_r) ; z = s(gi_z + gh_z) ;
//     n = tanh(gi_n + r * gh_n) ; h = (1 - z) n + z h          (PyTorch nn.GRU cell, called from
// reference medaka/architectures/gru.py:66) for a tile of 8 windows per work-group.
//
// MI355X mapping (DESIGN.md section "recurrence kernel"):
//   * one 256-thread work-group (4 waves, one per SIMD) per (8-window tile, direction);
//     wave w owns hidden units [32w, 32w+32) of all three gates = 6 MFMA column tiles;
//   * W_hh lives in registers for the whole kernel as pre-packed fp16 hi/lo B-fragments
//     (192 VGPR/AGPR per lane); h_t is staged in LDS as the fp16 hi/lo A-operand (4.25 KB,
//     double buffered) -- zero global-memory round trips on the step-to-step dependency;
//   * fp32 parity through an fp16x2 split: A rows = (window, hi|lo) -> 16 rows for 8 windows,
//     B = W_hi then W_lo into the same fp32 accumulator, so  acc[row hi] + acc[row lo]
//     = (h_hi + h_lo)(W_hi + W_lo) = h W to ~2^-22 relative, fp32 accumulate;
//   * sigmoid/tanh, the z-blend, the fp16 re-split and the store of h_t are fused behind the
//     MFMAs; gi (input projection, bias folded) is prefetched PF steps ahead into registers.
#pragma once
#include "common.hpp"

namespace mdk {

constexpr int kRecSeqs = 8;                 // windows per work-group
constexpr int kHGroupStride = 272;          // bytes: 16 rows x 16 B + 16 B pad (bank spread)
constexpr int kHKStride = 4 * kHGroupStride;  // one k-step (32 units) of the A image
constexpr int kHBufBytes = 4 * kHKStride;     // 4352 B per buffer

// Element i of lane-group gq in k-step ks stands for hidden unit (see pack_whh_frags()):
__host__ __device__ inline int rec_unit_of_slot(int ks, int gq, int i) {
    return 32 * ks + 16 * (i & 1) + 4 * gq + (i >> 1);
}

template <int PF>
__global__ __launch_bounds__(256, 1) void k_rec_mfma(
    const float *__restrict__ gi,      // [D][M][384] fp32, b_ih (+ b_hh for r,z) folded in
    const half8 *__restrict__ wfrag,   // [D][4 waves][6 tiles][4 ksteps][2 hi/lo][64 lanes]
    const float *__restrict__ b_hn,    // [D][128]
    float *__restrict__ out,           // [M][out_stride]; this direction at column d*128
    int B, int T, int out_stride, size_t gi_dir_stride, const float *__restrict__ inv_scale_p,
    int reverse_mask)
{
    __shared__ __attribute__((aligned(16))) unsigned char hbuf[2 * kHBufBytes];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d = blockIdx.y;
    const int c = lane & 15;   // MFMA column = unit within tile
    const int g = lane >> 4;   // MFMA row group: rows 4g..4g+3 = windows 2g, 2g+1 (hi, lo)
    const bool reverse = (reverse_mask >> d) & 1;
    const float inv_scale = inv_scale_p[d];

    // ---- recurrent weights -> registers (once)
    half8 wf[6][4][2];
    {
        const half8 *wp = wfrag + ((size_t)(d * 4 + w) * 48) * 64 + lane;
#pragma unroll
        for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp)
                    wf[t6][ks][sp] = wp[(size_t)((t6 * 4 + ks) * 2 + sp) * 64];
    }

    // ---- h_0 = 0 in both LDS buffers
    for (int i = tid; i < 2 * kHBufBytes / 4; i += 256) reinterpret_cast<uint32_t *>(hbuf)[i] = 0u;

    // ---- per-lane bookkeeping: 2 windows (q) x 2 sub-tiles (s) = 4 hidden values per lane
    const int seq_base = blockIdx.x * kRecSeqs + 2 * g;
    int u[2];
    float bhn[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        u[s] = 32 * w + 16 * s + c;
        bhn[s] = b_hn[d * kH + u[s]];
    }
    bool valid[2];
    size_t row0[2];   // first row (t = 0) of window q in the [M] dimension
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int sq = seq_base + q;
        valid[q] = sq < B;
        if (sq >= B) sq = B - 1;
        row0[q] = (size_t)sq * T;
    }
    const float *gi_d = gi + (size_t)d * gi_dir_stride;
    float hprev[2][2] = {{0.f, 0.f}, {0.f, 0.f}};

    // gi prefetch ring: gq[p][(s*2+q)*3 + gate]
    float gq[PF][12];
    auto load_gi = [&](int step, float (&dst)[12]) {
        if (step >= T) step = T - 1;
        const int t = reverse ? (T - 1 - step) : step;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float *p = gi_d + (row0[q] + t) * kG + u[s];
#pragma unroll
                for (int gate = 0; gate < 3; ++gate)
                    dst[(s * 2 + q) * 3 + gate] = p[gate * kH];
            }
    };
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int i = 0; i < 12; ++i) gq[p][i] = 0.f;

    // LDS addressing (bytes)
    const int rd_off = g * kHGroupStride + c * 16;                          // + ks*kHKStride
    const int wr_off = w * kHKStride + (c >> 2) * kHGroupStride + (4 * g) * 16 + (c & 3) * 4;

    // Pin every loop-invariant global load (weights, b_hn) as complete BEFORE the loop: an empty
    // asm use makes hipcc wait for the value here.  Otherwise its waitcnt pass carries them as
    // "possibly pending" around the back edge and emits vmcnt(0) at their first use in every
    // iteration -- one exposed HBM round trip per step.
#pragma unroll
    for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) asm volatile("" ::"v"(wf[t6][ks][sp]));
    asm volatile("" ::"v"(bhn[0]), "v"(bhn[1]));
    __syncthreads();

    // The ring is primed by running the loop from step -PF with the compute skipped: every gi
    // load is issued from one static site per ring slot, unconditionally and in ring order, which
    // lets hipcc's waitcnt pass emit counted vmcnt(N>0) waits (a separate prologue, or a load
    // under the step branch, degrades to vmcnt(0) = one HBM round trip per step).
    for (int step0 = -PF; step0 < T; step0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int step = step0 + p;
            if (step >= 0 && step < T) {   // wave-uniform
                const int cur = (step & 1) * kHBufBytes;
                const int nxt = kHBufBytes - cur;
                const int t = reverse ? (T - 1 - step) : step;

                // 1. A operand: h_{t-1} as fp16 (hi, lo) rows, all 128 units
                half8 a[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    a[ks] = *reinterpret_cast<const half8 *>(hbuf + cur + ks * kHKStride + rd_off);

                // 2. gh = h W_hh^T on the matrix core (sub-tile 0 first so that its gate math
                //    can overlap the MFMAs of sub-tile 1)
                floatx4 acc[6];
#pragma unroll
                for (int t6 = 0; t6 < 6; ++t6) acc[t6] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int gate = 0; gate < 3; ++gate) {
                            const int t6 = s * 3 + gate;
                            acc[t6] = mfma16(a[ks], wf[t6][ks][0], acc[t6]);
                            acc[t6] = mfma16(a[ks], wf[t6][ks][1], acc[t6]);
                        }

                // 3. gates, blend, store, re-split into the other LDS buffer
                float hn[2][2];
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const float gh_r = (acc[s * 3 + 0][2 * q] + acc[s * 3 + 0][2 * q + 1]) * inv_scale;
                        const float gh_z = (acc[s * 3 + 1][2 * q] + acc[s * 3 + 1][2 * q + 1]) * inv_scale;
                        const float gh_n = (acc[s * 3 + 2][2 * q] + acc[s * 3 + 2][2 * q + 1]) * inv_scale;
                        const float *gv = &gq[p][(s * 2 + q) * 3];
                        const float r = sigmoid_f(gv[0] + gh_r);
                        const float z = sigmoid_f(gv[1] + gh_z);
                        const float n = tanh_f(gv[2] + r * (gh_n + bhn[s]));
                        const float h = n + z * (hprev[s][q] - n);
                        hprev[s][q] = h;
                        hn[s][q] = h;
                        if (valid[q]) out[(row0[q] + t) * out_stride + d * kH + u[s]] = h;
                    }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    _Float16 hi0, lo0, hi1, lo1;
                    split_f16(hn[0][q] * kActScale, hi0, lo0);
                    split_f16(hn[1][q] * kActScale, hi1, lo1);
                    half2_t vhi = {hi0, hi1};
                    half2_t vlo = {lo0, lo1};
                    *reinterpret_cast<half2_t *>(hbuf + nxt + wr_off + (2 * q) * 16) = vhi;
                    *reinterpret_cast<half2_t *>(hbuf + nxt + wr_off + (2 * q + 1) * 16) = vlo;
                }
                lds_barrier();
            }
            // 4. refill this ring slot PF steps ahead -- unconditionally, so that every path
            //    through the unrolled body issues the same loads in the same order
            load_gi(step + PF, gq[p]);
        }
    }
}

}  // namespace mdk
