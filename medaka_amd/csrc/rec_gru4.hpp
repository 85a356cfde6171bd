// GRU recurrence, four-wave form: the same arithmetic, layouts and packed weights as k_rec_mfma
// (rec_mfma.hpp; PyTorch nn.GRU cell called from reference medaka/architectures/gru.py:66), with ONE wave per
// SIMD that owns 32 hidden units (two 16-unit tiles = the work of waves 2*w4 and 2*w4+1 of the eight-wave
// kernel) instead of two waves per SIMD owning 16 each.
//
// Why (profiles/r2_ablation.txt): the eight-wave step is two serial phases -- matrix pipe busy for 768
// cycles (48 MFMAs per SIMD), then ~600 cycles of chain (tail of the later wave, LDS publish, barrier, LDS
// read) during which the pipe idles -- and the two waves of a SIMD cannot hide each other's chain because the
// younger one's MFMAs queue behind the older one's.  With one wave per SIMD
//   * the A operand (the 4 KB fp16 image of h) is read from LDS by 4 waves instead of 8: 16 KB per step
//     instead of 32 KB, half the LDS time at the head of the step;
//   * the 48 MFMAs of a SIMD are one in-order stream: the first unit tile's gate math, its LDS publish
//     and its HBM store are interleaved with the second tile's MFMAs, so only the second tile's tanh chain
//     follows the last MFMA; there is no older/younger skew in front of the barrier;
//   * every accumulator still receives its MFMAs in the same order: results are bit-identical to k_rec_mfma.
// 4 x (192 weight + ~100 other) VGPRs per lane: one work-group per CU, which is what the latency-bound
// regime (<= 256 work-groups) wants anyway.
#pragma once
#include "common.hpp"
#include "layout.hpp"
#include "rec_mfma.hpp"

namespace mdk {

template <int PF, int NQ, bool XIN, bool HP>
__global__ __launch_bounds__(256, 1) void k_rec_gru4(
    const float *__restrict__ gi,      // !XIN: gi_t (layout.hpp), folded bias, PRE-SCALED by S_d
    const half8 *__restrict__ xfrag,   //  XIN: packed x A-fragments [work-group][t][64 lanes]
    const half8 *__restrict__ wxfrag,  //  XIN: W_ih (+bias row) B-fragments [D][8][3][2][64]
    const half8 *__restrict__ wfrag,   // W_hh B-fragments [D][8][4][3][2][64]
    const float *__restrict__ b_hn,    // [D][128]  (unscaled)
    float *__restrict__ out,           // act_t (layout.hpp)
    int n_tiles, int T, int D, const float *__restrict__ inv_scale_p, int reverse_mask,
    const int *__restrict__ cond, int want, int s0, int ns)
{
    __shared__ __attribute__((aligned(16))) unsigned char hbuf[2 * kHBufBytes];
    if (cond != nullptr && ((*cond != 0) != (want != 0))) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w4 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d = blockIdx.y;
    const int c = lane & 15;
    const int g = lane >> 4;
    const bool reverse = (reverse_mask >> d) & 1;
    const float inv_scale = inv_scale_p[d];
    const float c_sig = -inv_scale * 1.44269504088896340736f;
    const float c_tanh = 2.0f * inv_scale * 1.44269504088896340736f;
    constexpr int NS = HP ? 1 : 2;
    static_assert(HP || NQ <= 2, "fp32-parity mode carries at most 2 windows per lane");

    // weights of unit tiles j = 0, 1 (= waves 2*w4 + j of the eight-wave packing)
    half8 wf[2][4][3][NS];
    half8 wx[2][3][NS];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const half8 *wp = wfrag + ((size_t)(d * 8 + 2 * w4 + j) * 24) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int gate = 0; gate < 3; ++gate)
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) wf[j][ks][gate][sp] = wp[(size_t)((ks * 3 + gate) * 2 + sp) * 64];
        if constexpr (XIN) {
            const half8 *xw = wxfrag + ((size_t)(d * 8 + 2 * w4 + j) * 6) * 64 + lane;
#pragma unroll
            for (int gate = 0; gate < 3; ++gate)
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) wx[j][gate][sp] = xw[(size_t)(gate * 2 + sp) * 64];
        }
    }
    for (int i = tid; i < 2 * kHBufBytes / 4; i += 256) reinterpret_cast<uint32_t *>(hbuf)[i] = 0u;

    float bhn[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bhn[j] = b_hn[d * kH + 16 * (2 * w4 + j) + c] * (1.0f / inv_scale);
    const long tstep = reverse ? -1 : 1;
    const int s_end = s0 + ns;
    const int t_first = reverse ? (T - 1 - s0) : s0;
    // tile j = 1 sits at constant offsets from tile 0 in both layouts (gi: 2 sub-blocks of 3 x 64; act: 2 x 64)
    constexpr int kGiTile = 2 * 3 * 64, kActTile = 2 * 64;
    const float *gp[NQ];
    float *op[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        int win = blockIdx.x * (4 * NQ) + NQ * g + q;
        if (win >= n_tiles * kTileWin) win = n_tiles * kTileWin - 1;
        const int tile = win >> 3, wt = win & 7;
        const int llane = (wt >> 1) * 16 + c, lq = wt & 1;
        gp[q] = gi + gi_block(d, n_tiles, tile, T, t_first, 3) + gi_in_block(2 * w4, lq, 0, llane, 3);
        op[q] = out + act_block(D, tile, T, t_first) + act_in_block(d, 2 * w4, lq, llane);
    }
    const half8 *xp = xfrag + ((size_t)blockIdx.x * T + t_first) * 64 + lane;
    const long gstride = tstep * gi_block_floats(3);
    const long ostride = tstep * (long)(D * 1024);
    const long xstride = tstep * 64;
    float hprev[2][NQ];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < NQ; ++q) hprev[j][q] = 0.f;

    // prefetch ring (see rec_mfma.hpp for why it is primed and drained like this)
    float gq[PF][2][3 * NQ];
    half8 xq[PF];
    auto refill = [&](int p, bool advance) {
        if constexpr (XIN) {
            xq[p] = *xp;
            if (advance) xp += xstride;
        } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const auto v = load_run<3>(gp[q] + j * kGiTile);
#pragma unroll
                    for (int gate = 0; gate < 3; ++gate) gq[p][j][q * 3 + gate] = v[gate];
                }
                if (advance) gp[q] += gstride;
            }
        }
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) {
#pragma unroll
        for (int i = 0; i < 8; ++i) xq[p][i] = (_Float16)0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 3 * NQ; ++i) gq[p][j][i] = 0.f;
    }
#pragma unroll
    for (int p = 0; p + 1 < PF; ++p) refill(p, s0 + p + 1 < s_end);
#pragma unroll
    for (int p = 0; p + 1 < PF; ++p) {
        if constexpr (XIN) asm volatile("" ::"v"(xq[p]));
        else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 3 * NQ; ++i) asm volatile("" ::"v"(gq[p][j][i]));
        }
    }

    const int rd_off = g * kHGroupStride + c * 16;
    int wr_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
        wr_off[j] = w4 * kHKStride + (2 * j + (c >> 3)) * kHGroupStride + (4 * g) * 16 + (c & 7) * 2;
    auto publish = [&](int j, const float (&hv)[NQ], int nxt) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            _Float16 hi, lo;
            split_f16(hv[q] * kActScale, hi, lo);
            if constexpr (HP) {
                *reinterpret_cast<_Float16 *>(hbuf + nxt + wr_off[j] + q * 16) = hi;
            } else {
                *reinterpret_cast<_Float16 *>(hbuf + nxt + wr_off[j] + (2 * q) * 16) = hi;
                *reinterpret_cast<_Float16 *>(hbuf + nxt + wr_off[j] + (2 * q + 1) * 16) = lo;
            }
        }
    };
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int gate = 0; gate < 3; ++gate)
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) asm volatile("" ::"v"(wf[j][ks][gate][sp]));
    __syncthreads();
    if (s0 > 0) {   // resume: h of scan step s0 - 1 from the output, and its fp16 image
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float hv[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) { hv[q] = *(op[q] + j * kActTile - ostride); hprev[j][q] = hv[q]; }
            publish(j, hv, (s0 & 1) * kHBufBytes);
        }
        __syncthreads();
    }

    floatx4 xar[2], xaz[2], xgn[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { xar[j] = floatx4{0.f, 0.f, 0.f, 0.f}; xaz[j] = xar[j]; xgn[j] = xar[j]; }
    if constexpr (XIN) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) {
                xar[j] = mfma16(xq[0], wx[j][0][sp], xar[j]);
                xaz[j] = mfma16(xq[0], wx[j][1][sp], xaz[j]);
                xgn[j] = mfma16(xq[0], wx[j][2][sp], xgn[j]);
            }
    }
    auto rows = [&](const floatx4 &v, int q) {
        if constexpr (HP) return v[q]; else return v[2 * q] + v[2 * q + 1];
    };

    for (int step0 = s0; step0 < s_end; step0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int step = step0 + p;       // steps >= s_end (ns not a multiple of PF) run too, stores masked
            const int cur = (step & 1) * kHBufBytes;
            const int nxt = kHBufBytes - cur;
            half8 a[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a[ks] = *reinterpret_cast<const half8 *>(hbuf + cur + ks * kHKStride + rd_off);

            floatx4 ar[2], az[2], anh[2], anl[2], gin[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                ar[j] = floatx4{0.f, 0.f, 0.f, 0.f}; az[j] = ar[j]; anh[j] = ar[j]; anl[j] = ar[j]; gin[j] = ar[j];
                if constexpr (XIN) { ar[j] = xar[j]; az[j] = xaz[j]; gin[j] = xgn[j]; }
            }
            // ---- phase A: all 24 MFMAs of unit tile 0 (r, z interleaved, then n)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) {
                    ar[0] = mfma16(a[ks], wf[0][ks][0][sp], ar[0]);
                    az[0] = mfma16(a[ks], wf[0][ks][1][sp], az[0]);
                }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                anh[0] = mfma16(a[ks], wf[0][ks][2][0], anh[0]);
                if constexpr (!HP) anl[0] = mfma16(a[ks], wf[0][ks][2][1], anl[0]);
            }
            refill((p + PF - 1) % PF, (step + PF) < s_end);
            // h of the PREVIOUS step leaves for HBM from here (off the tail): h_prev still holds it
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    if (step > s0 && step <= s_end) *(op[q] + j * kActTile - ostride) = hprev[j][q];
            __builtin_amdgcn_sched_barrier(0);
            // ---- phase B: r, z tiles of unit tile 1; ALL of tile 0's gate math and its LDS publish under them
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) {
                    ar[1] = mfma16(a[ks], wf[1][ks][0][sp], ar[1]);
                    az[1] = mfma16(a[ks], wf[1][ks][1][sp], az[1]);
                }
            auto gate_rz = [&](int j, float (&rr)[NQ], float (&zz)[NQ], float (&gnv)[NQ]) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    float tr, tz;
                    if constexpr (XIN) { tr = rows(ar[j], q); tz = rows(az[j], q); gnv[q] = rows(gin[j], q); }
                    else {
                        tr = gq[p][j][q * 3] + rows(ar[j], q);
                        tz = gq[p][j][q * 3 + 1] + rows(az[j], q);
                        gnv[q] = gq[p][j][q * 3 + 2];
                    }
                    rr[q] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(tr * c_sig));
                    zz[q] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(tz * c_sig));
                }
            };
            auto gate_n = [&](int j, const float (&rr)[NQ], const float (&zz)[NQ], const float (&gnv)[NQ], float (&hv)[NQ]) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    float tn;
                    if constexpr (HP) tn = anh[j][q] + bhn[j];
                    else tn = ((anh[j][2 * q] + anl[j][2 * q]) + (anh[j][2 * q + 1] + anl[j][2 * q + 1])) + bhn[j];
                    const float an = __builtin_fmaf(rr[q], tn, gnv[q]);
                    const float e = __builtin_amdgcn_exp2f(an * c_tanh);
                    const float n = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
                    const float h = __builtin_fmaf(zz[q], hprev[j][q] - n, n);
                    hprev[j][q] = h;
                    hv[q] = h;
                }
            };
            {
                float rr[NQ], zz[NQ], gnv[NQ], hv[NQ];
                gate_rz(0, rr, zz, gnv);
                gate_n(0, rr, zz, gnv, hv);
                publish(0, hv, nxt);
            }
#pragma unroll
            for (int i = 0; i < 4 * NS; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);              // 2 MFMAs
                __builtin_amdgcn_sched_group_barrier(0x002, 6 * NQ + 2, 0);       // VALU of tile 0's chain
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);              // its LDS stores where they fall
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- phase C: n tile of unit tile 1 with its own sigmoids under it
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                anh[1] = mfma16(a[ks], wf[1][ks][2][0], anh[1]);
                if constexpr (!HP) anl[1] = mfma16(a[ks], wf[1][ks][2][1], anl[1]);
            }
            float rr1[NQ], zz1[NQ], gnv1[NQ], hv1[NQ];
            gate_rz(1, rr1, zz1, gnv1);
#pragma unroll
            for (int i = 0; i < 4 * NS; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2 * NQ, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- tail: only tile 1's tanh chain, blend and publish follow the last MFMA
            gate_n(1, rr1, zz1, gnv1, hv1);
            publish(1, hv1, nxt);
#pragma unroll
            for (int q = 0; q < NQ; ++q) op[q] += ostride;
            if constexpr (XIN) {
                // layer-0 input projection of the NEXT step: independent of h, runs under the LDS/barrier latency
                const half8 xn = xq[(p + 1) % PF];
                const floatx4 zero = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    xar[j] = mfma16(xn, wx[j][0][0], zero);
                    xaz[j] = mfma16(xn, wx[j][1][0], zero);
                    xgn[j] = mfma16(xn, wx[j][2][0], zero);
                    if constexpr (!HP) {
                        xar[j] = mfma16(xn, wx[j][0][1], xar[j]);
                        xaz[j] = mfma16(xn, wx[j][1][1], xaz[j]);
                        xgn[j] = mfma16(xn, wx[j][2][1], xgn[j]);
                    }
                }
            }
            lds_barrier();
        }
    }
    // the loop runs whole groups of PF steps: when the last of them is step s_end - 1 nobody stored it yet
    if ((s_end - s0) % PF == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < NQ; ++q) *(op[q] + j * kActTile - ostride) = hprev[j][q];
    }
}

}  // namespace mdk
