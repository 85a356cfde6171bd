// GRU recurrence on the matrix cores -- the dominant kernel of the consensus forward pass.
//
// Computes, for one layer and one direction, the T dependent steps of
//     gh = W_hh h + b_hh ;  r = s(gi_r + gh_r) ; z = s(gi_z + gh_z) ;
//     n = tanh(gi_n + r * gh_n) ; h = (1 - z) n + z h          (PyTorch nn.GRU cell, called from
// reference medaka/architectures/gru.py:66) for a tile of 8 windows per work-group.
//
// MI355X mapping (DESIGN.md section "recurrence kernel"):
//   * one 512-thread work-group (8 waves, two per SIMD) per (8-window tile, direction);
//     wave w8 owns hidden units [16*w8, 16*w8+16) of all three gates = 3 MFMA column tiles;
//   * W_hh lives in registers for the whole kernel as pre-packed fp16 hi/lo B-fragments
//     (96 VGPRs per lane); h_t is staged in LDS as the fp16 hi/lo A-operand (4 KB, double
//     buffered) -- zero global-memory round trips on the step-to-step dependency;
//   * fp32 parity through an fp16x2 split: A rows = (window, hi|lo) -> 16 rows for 8 windows,
//     B = W_hi then W_lo into fp32 accumulators, so  acc[row hi] + acc[row lo]
//     = (h_hi + h_lo)(W_hi + W_lo) = h W to ~2^-22 relative, fp32 accumulate;
//   * sigmoid/tanh, the z-blend, the fp16 re-split and the store of h_t are fused behind the
//     MFMAs; gi (input projection, bias folded, pre-scaled) is prefetched PF steps ahead.
#pragma once
#include "common.hpp"
#include "layout.hpp"

#ifndef MDK_REC_PRIO
#define MDK_REC_PRIO 3      // wave priority of the recurrence kernels (0..3)
#endif

namespace mdk {


// byte offset of a __shared__ object inside the work-group's LDS allocation
__device__ __forceinline__ unsigned lds_offset(const void *p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void *)p;
}

constexpr int kXfragLanes = 32;             // lanes of a packed layer-0 input block (k_pack_x): K + 1 <= 16 of the k-step's 32 slots
// One lane group of the A image: 16 rows x 16 B, NO pad.  ds_read_b128 is served in four groups of 16 lanes that mix
// lanes of two lane groups ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS table): with rows 256 B apart those 16
// lanes cover all 64 banks once; the 16-byte pad rounds 1-4 used (for the 2-byte publishes of the two half-groups of a
// wave, which then share banks) made lanes 12 and 27 of every read meet on banks 48-51 -- and the reads move sixteen times
// the bytes of the publishes.  Measured (profiles/r5_experiments/README.md): half precision 3.61 -> 3.50 ms per forward,
// fp32 parity 6.34 -> 6.31; 288 is worse than 272.
constexpr int kHGroupStride = 256;
constexpr int kHKStride = 4 * kHGroupStride;  // one k-step (32 units) of the A image
constexpr int kHBufBytes = 4 * kHKStride;     // 4096 B per buffer

// Why 8 waves: measured on MI355X, a 4-wave version (one wave per SIMD, 6 tiles per wave) is
// instruction-ISSUE bound (~250 instructions x ~4 cycles per step) -- removing all of its MFMAs
// shortened the step by only 3 %.  Two waves per SIMD double the issue rate and let one wave's
// gate math run while the other is blocked on the matrix pipe (8.3 -> 6.6 ms per launch).
// The arithmetic stays in the scaled domain of the accumulator: gi and b_hn arrive multiplied by
// S = 2^10 * w_scale and 1/S is folded into the exp2 argument constants.
// ABL: timing-only ablation mask (1 no MFMA, 2 no gate math, 4 no barrier, 8 no gi loads,
// 16 no stores); results are garbage unless ABL == 0.
// XIN (layer 0): the input projection is fused -- see k_pack_x below.
//   wave w8 owns hidden units [16*w8, 16*w8+16) of the three gates = 3 MFMA column tiles;
//   W_hh fragments: 96 registers per lane; 2 hidden values per lane per step.
// Fragment layout [D][8 waves][4 ksteps][3 gates][2 hi/lo][64 lanes], natural k order
// (slot (ks, lane-group gq, i) = unit 32*ks + 8*gq + i).
// NQ = windows per lane (1 or 2): a work-group carries 4*NQ windows.  NQ = 1 leaves half of the
// MFMA rows zero -- free, the matrix pipe is not the limiter -- and halves the per-step VALU and
// memory instruction count; it is used whenever 4-window tiles still fit the chip in one wave of
// work-groups (B <= ~500).
// HP: half-precision mode (`TorchModel.half()`): fp16 operands without the hi/lo split -- one A
// row per window (row 4g + q, up to NQ = 4 -> 16 windows per work-group), 12 MFMAs per wave.
// CELL: 0 = GRU (3 gate tiles r,z,n), 1 = LSTM (4 gate tiles i,f,g,o; PyTorch nn.LSTM cell, used by
// the read-level model, reference latent_space_lstm.py:129-149).
// DS (GRU, barrier schedule): the HBM store of h_t is deferred to the MFMA phase of step t+1 (h_prev still
// holds the value), so its address arithmetic and issue leave the tail between the last MFMA and the LDS
// publish.  Ablations (profiles/r2_ablation.txt): the stores cost ~115 cycles of a ~1380-cycle step.
template <int PF, int NQ, bool XIN, bool HP, int CELL = 0, int ABL = 0, bool DS = false>
__global__ __launch_bounds__(512, 2) void k_rec_mfma(
    const float *__restrict__ gi,      // !XIN: gi_t (layout.hpp), folded bias, PRE-SCALED by S_d
    const half8 *__restrict__ xfrag,   //  XIN: packed x A-fragments [work-group][t][kXfragLanes]
    const half8 *__restrict__ wxfrag,  //  XIN: W_ih (+bias row) B-fragments [D][8][3][2][64]
    const half8 *__restrict__ wfrag,   // W_hh B-fragments [D][8][4][3][2][64]
    const float *__restrict__ b_hn,    // [D][128]  (unscaled)
    float *__restrict__ out,           // act_t (layout.hpp)
    int n_tiles, int T, int D, const float *__restrict__ inv_scale_p, int reverse_mask,
    const int *__restrict__ cond, int want,
    int s0, int ns)   // steps [s0, s0 + ns) of the scan (scan step s is t = s, or T-1-s when reversed);
                      // s0 > 0 resumes from the h this kernel stored at scan step s0 - 1 (GRU only)
{
    __shared__ __attribute__((aligned(16))) unsigned char hbuf[2 * kHBufBytes];
    // fused/unfused layer-0 selection is made on the device (input range flag of k_pack_x)
    if (cond != nullptr && ((*cond != 0) != (want != 0))) return;
    // The recurrence is a chain of 2 x T dependent steps: whatever else is resident on this CU -- the side-stream
    // projection / head / pack kernels of this forward, or another process's kernels under `launch.py
    // --procs-per-gpu`, or a foreign tenant -- must not win instruction-issue arbitration against it.  Measured (round 3,
    // profiles/r3_experiments/README.md): next to a co-resident kernel that issues MFMAs back to back on every SIMD the
    // forward takes 29 ms at priority 0 and 19 ms at priority 3 (what is left is the clock dropping under the extra
    // load); alone, and with K of our own processes sharing the GPU, nothing changes (A/B within one box).
    __builtin_amdgcn_s_setprio(MDK_REC_PRIO);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d = blockIdx.y;
    const int c = lane & 15;
    const int g = lane >> 4;
    const bool reverse = (reverse_mask >> d) & 1;
    const float inv_scale = inv_scale_p[d];
    const float c_sig = -inv_scale * 1.44269504088896340736f;
    const float c_tanh = 2.0f * inv_scale * 1.44269504088896340736f;

    constexpr int NS = HP ? 1 : 2;   // fp16 pieces per operand
    constexpr int NG = CELL ? 4 : 3; // gate tiles per wave
    static_assert(!(XIN && CELL), "the fused input projection exists for the GRU layer 0 only");
    static_assert(!(CELL && ABL), "ablation builds exist for the GRU cell only");
    static_assert(HP || NQ <= 2, "fp32-parity mode carries at most 2 windows per lane");
    half8 wf[4][NG][NS];
    {
        const half8 *wp = wfrag + ((size_t)(d * 8 + w8) * (8 * NG)) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int gate = 0; gate < NG; ++gate)
#pragma unroll
                for (int sp = 0; sp < NS; ++sp)
                    wf[ks][gate][sp] = wp[(size_t)((ks * NG + gate) * 2 + sp) * 64];
    }
    half8 wx[3][NS];
    if constexpr (XIN) {
        const half8 *wp = wxfrag + ((size_t)(d * 8 + w8) * 6) * 64 + lane;
#pragma unroll
        for (int gate = 0; gate < 3; ++gate)
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) wx[gate][sp] = wp[(size_t)(gate * 2 + sp) * 64];
    }
    for (int i = tid; i < 2 * kHBufBytes / 4; i += 512) reinterpret_cast<uint32_t *>(hbuf)[i] = 0u;

    const int u = 16 * w8 + c;
    const float bhn = CELL ? 0.f : b_hn[d * kH + u] * (1.0f / inv_scale);
    // this work-group carries windows [4*NQ*blockIdx.x, +4*NQ); lane group g holds windows
    // NQ*g + q of it.  Layout tiles are 8 windows (layout.hpp): window w -> tile w>>3,
    // lane-group (w&7)>>1, q (w&7)&1.
    const long tstep = reverse ? -1 : 1;
    const int s_end = s0 + ns;
    const int t_first = reverse ? (T - 1 - s0) : s0;
    const float *gp[NQ];
    float *op[NQ];
    bool live[NQ];
    auto stores = [&](int q) { if constexpr (NQ == 4) return live[q]; else return true; };
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        int win = blockIdx.x * (4 * NQ) + NQ * g + q;
        // only NQ = 4 can run past the padding (a tile count that is odd): those lanes load from the last window, for
        // the addresses' sake, and store nothing -- with the fused layer-0 input their x rows are zero, not a copy
        live[q] = win < n_tiles * kTileWin;
        if (!live[q]) win = n_tiles * kTileWin - 1;
        const int tile = win >> 3, wt = win & 7;
        const int llane = (wt >> 1) * 16 + c, lq = wt & 1;
        gp[q] = gi + gi_block(d, n_tiles, tile, T, t_first, NG) + gi_in_block(w8, lq, 0, llane, NG);
        op[q] = out + act_block(D, tile, T, t_first) + act_in_block(d, w8, lq, llane);
    }
    // (only lane groups 0, 1 of the k-step carry data -- features, the bias row -- and W_ih's rows behind them are zero: the
    // packed block is 32 lanes = 512 bytes, lanes 32..63 read a copy of lanes 0..31 that multiplies zeros)
    const half8 *xp = xfrag + ((size_t)blockIdx.x * T + t_first) * kXfragLanes + (lane & (kXfragLanes - 1));
    const long gstride = tstep * gi_block_floats(NG);
    const long ostride = tstep * (long)(D * 1024);
    const long xstride = tstep * kXfragLanes;
    float hprev[NQ];   // GRU: h_{t-1};  LSTM: the cell state c_{t-1}
#pragma unroll
    for (int q = 0; q < NQ; ++q) hprev[q] = 0.f;

    // rows of the A operand: fp32-parity mode 4g + 2q + {0: hi, 1: lo}; half mode 4g + q.
    // Unused rows stay zero.
    // Prefetch ring, PF steps deep: gq[p][q*3 + gate] (or the packed x fragment xq[p]).  Primed
    // here and fully drained once (one HBM round trip per launch): the main loop is then entered
    // with nothing in flight, so hipcc's waitcnt pass sees the ring only in its steady-state issue
    // order and emits counted vmcnt(N >> 0) waits.  (A peeled / reordered priming sequence makes it
    // emit vmcnt(~0) in the first unrolled step of EVERY iteration = one exposed HBM latency per
    // PF steps.)
    float gq[PF][NG * NQ];
    half8 xq[PF];
    auto refill = [&](int p, bool advance) {
        if constexpr (XIN) {
            if constexpr (!(ABL & 8)) xq[p] = *xp;
            if (advance) xp += xstride;
        } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if constexpr (ABL & 8) {
#pragma unroll
                    for (int gate = 0; gate < NG; ++gate) gq[p][q * NG + gate] = 0.f;
                } else {
                    // one 12-byte (GRU) / 16-byte (LSTM) load: the NG gates of this lane's (unit, window) are adjacent
                    const auto v = load_run<NG>(gp[q]);
#pragma unroll
                    for (int gate = 0; gate < NG; ++gate) gq[p][q * NG + gate] = v[gate];
                }
                if (advance) gp[q] += gstride;   // stop advancing at the last row
            }
        }
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) {
#pragma unroll
        for (int i = 0; i < 8; ++i) xq[p][i] = (_Float16)0.f;
#pragma unroll
        for (int i = 0; i < NG * NQ; ++i) gq[p][i] = 0.f;
    }
    // slots 0..PF-2 hold steps 0..PF-2; slot PF-1 is filled during step 0 (for step PF-1): a slot
    // is always refilled one step after it was consumed, from inside the MFMA phase
#pragma unroll
    for (int p = 0; p + 1 < PF; ++p) {
        refill(p, s0 + p + 1 < s_end);
    }
#pragma unroll
    for (int p = 0; p + 1 < PF; ++p) {
        if constexpr (XIN) asm volatile("" ::"v"(xq[p]));
        else {
#pragma unroll
            for (int i = 0; i < NG * NQ; ++i) asm volatile("" ::"v"(gq[p][i]));
        }
    }

    const int rd_off = g * kHGroupStride + c * 16;
    const int wr_off = (w8 >> 1) * kHKStride + (2 * (w8 & 1) + (c >> 3)) * kHGroupStride +
                       (4 * g) * 16 + (c & 7) * 2;
    auto publish = [&](const float (&hv)[NQ], int nxt) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            _Float16 hi, lo;
            split_f16(hv[q] * kActScale, hi, lo);
            if constexpr (HP) {
                *reinterpret_cast<_Float16 *>(hbuf + nxt + wr_off + q * 16) = hi;
            } else {
                *reinterpret_cast<_Float16 *>(hbuf + nxt + wr_off + (2 * q) * 16) = hi;
                *reinterpret_cast<_Float16 *>(hbuf + nxt + wr_off + (2 * q + 1) * 16) = lo;
            }
        }
    };

#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int gate = 0; gate < NG; ++gate)
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) asm volatile("" ::"v"(wf[ks][gate][sp]));
    if constexpr (XIN) {
#pragma unroll
        for (int gate = 0; gate < 3; ++gate)
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) asm volatile("" ::"v"(wx[gate][sp]));
    }
    asm volatile("" ::"v"(bhn));
    __syncthreads();
    if constexpr (CELL == 0) {
        if (s0 > 0) {   // resume: h of scan step s0 - 1 from the output, and its fp16 image
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const float h = *(op[q] - ostride);
                hprev[q] = h;
                _Float16 hi, lo;
                split_f16(h * kActScale, hi, lo);
                unsigned char *img = hbuf + (s0 & 1) * kHBufBytes + wr_off;
                if constexpr (HP) {
                    *reinterpret_cast<_Float16 *>(img + q * 16) = hi;
                } else {
                    *reinterpret_cast<_Float16 *>(img + (2 * q) * 16) = hi;
                    *reinterpret_cast<_Float16 *>(img + (2 * q + 1) * 16) = lo;
                }
            }
            __syncthreads();
        }
    }

    floatx4 xar = floatx4{0.f, 0.f, 0.f, 0.f}, xaz = xar, xgn = xar;
    if constexpr (XIN && !(ABL & 1)) {
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) {
            xar = mfma16(xq[0], wx[0][sp], xar);
            xaz = mfma16(xq[0], wx[1][sp], xaz);
            xgn = mfma16(xq[0], wx[2][sp], xgn);
        }
    }
    // ABL & 64: per-phase cycle accounting with s_memtime (perturbs the schedule; debug only)
    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tprev = 0;
    auto stamp = [&](int ph) {
        if constexpr (ABL & 64) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            tph[ph] += now - tprev;
            tprev = now;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if constexpr (ABL & 64) tprev = __builtin_amdgcn_s_memtime();


    for (int step0 = s0; step0 < s_end; step0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int step = step0 + p;
            {   // steps >= s_end (ns not a multiple of PF) run too, with their stores masked
                const int cur = (step & 1) * kHBufBytes;
                const int nxt = kHBufBytes - cur;
                stamp(0);   // refill issue + loop overhead

                half8 a[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    a[ks] = *reinterpret_cast<const half8 *>(hbuf + cur + ks * kHKStride + rd_off);

                if constexpr (ABL & 64) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp(1); }   // LDS read latency
                // sum of this window's accumulator rows (hi + lo rows, or the single fp16 row)
                auto rows = [&](const floatx4 &v, int q) {
                    if constexpr (HP) return v[q]; else return v[2 * q] + v[2 * q + 1];
                };
                float hn[NQ];
                if constexpr (CELL == 0) {
                    floatx4 ar = floatx4{0.f, 0.f, 0.f, 0.f}, az = ar, anh = ar, anl = ar, gin = ar;
                    if constexpr (XIN) { ar = xar; az = xaz; gin = xgn; }   // projected one step ahead
                    if constexpr (ABL & 1) {
    #pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            ar[ks] = (float)a[ks][0]; az[ks] = (float)a[ks][2];
                            anh[ks] = (float)a[ks][4]; anl[ks] = (float)a[ks][6];
                        }
                    } else {
    #pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
    #pragma unroll
                            for (int sp = 0; sp < NS; ++sp) {
                                ar = mfma16(a[ks], wf[ks][0][sp], ar);
                                az = mfma16(a[ks], wf[ks][1][sp], az);
                            }
                        }
                    }
                    // refill the ring slot consumed in the PREVIOUS step (data of step + PF - 1): the
                    // vector-memory issue hides under the MFMAs; unconditional, in ring order
                    refill((p + PF - 1) % PF, (step + PF) < s_end);
                    if constexpr (DS) {
                        static_assert(!(DS && (ABL || CELL)), "deferred stores: GRU production builds only");
#pragma unroll
                        for (int q = 0; q < NQ; ++q)
                            if (step > s0 && step <= s_end && stores(q)) *(op[q] - ostride) = hprev[q];   // h of the previous step
                    }
                    // --- scheduling fence: everything above (r,z tiles) is issued before the n tiles;
                    // the sigmoids of r,z below share a region with the n MFMAs and are interleaved
                    // with them (1 MFMA : 2 VALU), so only the tanh/blend/split chain of n is exposed
                    // after the last MFMA.
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (!(ABL & 1)) {
    #pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            anh = mfma16(a[ks], wf[ks][2][0], anh);
                            if constexpr (!HP) anl = mfma16(a[ks], wf[ks][2][1], anl);
                        }
                    }
                    float rr[NQ], zz[NQ], gnv[NQ];
    #pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        float gr, gz;
                        if constexpr (XIN) { gr = 0.f; gz = 0.f; gnv[q] = rows(gin, q); }
                        else { gr = gq[p][q * NG]; gz = gq[p][q * NG + 1]; gnv[q] = gq[p][q * NG + 2]; }
                        const float tr = XIN ? rows(ar, q) : (gr + rows(ar, q));
                        const float tz = XIN ? rows(az, q) : (gz + rows(az, q));
                        rr[q] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(tr * c_sig));
                        zz[q] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(tz * c_sig));
                    }
                    if constexpr (!(ABL & 1)) {
    #pragma unroll
                        for (int i = 0; i < 4 * NS; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // 2 VALU
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    stamp(2);   // MFMA issue (+ sigmoids)
    #pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        if constexpr (ABL & 2) {
                            const float h = ((rows(ar, q) + rows(az, q)) + (rows(anh, q) + rows(anl, q))) * 1e-6f +
                                            (rr[q] + zz[q] + gnv[q]) * 1e-9f;
                            hprev[q] = h; hn[q] = h;
                            if constexpr (!(ABL & 16)) { if (step < s_end && stores(q)) op[q][0] = h; }
                            continue;
                        }
                        float tn;
                        if constexpr (HP) tn = anh[q] + bhn;
                        else tn = ((anh[2 * q] + anl[2 * q]) + (anh[2 * q + 1] + anl[2 * q + 1])) + bhn;
                        const float an = __builtin_fmaf(rr[q], tn, gnv[q]);
                        const float e = __builtin_amdgcn_exp2f(an * c_tanh);
                        const float n = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
                        const float h = __builtin_fmaf(zz[q], hprev[q] - n, n);
                        hprev[q] = h;
                        hn[q] = h;
                        if constexpr (!(ABL & 16) && !DS) { if (step < s_end && stores(q)) op[q][0] = h; }
                    }
                } else {
                    // ---- LSTM cell: i, f, g tiles first; the o tile last, with the cell update
                    // (3 sigmoids/tanh, c = f c + i g, tanh c) interleaved under its MFMAs; only
                    // sigmoid(o) * tanh(c) is exposed after the last MFMA.
                    floatx4 ai = floatx4{0.f, 0.f, 0.f, 0.f}, af = ai, ag = ai, ao = ai;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                        for (int sp = 0; sp < NS; ++sp) {
                            ai = mfma16(a[ks], wf[ks][0][sp], ai);
                            af = mfma16(a[ks], wf[ks][1][sp], af);
                            ag = mfma16(a[ks], wf[ks][2][sp], ag);
                        }
                    }
                    refill((p + PF - 1) % PF, (step + PF) < s_end);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                        for (int sp = 0; sp < NS; ++sp) ao = mfma16(a[ks], wf[ks][3][sp], ao);
                    }
                    float tc[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const float ti = gq[p][q * NG + 0] + rows(ai, q);
                        const float tf = gq[p][q * NG + 1] + rows(af, q);
                        const float tg = gq[p][q * NG + 2] + rows(ag, q);
                        const float iv = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ti * c_sig));
                        const float fv = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(tf * c_sig));
                        const float gv = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(tg * c_tanh)), 1.0f);
                        const float cv = __builtin_fmaf(fv, hprev[q], iv * gv);
                        hprev[q] = cv;
                        tc[q] = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(cv * 2.88539008177792681472f)), 1.0f);
                    }
#pragma unroll
                    for (int i = 0; i < 4 * NS; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // 3 VALU
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const float to = gq[p][q * NG + 3] + rows(ao, q);
                        const float ov = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(to * c_sig));
                        const float h = ov * tc[q];
                        hn[q] = h;
                        if (step < s_end && stores(q)) op[q][0] = h;
                    }
                }
                if constexpr (ABL & 64) { asm volatile("" ::"v"(hn[0])); stamp(3); }   // MFMA drain + tanh/blend chain
                publish(hn, nxt);
#pragma unroll
                for (int q = 0; q < NQ; ++q) op[q] += ostride;
                if constexpr (ABL & 64) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp(4); }   // split + LDS write
                if constexpr (XIN && !(ABL & 1)) {
                    // layer-0 input projection of the NEXT step: independent of h, so it is
                    // issued here and runs on the matrix pipe while this wave sits in the LDS
                    // write -> barrier -> LDS read latency chain
                    const half8 xn = xq[(p + 1) % PF];
                    const floatx4 zero = floatx4{0.f, 0.f, 0.f, 0.f};
                    xar = mfma16(xn, wx[0][0], zero);
                    xaz = mfma16(xn, wx[1][0], zero);
                    xgn = mfma16(xn, wx[2][0], zero);
                    if constexpr (!HP) {
                        xar = mfma16(xn, wx[0][1], xar);
                        xaz = mfma16(xn, wx[1][1], xaz);
                        xgn = mfma16(xn, wx[2][1], xgn);
                    }
                }
                if constexpr (ABL & 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else
                lds_barrier();
                stamp(5);   // barrier wait
            }
        }
    }
    if constexpr (DS) {
        // the loop runs whole groups of PF steps: when the last of them is step s_end - 1 nobody stored it yet
        if ((s_end - s0) % PF == 0) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) if (stores(q)) *(op[q] - ostride) = hprev[q];
        }
    }
    if constexpr (ABL & 64) {
        if (lane == 0 && blockIdx.x < 4 && cond != nullptr) {
            unsigned long long *dbg = reinterpret_cast<unsigned long long *>(const_cast<int *>(cond) + 16) +
                                      ((size_t)(blockIdx.y * 4 + blockIdx.x) * 8 + w8) * 6;
#pragma unroll
            for (int i = 0; i < 6; ++i) dbg[i] = tph[i];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Layer-0 fusion.  The input projection of layer 0 has K = num_features (10): instead of
// materialising gi (3 KB per column written and read back, 12 GB per 2 M columns) the counts are
// packed ONCE into the A-fragment the recurrence kernel wants -- rows (window, hi|lo), one k-step
// of 32 slots = [features 0..I-1, constant 1 (bias row), zeros], fp16 hi/lo split of x * sx --
// 1 KB per (work-group, t), and the recurrence issues 6 extra MFMAs per wave per step under the
// LDS latency.  sx is chosen at load time so that (x*sx)(W_ih*swx) lands in the accumulator
// scale S; inputs with |x| * sx beyond fp16 range raise `oor` and the engine falls back to the
// exact fp32 projection kernel ON THE DEVICE (both paths are enqueued, the flag selects).
static __global__ __launch_bounds__(256) void k_pack_x(
    const float *__restrict__ x,   // [B][T][I]
    half8 *__restrict__ xfrag,     // [n_wg][T][kXfragLanes]
    int B, int T, int I, int nq, int hp, int n_wg, float sx, int *__restrict__ oor,
    int t_lo, int nt,              // columns [t_lo, t_lo + nt) of every window (the host path streams x in time slabs)
    SplitPlan sp)                  // sp.S > 1: B x T is the VIRTUAL batch of a split scan and x the real (sp.B, sp.T, I) one --
                                   // virtual window v = k * sp.B + w is columns [sp.start[k], +T) of window w (scan_split.hpp)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n_wg * nt * kXfragLanes;
    if (idx >= total) return;
    const int lane = (int)(idx & (kXfragLanes - 1));
    const size_t wt = idx / kXfragLanes;
    const int t = t_lo + (int)(wt % nt);
    const int wg = (int)(wt / nt);
    const int row = lane & 15, gq = lane >> 4;
    // row -> (window slot q of lane-group g, hi|lo piece): fp32-parity 4g + 2q + split, half 4g + q
    const int g = row >> 2;
    const int q = hp ? (row & 3) : ((row >> 1) & 1);
    const int split = hp ? 0 : (row & 1);
    const int win = wg * 4 * nq + nq * g + q;
    const bool live = (q < nq) && (win < B);
    size_t src = ((size_t)win * T + t) * I;
    if (sp.S > 1 && live) {
        const int k = win / sp.B;
        src = ((size_t)(win - k * sp.B) * sp.T + sp.start[k] + t) * I;
    }
    half8 v;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int f = 8 * gq + i;
        float val = 0.f;
        if (live) {
            if (f < I) val = x[src + f] * sx;
            else if (f == I) val = sx;       // bias row of W_ih
        }
        bad |= !(fabsf(val) <= 60000.0f);
        _Float16 hi, lo;
        split_f16(val, hi, lo);
        v[i] = split ? lo : hi;
    }
    xfrag[((size_t)wg * T + t) * kXfragLanes + lane] = v;
    if (bad) atomicOr(oor, 1);
}

}  // namespace mdk
