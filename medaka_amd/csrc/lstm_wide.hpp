// LSTM(384) of the bundled read-level models (`rl_lstm384`: LatentSpaceLSTM(lstm_size=384,
// cnn_size=128, bidirectional=False), reference medaka/architectures/latent_space_lstm.py:129-149).
//
// One direction's recurrent matrix is 1536 x 384 = 2.4 MB as fp16 hi/lo fragments: it does not fit
// the registers + LDS of one CU (0.67 MB), and re-streaming it from L2 every step would cost ~9 us
// per step.  So the hidden units are split over a CLUSTER of 12 work-groups (= 12 CUs):
//
//   * member m owns units [32m, 32m+32) of all four gates = 128 gate columns; its wave w8 owns the
//     16 gate columns (i, f, g, o) x units 32m + 4*w8 + 0..3 with their W_hh fragments resident in
//     registers (12 k-steps x hi/lo = 96 VGPRs).  W is the *A* operand of the MFMA (rows = gate
//     columns ordered 4*unit + gate) and h the B operand (columns = windows), so the accumulator
//     of lane (g, c) is exactly (i, f, g, o) of unit g for window c: the cell update needs no
//     cross-lane traffic (fp32-parity mode: one DPP add joins the hi and lo columns of a window);
//   * every step each member needs the WHOLE h_{t-1} (8 windows x 384 units).  Members publish their
//     32 units as 8-byte {fp16 hi, fp16 lo, step tag} granules, one 8-byte store each, and gather
//     all 3072 granules of the step with 16-byte L1-bypassing (sc1) loads, re-polling until each
//     8-byte half carries the current tag: the data-tagged granule needs no flag and no fence
//     (MI355X_MICROARCH.md, hand-off form R2).  The stores are agent-scope atomics (write-through,
//     valid across XCDs) unless the members verified at kernel start that they share one XCD, in
//     which case plain stores that stay in that XCD's L2 are several times faster.  Two parity
//     buffers suffice: nobody can publish step t+2 before everybody has gathered step t
//     (DESIGN.md 4.5);
//   * the gathered granules are written into the same LDS A-operand image the 128-unit kernel
//     uses (rec_mfma.hpp), 12 k-steps long; rows = (window, hi|lo) as there;
//   * above 16 groups, two 8-window groups are interleaved per cluster so that one group's
//     exchange latency is covered by the other group's MFMAs;
//
// Cluster members must be co-resident (they spin on each other): the grid is 8 XCDs x 2 clusters
// x 12 members = 192 work-groups <= 256 CUs, one per CU, launched on an otherwise idle device;
// work-group b lands on XCD b % 8 (observed, used for speed only: a cluster shares one L2), and
// every spin is bounded -- on time-out the kernel raises `status[0]` and exits instead of hanging.
#pragma once
#include "common.hpp"
#include "rec_mfma.hpp"

namespace mdk {

constexpr int kWH = 384;                       // hidden units
constexpr int kWG4 = 4 * kWH;                  // gate columns
constexpr int kWC = 12;                        // work-groups (CUs) per cluster
constexpr int kWKS = kWH / 32;                 // k-steps of the recurrent contraction
constexpr int kWWin = 8;                       // windows per cluster (fp32-parity rows = 16)
constexpr int kWImgBytes = kWKS * kHKStride;   // 13 056 B per A image
constexpr int kWMaxClusters = 16;              // 2 per XCD
constexpr int kWGranules = kWWin * kWH;        // per parity buffer
constexpr size_t kWExchWords = (size_t)kWMaxClusters * 4 * kWGranules + (size_t)kWMaxClusters * 16;   // 2 groups x 2 parities + XCD headers
constexpr int kWSpinLimit = 1 << 20;           // ~1-2 s of polling before giving up (a member died mid-kernel: never seen)
// The placement handshake is where a cluster finds out that its 12 members are NOT all resident (fewer than 192
// CUs free: another tenant holds them).  It is bounded in wall-clock time, not in polls: 50 ms of the constant
// 100 MHz clock (s_memrealtime) -- launch skew between resident work-groups is microseconds, a foreign kernel may
// hold CUs for a few milliseconds -- so that a GPU that cannot host the kernel is reported within ~0.1 s (two tries,
// rl_api.hip) instead of after seconds of spinning.
constexpr unsigned long long kWHandshakeTicks = 5000000ull;

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// ABL: timing-only ablation mask (1 no gi loads, 2 accept the first poll, 4 no exchange at all,
// 8 no h stores); results are garbage unless ABL == 0.
//
// Two window groups (A, B) of 8 windows are interleaved per cluster: while one group's published h
// travels through L2, the other group's MFMAs and cell math run, so a gather normally finds its
// granules on the first poll:
//     [C_A(t) + G_B(t-1)]  barrier  [C_B(t) + G_A(t)]  barrier     (C = compute + publish, G = gather)
// NGRP = 1: one group per cluster, the exchange latency is exposed every step -- used while the batch
// has no more groups than clusters (then more clusters run in parallel instead).
// HP: half precision (`TorchModel.half()`): fp16 operands without the hi/lo split -- one A row per
// window, so a group is 16 windows (4 per lane), 12 MFMAs per wave and step, W_hi only; a granule
// carries the fp16 h of TWO windows (2wp, 2wp+1), which makes the exchange byte-for-byte the same
// code as the (hi, lo) granules of the fp32-parity mode.
template <int PF, int NGRP, bool HP = false, int ABL = 0>
__global__ __launch_bounds__(512, 1) void k_lstm_wide(
    const float *__restrict__ gi,       // [B*T][1536] permuted gate columns, bias folded, PRE-SCALED by S
    const half8 *__restrict__ wfrag,    // [12 members][8 waves][12 ks][2 hi/lo][64]
    float *__restrict__ out,            // [B*T][384]
    unsigned long long *exch,           // [clusters][2 groups][2 parity][8 windows][384 units] granules + headers, zeroed
    int *status,                        // [0] != 0: a cluster timed out
    int B, int T, int reverse, float inv_scale, int n_clusters, int n_units, int force_wt, int poll_delay,
    int s0, int ns, float *__restrict__ cstate,   // scan steps [s0, s0 + ns); s0 > 0 resumes from the h this
                                                  // kernel stored at scan step s0 - 1 and the cell state in
                                                  // cstate [B][384], which every launch leaves behind
    int skip_if_lost)                             // synchronous forwards: return at once when status[0] is already up
{
    __shared__ __attribute__((aligned(16))) unsigned char img[2][2][kWImgBytes];   // [group][parity]
    __shared__ int s_abort[2];
    __shared__ int s_same;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int cluster = (idx / kWC) * 8 + xcd, member = idx % kWC;
    if (cluster >= n_clusters) return;
    // a cluster of an EARLIER launch of this forward timed out: the forward is lost and will be re-run or reported;
    // do not spend another handshake time-out on each of its remaining launches.  (Not in asynchronous mode, where
    // the flag of an earlier forward stays up until the caller asks with mdk_rl_check: later forwards must still run.)
    if (skip_if_lost && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    const int c = lane & 15, g = lane >> 4;   // accumulator: rows 4g..4g+3 = gates of unit g, column c
    constexpr int NS = HP ? 1 : 2;          // fp16 pieces per operand
    constexpr int GW = HP ? 16 : 8;         // windows per group: column c = window (HP) or 2*window + {hi, lo}
    const int wl = HP ? c : (c >> 1);       // this lane's window within the group
    const bool lead = HP || !(c & 1);       // fp32-parity: the hi column's lane finishes the cell

    half8 wf[kWKS][NS];
    {
        const half8 *wp = wfrag + ((size_t)(member * 8 + w8) * (kWKS * 2)) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < kWKS; ++ks)
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) wf[ks][sp] = wp[(size_t)(ks * 2 + sp) * 64];
    }
    constexpr float L2E = 1.44269504088896340736f;
    const float c_sig = -L2E * inv_scale, c_tanh = 2.0f * L2E * inv_scale;
    const int col = (member * 8 + w8) * 16 + 4 * g;  // permuted gi columns (i, f, g, o) of this lane's unit
    const int unit = 32 * member + 4 * w8 + g;
    unsigned long long *ex = exch + (size_t)cluster * (4 * kWGranules);
    if (tid < 2) s_abort[tid] = 0;

    // Do all 12 members share an XCD (= one L2)?  Then plain stores (kept in that L2) + L1-bypassing
    // loads are coherent and several times faster than write-through granules that every reader must
    // fetch from the fabric.  Placement is only OBSERVED to be block % 8, so the members tell each
    // other their XCC_ID through the always-valid write-through protocol first and all take the same
    // decision from the same 12 values.
    {
        unsigned long long *hdr = exch + (size_t)kWMaxClusters * (4 * kWGranules) + (size_t)cluster * 16;
        const unsigned int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;   // HW_REG_XCC_ID[3:0]
        if (tid == 0)
            __hip_atomic_store(hdr + member, (0x7fffffffull << 32) | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 64) {
            unsigned long long x = 0;
            const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
            bool ok;
            do {
                if (lane < kWC) x = __hip_atomic_load(hdr + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = lane >= kWC || (unsigned int)(x >> 32) == 0x7fffffffu;
                if (!__all(ok)) __builtin_amdgcn_s_sleep(4);
            } while (!__all(ok) && __builtin_amdgcn_s_memrealtime() - t_begin < kWHandshakeTicks);
            const bool same = lane >= kWC || ((unsigned int)x & 0xf) == xcc;
            if (lane == 0) s_same = (__all(ok) && __all(same)) ? 1 : (__all(ok) ? 0 : -1);
        }
        __syncthreads();
        if (s_same < 0) {
            if (tid == 0) atomicExch(status, 1);
            return;
        }
    }
    const bool same_xcd = s_same == 1 && !force_wt;

    // gather: 1536 granule PAIRS (units u, u+1 of one window) per step; thread t takes pairs
    // t, t+512, t+1024 with one 16-byte sc1 load each -- every load instruction of a wave covers
    // 1 KB of contiguous memory -- and writes the fp16 hi pair / lo pair with two 4-byte LDS stores.
    // (Each 8-byte half carries its own tag, so a torn 16-byte load is harmless.)
    int g_off[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int gidx = 2 * (tid + 512 * j);
        const int w = gidx / kWH, u = gidx % kWH;
        g_off[j] = (u >> 5) * kHKStride + ((u >> 3) & 3) * kHGroupStride + (2 * w) * 16 + (u & 7) * 2;
    }
    const int rd_off = g * kHGroupStride + c * 16;
    const long tstep = reverse ? -1 : 1;
    const int s_end = s0 + ns;
    const int t_first = reverse ? (T - 1 - s0) : s0;
    const long gstride = tstep * (long)kWG4, ostride = tstep * (long)kWH;

#pragma unroll
    for (int ks = 0; ks < kWKS; ++ks)
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) asm volatile("" ::"v"(wf[ks][sp]));

    unsigned int tag = 0;
    for (int it = cluster; it < n_units; it += n_clusters) {   // unit = NGRP consecutive 8-window groups
        const float *gp[2];
        float *op[2];
        bool wok[2];
        float cst[2];
        float4 gq[2][PF];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            int win = (NGRP * it + x) * GW + wl;
            wok[x] = win < B;
            if (!wok[x]) win = B - 1;
            gp[x] = gi + ((size_t)win * T + t_first) * kWG4 + col;
            op[x] = out + ((size_t)win * T + t_first) * kWH + unit;
            cst[x] = (s0 > 0 && x < NGRP) ? cstate[(size_t)win * kWH + unit] : 0.f;
        }
        __syncthreads();                                  // previous pair's images are dead
#pragma unroll
        for (int x = 0; x < 2; ++x) {   // the images the first step reads: h_0 = 0, or h of scan step s0 - 1
            if (s0 == 0 || x >= NGRP) {
                uint32_t *z = reinterpret_cast<uint32_t *>(img[x][tag & 1]);
                for (int i = tid; i < kWImgBytes / 4; i += 512) z[i] = 0u;
            } else {
                const long tprev = (long)t_first - tstep;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int gidx = 2 * (tid + 512 * j);
                    const int w = gidx / kWH, u = gidx % kWH;      // image rows 2w, 2w + 1; units u, u + 1
                    auto row_of = [&](int wi) {
                        int win = (NGRP * it + x) * GW + wi;
                        if (win >= B) win = B - 1;
                        return out + ((size_t)win * T + tprev) * kWH + u;
                    };
                    unsigned int r0, r1;
                    if constexpr (HP) {      // rows = windows 2w, 2w + 1
                        const float2 a = *reinterpret_cast<const float2 *>(row_of(2 * w));
                        const float2 b2 = *reinterpret_cast<const float2 *>(row_of(2 * w + 1));
                        r0 = (unsigned int)__builtin_bit_cast(unsigned short, (_Float16)(a.x * kActScale)) |
                             ((unsigned int)__builtin_bit_cast(unsigned short, (_Float16)(a.y * kActScale)) << 16);
                        r1 = (unsigned int)__builtin_bit_cast(unsigned short, (_Float16)(b2.x * kActScale)) |
                             ((unsigned int)__builtin_bit_cast(unsigned short, (_Float16)(b2.y * kActScale)) << 16);
                    } else {                 // rows = (window w, hi), (window w, lo)
                        const float2 a = *reinterpret_cast<const float2 *>(row_of(w));
                        _Float16 h0, l0, h1, l1;
                        split_f16(a.x * kActScale, h0, l0);
                        split_f16(a.y * kActScale, h1, l1);
                        r0 = (unsigned int)__builtin_bit_cast(unsigned short, h0) | ((unsigned int)__builtin_bit_cast(unsigned short, h1) << 16);
                        r1 = (unsigned int)__builtin_bit_cast(unsigned short, l0) | ((unsigned int)__builtin_bit_cast(unsigned short, l1) << 16);
                    }
                    *reinterpret_cast<unsigned int *>(img[x][tag & 1] + g_off[j]) = r0;
                    *reinterpret_cast<unsigned int *>(img[x][tag & 1] + g_off[j] + 16) = r1;
                }
            }
        }
        auto refill = [&](int x, int p, bool advance) {
            if constexpr (ABL & 1) gq[x][p] = make_float4(0.f, 0.f, 0.f, 0.f);
            else gq[x][p] = *reinterpret_cast<const float4 *>(gp[x]);
            if (advance) gp[x] += gstride;
        };
#pragma unroll
        for (int x = 0; x < 2; ++x) {
#pragma unroll
            for (int p = 0; p < PF; ++p) gq[x][p] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int p = 0; p + 1 < PF; ++p) refill(x, p, s0 + p + 1 < s_end);
        }
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int p = 0; p + 1 < PF; ++p) {
                asm volatile("" ::"v"(gq[x][p].x)); asm volatile("" ::"v"(gq[x][p].y));
                asm volatile("" ::"v"(gq[x][p].z)); asm volatile("" ::"v"(gq[x][p].w));
            }
        __syncthreads();

        // One half-step: compute + publish step `tag` of group x, and gather step `gtag` of the OTHER
        // group (published one half-step ago) into its next image.  Vector-memory issue order is
        // [gather loads] [gi refill] ... [publish + h stores] [wait gather]: the wait covers only the
        // gather loads (vmcnt retires in order) whose data arrived under the MFMAs; the refill and the
        // stores drain during the next half-step.  The barrier is LDS-only for the same reason.
        auto gather_issue = [&](int y, unsigned int gtag, uint4 (&v)[3]) {
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(ex + (size_t)(2 * y + (gtag & 1)) * kWGranules, 0,
                                                                kWGranules * 8, 0x00020000);
#pragma unroll
            for (int j = 0; j < 3; ++j)
                v[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (tid + 512 * j) * 16, 0, 16));
        };
        auto gather_finish = [&](int y, unsigned int gtag, uint4 (&v)[3]) {
            unsigned char *wb = img[y][gtag & 1];
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(ex + (size_t)(2 * y + (gtag & 1)) * kWGranules, 0,
                                                                kWGranules * 8, 0x00020000);
            int spins = 0;
            bool bad;
            do {
                bad = false;
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (!(ABL & 2) && (v[j].y != gtag || v[j].w != gtag)) {
                        bad = true;
                        v[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (tid + 512 * j) * 16, 0, 16));
                    }
                if (bad) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > kWSpinLimit) { s_abort[tag & 1] = 1; break; }
                }
            } while (bad);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                *reinterpret_cast<unsigned int *>(wb + g_off[j]) = (v[j].x & 0xffffu) | (v[j].z << 16);
                *reinterpret_cast<unsigned int *>(wb + g_off[j] + 16) = (v[j].x >> 16) | (v[j].z & 0xffff0000u);
            }
        };
        auto half_step = [&](int x, int p, int step, bool do_gather, unsigned int gtag) {
            const unsigned char *rb = img[x][(tag - 1) & 1];
            // the other group published early in the previous half-step: its granules are in L2 by now
            uint4 v[3];
            if constexpr (NGRP == 2 && !(ABL & 4)) { if (do_gather) gather_issue(1 - x, gtag, v); }
            // gi prefetch is issued AFTER the gather loads: vmcnt retires in order, so this half-step's
            // gather wait does not include it and it has until the next half-step's to arrive
            if constexpr (NGRP == 2) refill(x, (p + PF - 1) % PF, (step + PF) < s_end);    // the slot consumed one step ago
            __builtin_amdgcn_sched_barrier(0);
            floatx4 acc0 = floatx4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
            for (int ks = 0; ks < kWKS; ks += 2) {
                const half8 a0 = *reinterpret_cast<const half8 *>(rb + ks * kHKStride + rd_off);
                const half8 a1 = *reinterpret_cast<const half8 *>(rb + (ks + 1) * kHKStride + rd_off);
                acc0 = mfma16(wf[ks][0], a0, acc0);          // A = W (rows = gate columns), B = h (columns = windows)
                acc1 = mfma16(wf[ks + 1][0], a1, acc1);
                if constexpr (!HP) {
                    acc0 = mfma16(wf[ks][1], a0, acc0);
                    acc1 = mfma16(wf[ks + 1][1], a1, acc1);
                }
            }
            // acc[r] = gate r (i, f, g, o) of unit g for column c; fp32-parity: add the lo column (lane c ^ 1)
            float pre[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float dot = acc0[r] + acc1[r];
                if constexpr (!HP) dot += dpp_mov<0xB1>(dot);     // quad_perm:[1,0,3,2]
                pre[r] = dot;
            }
            const float4 gv4 = gq[x][p];
            const float iv = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((pre[0] + gv4.x) * c_sig));
            const float fv = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((pre[1] + gv4.y) * c_sig));
            const float gg = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((pre[2] + gv4.z) * c_tanh)), 1.0f);
            const float ov = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((pre[3] + gv4.w) * c_sig));
            const float cv = __builtin_fmaf(fv, cst[x], iv * gg);
            if (step < s_end) cst[x] = cv;        // (padding steps must not disturb the state a later launch resumes from)
            const float tc = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(cv * (2.0f * L2E))), 1.0f);
            const float h = ov * tc;
            unsigned int payload;
            if constexpr (HP) {   // a granule carries windows (2wp, 2wp + 1): take the odd neighbour's half
                const unsigned int hb = __builtin_bit_cast(unsigned short, (_Float16)(h * kActScale));
                const unsigned int nb = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)hb, 0xB1, 0xf, 0xf, true);
                payload = hb | (nb << 16);
            } else {
                _Float16 hi, lo;
                split_f16(h * kActScale, hi, lo);
                payload = (unsigned int)__builtin_bit_cast(unsigned short, hi) |
                          ((unsigned int)__builtin_bit_cast(unsigned short, lo) << 16);
            }
            unsigned long long *dst = ex + (size_t)(2 * x + (tag & 1)) * kWGranules;
            if (!(c & 1)) {       // granule row c >> 1: (window, hi|lo) or a pair of windows
                const unsigned long long gran = ((unsigned long long)tag << 32) | payload;
                if constexpr (!(ABL & 4)) {
                    if (same_xcd)
                        __hip_atomic_store(dst + (c >> 1) * kWH + unit, gran, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);   // plain store: stays in the shared L2
                    else
                        __hip_atomic_store(dst + (c >> 1) * kWH + unit, gran, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);       // write-through (sc1)
                }
            }
            if (lead) {
                if constexpr (!(ABL & 8)) { if (step < s_end && wok[x]) op[x][0] = h; }
            }
            op[x] += ostride;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NGRP == 1 && !(ABL & 4)) {
                // own group, just published: a poll that misses costs a second L2 round trip, so give the
                // other members' stores time to land first
                for (int i = 0; i < poll_delay; ++i) __builtin_amdgcn_s_sleep(1);
                gather_issue(x, gtag, v);
            }
            if constexpr (!(ABL & 4)) { if (do_gather) gather_finish(NGRP == 2 ? 1 - x : x, gtag, v); }
            if constexpr (NGRP == 1) refill(x, (p + PF - 1) % PF, (step + PF) < s_end);
            lds_barrier();
        };

        for (int step0 = s0; step0 < s_end; step0 += PF) {
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                const int step = step0 + p;      // steps >= s_end run too (stores masked): all members agree
                ++tag;
                if constexpr (NGRP == 2) {
                    half_step(0, p, step, step > s0, tag - 1);
                    half_step(1, p, step, true, tag);
                } else {
                    half_step(0, p, step, true, tag);
                }
                if (s_abort[tag & 1]) {
                    if (tid == 0) atomicExch(status, 1);
                    return;
                }
            }
        }
        if constexpr (NGRP == 2 && !(ABL & 4)) {   // B's last step: keeps "nobody publishes t+2 before everybody gathered t" across pairs
            uint4 v[3];
            gather_issue(1, tag, v);
            gather_finish(1, tag, v);
        }
        __syncthreads();
        if (s_abort[0] | s_abort[1]) {
            if (tid == 0) atomicExch(status, 1);
            return;
        }
#pragma unroll
        for (int x = 0; x < NGRP; ++x)
            if (lead && wok[x]) cstate[(size_t)((NGRP * it + x) * GW + wl) * kWH + unit] = cst[x];
    }
}

// gi = (A W^T) * alpha + bias for the wide LSTM: A fp32 [M][32*KS] natural rows, W pre-packed
// as fp16 hi/lo B-fragments in the PERMUTED column order k_lstm_wide reads ([96 tiles][KS][2][64]),
// fp16x2 split with three products, fp32 accumulate.  Work-group = 64 rows (one contiguous run of
// A) x all 1536 columns: A is converted once into LDS (hi and lo images, 16-byte fragments), the
// 8 waves walk 12 column tiles each in chunks of 3 with B streaming from L2.
constexpr int kWGemmRows = 64;
constexpr int kWGemmBlk = kWGemmRows * 16 + 16;   // one (k-step, lane-group) block of an image + pad

template <int KS, bool HP = false>   // HP: one fp16 product, hi image only
__global__ __launch_bounds__(512, 1) void k_gemm_rows(
    const float *__restrict__ A, const half8 *__restrict__ wfrag, const float *__restrict__ bias,
    float *__restrict__ out, int T, int t_begin, int t_len, float a_scale, float alpha)   // rows (window blockIdx.y,
                                                                                            // t in [t_begin, t_begin + t_len))
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int K = 32 * KS;
    constexpr int IMG = KS * 4 * kWGemmBlk;
    unsigned char *ahi = lds, *alo = lds + IMG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const long row0 = (long)blockIdx.y * T + t_begin + (long)blockIdx.x * kWGemmRows;
    const long M = (long)blockIdx.y * T + t_begin + t_len;   // first row beyond this window's range

    for (int it = tid; it < kWGemmRows * 4 * KS; it += 512) {
        const int row = it / (4 * KS), k8 = it % (4 * KS);
        float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0;
        if (row0 + row < M) {
            const float4 *src = reinterpret_cast<const float4 *>(A + (size_t)(row0 + row) * K + k8 * 8);
            x0 = src[0]; x1 = src[1];
        }
        const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        half8 hi, lo;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            _Float16 a, b;
            split_f16(xv[i] * a_scale, a, b);
            hi[i] = a; lo[i] = b;
        }
        *reinterpret_cast<half8 *>(ahi + k8 * kWGemmBlk + row * 16) = hi;
        if constexpr (!HP) *reinterpret_cast<half8 *>(alo + k8 * kWGemmBlk + row * 16) = lo;
    }
    __syncthreads();

#pragma unroll 1
    for (int chunk = 0; chunk < 4; ++chunk) {
        const int nt0 = w8 * 12 + chunk * 3;
        floatx4 acc[4][3];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[mt][j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int ks = 0; ks < KS; ++ks) {
            half8 bh[3], bl[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const half8 *wp = wfrag + (((size_t)(nt0 + j) * KS + ks) * 2) * 64 + lane;
                bh[j] = wp[0];
                if constexpr (!HP) bl[j] = wp[64];
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int off = (ks * 4 + g) * kWGemmBlk + (mt * 16 + c) * 16;
                const half8 ah = *reinterpret_cast<const half8 *>(ahi + off);
                half8 al;
                if constexpr (!HP) al = *reinterpret_cast<const half8 *>(alo + off);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    acc[mt][j] = mfma16(ah, bh[j], acc[mt][j]);
                    if constexpr (!HP) {
                        acc[mt][j] = mfma16(al, bh[j], acc[mt][j]);
                        acc[mt][j] = mfma16(ah, bl[j], acc[mt][j]);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int colj = (nt0 + j) * 16 + c;
            const float bv = bias[colj];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long row = row0 + mt * 16 + 4 * g + r;
                    if (row < M) out[(size_t)row * kWG4 + colj] = __builtin_fmaf(acc[mt][j][r], alpha, bv);
                }
        }
    }
}

}  // namespace mdk
