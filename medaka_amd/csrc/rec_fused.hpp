// GRU layer >= 1 with its input projection FUSED into the recurrence: gi never goes to HBM.
//
// Unfused (gi_proj.hpp + rec_mfma.hpp) a layer with K = 256 inputs writes gi = x W_ih^T + b as fp32 -- 3072 bytes per
// column and direction pair -- and the recurrence reads it back: 13.9 of the 24.6 GB a 200 x 10000 forward moves
// (profiles/traffic.json, round 3), and 2.7 ms of projection GEMM in front of a 2.2 ms recurrence.  In the throughput
// regime (split scan, or batches that fill the chip by themselves) every CU holds one recurrence work-group whose
// matrix pipe idles about half of every step, and the GEMM's own work-group shape is already the recurrence's:
// (8 windows x 8 steps) x 384 gate columns of ONE direction, with the accumulator element of lane (g, c), register
// 2q + tt of row-tile mt being exactly the pre-activation that lane consumes at step 2 mt + tt for window 2g + q
// (gi_proj.hpp epilogue == rec_mfma.hpp `gp[q]`).  So this kernel alternates, per strip of 8 scan steps:
//
//   projection phase   the strip's 64 rows x K of previous-layer activations sit in LDS as fp16 hi/lo A-fragments
//                      (64 KB, staged during the PREVIOUS strip's steps); every wave runs the k_gi_gemm inner loop for
//                      its 16 hidden units x 3 gates of this direction: 8 k-steps x 4 row-tiles x 3 gates x 3 split
//                      products = 288 MFMAs back to back, W_ih fragments streaming from L2 one k-step ahead;
//                      scale + folded bias applied in place: 48 accumulator registers per lane = the strip's gi;
//   recurrence phase   8 steps of the k_rec_mfma step (W_hh in registers, h image in LDS, one barrier per step) taking
//                      gi from those registers; under steps 0..3 each thread requests its 4 pieces (8 floats) of the
//                      NEXT strip's activations, under steps 2..5 it splits them to fp16 hi/lo and stores them in LDS.
//
// Same MFMAs in the same order on the same operands as the unfused pair, same fmaf for scale + bias, same gate
// arithmetic: the results are BIT-IDENTICAL to k_gi_gemm + k_rec_mfma (tests/test_parity_gpu.py
// ::test_fused_projection_agrees_bitwise), so every parity result of the unfused path carries over.
// HBM per column and direction: 1024 B of activations in (both input directions) + 512 B of h out, against
// 512 + 1536 (GEMM, its share) + 1536 + 512 (recurrence).
// fp32-parity mode (fp16 hi/lo split), 8-window work-groups, GRU cell; T, s0, ns multiples of 8.
//
// HEAD (last layer): the classifier's Linear(D*128 -> 5) is fused as well.  The fp16 hi/lo image of h_t that every step
// publishes in LDS for the next step IS the A operand of  logits_d[t] = h_d[t] W_lin[:, d*128 : d*128+128]^T ; with HEAD the
// images of a strip's 8 steps stay in an 8-slot ring, and at the start of the next projection phase wave j multiplies
// step j's image by the 8 W_lin fragments (4 k-steps x hi/lo: 8 MFMAs per wave and strip against 288 of projection) and
// stores this direction's 5 partial logits of its 8 windows (160 bytes per (tile, t, direction)).  k_head_combine
// (head.hpp) adds the two directions and the bias and takes the softmax: a 40-us kernel instead of the 0.41 ms
// k_head_tiled pass over the activations.  (Not small enough to run BESIDE a recurrence that holds every CU, though: launched
// next to one it finishes when the recurrence does, profiles/r4_experiments/README.md -- hence HEAD = 2 below.)
// Same fp16x2 split as everywhere (h hi+lo times W hi+lo, fp32 accumulate): logits agree with the fp32 FMA head to ~1e-7
// relative, not bit for bit.
#pragma once
#include "common.hpp"
#include "gi_proj.hpp"
#include "layout.hpp"
#include "rec_mfma.hpp"

namespace mdk {

#ifndef MDK_FIN_REQ
#define MDK_FIN_REQ 4      // HEAD = 2: the step of a strip under which the other direction's partial logits are requested
#endif
constexpr int kFusedSteps = 8;                         // scan steps per strip = rows 2 mt + tt of 4 MFMA row-tiles
constexpr int kFusedMT = kFusedSteps / 2;
__host__ __device__ inline constexpr size_t fused_lds_bytes(int KSTEPS, bool hp = false) { return (size_t)(hp ? 1 : 2) * kFusedMT * KSTEPS * 64 * 16; }

// HP: half-precision mode (`model.half()`): fp16 operands without the hi/lo split -- one product in the projection, one row
// per window in the recurrence (rows 4g + q of the A image), W_hi only; again bit-identical to the unfused HP pair.
//
// HEAD = 2 (the launches of the second half of a bidirectional scan, or every launch of a one-directional one): the column a
// step finishes is COMPLETE -- the other direction left its partial logits there in an earlier launch -- so the wave adds
// them, the bias, takes the softmax (the arithmetic of k_head_combine, operation for operation: same bits) and stores the
// 5 probabilities where the caller wants them ((B, T, 5); a split scan's chunk: its own columns of the real window).  No
// head kernel at all then, and -- the point -- finished columns can leave for the host by DMA while the scan runs on.
template <int KSTEPS, int HEAD, bool HP = false>   // K = 32 * KSTEPS = DIN * 128 input features
__global__ __launch_bounds__(512, 2) void k_rec_fused(
    const float *__restrict__ act_in,   // act_t of the previous layer (|x| < 1)
    const half8 *__restrict__ wihfrag,  // [D][8 waves][KSTEPS][3 gates][2 hi/lo][64 lanes]   (as k_gi_gemm)
    const float *__restrict__ bias,     // [D][384] folded bias                              (as k_gi_gemm)
    const half8 *__restrict__ wfrag,    // W_hh B-fragments [D][8][4][3][2][64]              (as k_rec_mfma)
    const float *__restrict__ b_hn,     // [D][128]
    float *__restrict__ out,            // act_t of this layer
    int n_tiles, int T, int D,
    const float *__restrict__ inv_scale_rec_p, const float *__restrict__ inv_scale_gi_p,
    const float *__restrict__ up_scale_rec_p, float a_scale,
    int reverse_mask, int s0, int ns,
    const half8 *__restrict__ wlin_frag,   // HEAD: W_lin B-fragments [D][4 ksteps][2 hi/lo][64 lanes] (column n = class, n >= 5 zero)
    float lin_inv_scale,                   // HEAD: 1 / (kActScale * W_lin's operand scale)
    float *__restrict__ lpart,             // HEAD: partial logits [D][n_tiles][T][8 windows][5]
    const float *__restrict__ lin_b,       // HEAD = 2: classifier bias [5]
    float *__restrict__ probs,             // HEAD = 2: the result, (nb, T, 5) or the split plan's (B, T, 5)
    int nb, int normalise, SplitPlan spl)  // HEAD = 2: windows of this pass; softmax or raw logits; spl.S > 1: split scan
{
    constexpr bool FIN = HEAD == 2;
    constexpr int DIN = KSTEPS / 4;
    constexpr int MT = kFusedMT;
    constexpr int NPIECE = kFusedSteps * DIN * 128 / 512;   // pieces (two 16-byte quads) per thread and strip
    static_assert(NPIECE == 2 || NPIECE == 4, "staging schedule below assumes 2 or 4 pieces per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half8 *xs = reinterpret_cast<half8 *>(smem);            // [split 2][mt MT][KSTEPS][64 lanes]
    constexpr int NIMG = HEAD ? 8 : 2;      // images of h kept: the step's two, or a whole strip's for the head
    __shared__ __attribute__((aligned(16))) unsigned char hbuf[NIMG * kHBufBytes];
    struct FinRow { long base; int lo, hi; };        // HEAD = 2: window w of the tile delivers local columns [lo, hi) to probs + base + 5 t
    __shared__ FinRow ftab[FIN ? kTileWin : 1];
    // HEAD: this direction's 8 W_lin fragments -- every wave multiplies its step's image by the same ones, once per strip;
    // from LDS instead of eight L2 round trips at the top of every projection phase
    __shared__ half8 wlds[HEAD ? 512 : 1];
    __builtin_amdgcn_s_setprio(MDK_REC_PRIO);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x;
    const int d = blockIdx.y;
    const int c = lane & 15;
    const int g = lane >> 4;
    const bool reverse = (reverse_mask >> d) & 1;
    const float inv_scale = inv_scale_rec_p[d];
    const float c_sig = -inv_scale * 1.44269504088896340736f;
    const float c_tanh = 2.0f * inv_scale * 1.44269504088896340736f;

    constexpr int NS = HP ? 1 : 2;          // fp16 pieces per operand
    half8 wf[4][3][NS];
    {
        const half8 *wp = wfrag + ((size_t)(d * 8 + w8) * 24) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int gate = 0; gate < 3; ++gate)
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) wf[ks][gate][sp] = wp[(size_t)((ks * 3 + gate) * 2 + sp) * 64];
    }
    for (int i = tid; i < NIMG * kHBufBytes / 4; i += 512) reinterpret_cast<uint32_t *>(hbuf)[i] = 0u;
    if constexpr (HEAD != 0) wlds[tid] = wlin_frag[(size_t)d * 8 * 64 + tid];
    if constexpr (FIN) {
        if (tid < kTileWin) {
            const int win = tile * kTileWin + tid;
            FinRow r{0, 0, 0};
            if (win < nb) {
                if (spl.S > 1) {
                    const int k = win / spl.B;
                    r.lo = spl.core0[k] - spl.start[k];
                    r.hi = spl.core0[k + 1] - spl.start[k];
                    r.base = ((long)(win - k * spl.B) * spl.T + spl.start[k]) * 5;
                } else {
                    r.hi = T;
                    r.base = (long)win * T * 5;
                }
            }
            ftab[tid] = r;
        }
    }

    const int u = 16 * w8 + c;
    const float bhn = b_hn[d * kH + u] * (1.0f / inv_scale);
    // projection epilogue constants (gi_proj.hpp): gi = acc * (inv_scale_gi * os) + bias * os, os = the recurrence's scale
    const float os = up_scale_rec_p[d];
    const float gi_scale = inv_scale_gi_p[d] * os;
    float bv[3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) bv[nt] = bias[(size_t)d * kG + nt * kH + u] * os;

    const int s_end = s0 + ns;
    // Every HBM address of the scan in the buffer form (common.hpp): wave-uniform base + one lane offset + a scalar that
    // follows the step -- no vector instruction per access in the step's in-order stream.
    // This layer's output: the block of column t at t * D * 4096 bytes, this lane's two values 256 bytes apart.
    const __amdgpu_buffer_rsrc_t orsrc = make_rsrc(out + act_block(D, tile, T, 0));
    const unsigned ovoff = (unsigned)act_in_block(d, w8, 0, lane) * 4u;
    const int obytes = D * 4096;
    auto ocol = [&](int s) { return (unsigned)((reverse ? (T - 1 - s) : s) * obytes); };     // scan step -> byte offset of its column
    float hprev[2] = {0.f, 0.f};

    const int rd_off = g * kHGroupStride + c * 16;
    const int wr_off = (w8 >> 1) * kHKStride + (2 * (w8 & 1) + (c >> 3)) * kHGroupStride + (4 * g) * 16 + (c & 7) * 2;

    // ---- staging of a strip's activations.  The image is the k_gi_gemm one (scan step tau of the strip is row 2 mt + tt = tau
    // of the M-tile whatever the direction: a reversed scan stages its columns in descending order); the REQUESTS are laid
    // out for the memory pipe: a wave's request is one contiguous KB -- lane l takes the 16-byte quad tid of a step's block --
    // instead of the two halves of a lane's own 32-byte piece (every request then touched all 16 lines of a 2 KB span and
    // used half of each; the eight requests per thread and strip cost 9 % of the kernel, profiles/r5_experiments/README.md).
    // A quad is 4 consecutive units of one window: half of one 8-slot k-group of its A-fragment row (an 8-byte LDS store).
    const __amdgpu_buffer_rsrc_t irsrc = make_rsrc(act_in + act_block(DIN, tile, T, 0));
    struct Piece { floatx4 v0, v1; };
    constexpr int CPS = DIN * 256;                  // 16-byte quads of one column's activation block
    auto quad_pos = [&](int it, int l, int &tau, int &c9) {
        const int C = (it * 2 + l) * 512 + tid;
        tau = C / CPS; c9 = C % CPS;
    };
    auto piece_load = [&](int strip, int it) {
        Piece p;
        int tau, c9;
        quad_pos(it, 0, tau, c9);
        { const int s = strip * kFusedSteps + tau; const int t = reverse ? (T - 1 - s) : s;
          p.v0 = buf_load_floatx4(irsrc, (unsigned)c9 * 16u, (unsigned)(t * (DIN * 4096))); }
        quad_pos(it, 1, tau, c9);
        { const int s = strip * kFusedSteps + tau; const int t = reverse ? (T - 1 - s) : s;
          p.v1 = buf_load_floatx4(irsrc, (unsigned)c9 * 16u, (unsigned)(t * (DIN * 4096))); }
        return p;
    };
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    auto piece_store = [&](int it, Piece p) {
        // the request stays in flight until HERE: without this the compiler multiplies by a_scale right behind the load
        // and waits for it in the step that issued it (seen in the ISA: vmcnt(0) under the next MFMAs)
        asm volatile("" : "+v"(p.v0), "+v"(p.v1));
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            int tau, c9;
            quad_pos(it, l, tau, c9);
            const floatx4 v = l ? p.v1 : p.v0;
            // block offset 4 c9 floats = act_in_block(chunk, q, lane' = 16 g + c): units c .. c + 3 of window 2 g + q
            const int cl = (c9 & 3) * 4, gg = (c9 >> 2) & 3, qq = (c9 >> 4) & 1, chunk = c9 >> 5;
            half4 hi, lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                _Float16 a, b;
                split_f16(v[i] * a_scale, a, b);
                hi[i] = a; lo[i] = b;
            }
            const int row = 4 * gg + 2 * qq + (tau & 1), mt = tau >> 1;
            const int k8 = chunk * 2 + (cl >> 3), ks = k8 >> 2;
            const int slot = (k8 & 3) * 16 + row;
            const int hb = (cl >> 2) & 1;                  // which half of the row's 8 k-slots
            *reinterpret_cast<half4 *>(reinterpret_cast<unsigned char *>(&xs[((0 * MT + mt) * KSTEPS + ks) * 64 + slot]) + 8 * hb) = hi;
            if constexpr (!HP)
                *reinterpret_cast<half4 *>(reinterpret_cast<unsigned char *>(&xs[((1 * MT + mt) * KSTEPS + ks) * 64 + slot]) + 8 * hb) = lo;
        }
    };

    const int strip0 = s0 / kFusedSteps, strip1 = s_end / kFusedSteps;
#pragma unroll
    for (int it = 0; it < NPIECE; ++it) piece_store(it, piece_load(strip0, it));

#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int gate = 0; gate < 3; ++gate)
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) asm volatile("" ::"v"(wf[ks][gate][sp]));
    asm volatile("" ::"v"(bhn));
    __syncthreads();
    if (s0 > 0) {   // resume: h of scan step s0 - 1 from the output, and its fp16 image (as k_rec_mfma)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float h = buf_load_float(orsrc, ovoff + q * 256, ocol(s0 - 1));
            hprev[q] = h;
            _Float16 hi, lo;
            split_f16(h * kActScale, hi, lo);
            unsigned char *img = hbuf + (s0 & (NIMG - 1)) * kHBufBytes + wr_off;
            if constexpr (HP) {
                *reinterpret_cast<_Float16 *>(img + q * 16) = hi;
            } else {
                *reinterpret_cast<_Float16 *>(img + (2 * q) * 16) = hi;
                *reinterpret_cast<_Float16 *>(img + (2 * q + 1) * 16) = lo;
            }
        }
        __syncthreads();
    }

    const half8 *wp = wihfrag + ((size_t)(d * 8 + w8) * KSTEPS) * 6 * 64 + lane;

    // HEAD: partial logits of strip `hs`, one scan step per wave (the image of h after step s sits in slot (s + 1) % 8)
    // HEAD = 2 works on both windows of a lane at once: lanes 16 g + 0..4 finish window 2g, lanes 16 g + 8..12 window 2g + 1
    // (one exp / quotient / store sequence instead of two: the sequence is ~200 VALU instructions at 4 cycles each, on
    // every wave at the same moment)
    const int cq = c & 7, qsel = c >> 3;
    const int cc = cq < 5 ? cq : 4;
    // the other direction's partial logits of this wave's column, requested under step MDK_FIN_REQ of the strip (nothing
    // else is in flight then) and used at the top of the next one
    const __amdgpu_buffer_rsrc_t lorsrc = make_rsrc(lpart + ((size_t)(D - 1 - d) * n_tiles + tile) * T * 40);   // the other direction's partial logits
    const __amdgpu_buffer_rsrc_t lprsrc = make_rsrc(lpart + ((size_t)d * n_tiles + tile) * T * 40);             // this direction's
    const unsigned lovoff = (unsigned)(((2 * g + qsel) * 5 + cc) * 4);
    float oth = 0.f;
    float lbs[5] = {0.f, 0.f, 0.f, 0.f, 0.f};          // the classifier bias: wave-uniform, lives in scalar registers
    if constexpr (FIN) {
#pragma unroll
        for (int cl = 0; cl < 5; ++cl) lbs[cl] = lin_b[cl];
    }
    auto head_strip = [&](int hs) {
        const int s = hs * kFusedSteps + w8;
        const int t = reverse ? (T - 1 - s) : s;
        const unsigned char *img = hbuf + ((w8 + 1) & 7) * kHBufBytes + rd_off;
        const half8 *wl = wlds + lane;
        floatx4 la = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const half8 a = *reinterpret_cast<const half8 *>(img + ks * kHKStride);
            la = mfma16(a, wl[(ks * 2 + 0) * 64], la);
            if constexpr (!HP) la = mfma16(a, wl[(ks * 2 + 1) * 64], la);
        }
        float own[2];
        if constexpr (HP) {
            own[0] = la[0] * lin_inv_scale;                // rows 4g, 4g + 1 = windows 2g, 2g + 1
            own[1] = la[1] * lin_inv_scale;
        } else {
            own[0] = (la[0] + la[1]) * lin_inv_scale;      // window 2g:     hi row + lo row
            own[1] = (la[2] + la[3]) * lin_inv_scale;      // window 2g + 1
        }
        if constexpr (!FIN) {
            if (c < 5) {
                const unsigned vo = (unsigned)(((2 * g) * 5 + c) * 4);
                buf_store_float(own[0], lprsrc, vo, (unsigned)(t * 160));
                buf_store_float(own[1], lprsrc, vo + 20, (unsigned)(t * 160));
            }
        } else {
            // k_head_combine's arithmetic (head.hpp), operation for operation: (part_0 + part_1) + bias; largest; exp; the
            // five terms summed in class order; quotient.  Window 2g + 1's logits move to lanes 8..12 of the group first.
            const float up = __shfl(own[1], lane - 8);
            float v = qsel ? up : own[0];
            if constexpr (DIN == 2) v = d == 0 ? v + oth : oth + v;
            float a[5];
#pragma unroll
            for (int cl = 0; cl < 5; ++cl) a[cl] = __shfl(v, (lane & 56) + cl) + lbs[cl];
            v = a[0];
#pragma unroll
            for (int cl = 1; cl < 5; ++cl) v = cq == cl ? a[cl] : v;
            float res = v;
            if (normalise) {
                float mx = a[0];
#pragma unroll
                for (int cl = 1; cl < 5; ++cl) mx = fmaxf(mx, a[cl]);
                float sum = 0.f;
#pragma unroll
                for (int cl = 0; cl < 5; ++cl) sum += __expf(a[cl] - mx);
                res = __expf(v - mx) / sum;
            }
            const FinRow r = ftab[2 * g + qsel];
            if (cq < 5 && t >= r.lo && t < r.hi) probs[r.base + (long)t * 5 + cq] = res;
        }
    };
    auto head_request = [&](int hs) {
        if constexpr (FIN) {
            if constexpr (DIN == 2) {
                const int s = hs * kFusedSteps + w8;
                const int t = reverse ? (T - 1 - s) : s;
                oth = buf_load_float(lorsrc, lovoff, (unsigned)(t * 160));
            }
        }
    };

    for (int strip = strip0; strip < strip1; ++strip) {
        if constexpr (HEAD) { if (strip > strip0) head_strip(strip - 1); }
        // ================= projection phase: gi of this strip into acc (k_gi_gemm inner loop, this direction only)
        floatx4 acc[MT][3];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) acc[mt][nt] = floatx4{0.f, 0.f, 0.f, 0.f};
        if constexpr (HP) {
            // half precision: 12 MFMAs per k-step (192 cycles) cannot cover an L2 round trip, and there are registers to
            // spare (no lo halves anywhere): the W_ih fragments of the next THREE k-steps are on their way while one is
            // multiplied -- four register sets, four k-steps per trip so that the set index is static, requests
            // unconditional (past the end the last k-step is requested again: a branch would cost vmcnt(0))
            static_assert(KSTEPS % 4 == 0, "four k-steps per trip");
            half8 bq[4][3];
            auto load_b = [&](int ks, int set) {
                const int k = ks < KSTEPS ? ks : KSTEPS - 1;
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) bq[set][nt] = wp[(size_t)((k * 3 + nt) * 2 + 0) * 64];
            };
            auto kstep = [&](int ks, int set) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const half8 ah = xs[((0 * MT + mt) * KSTEPS + ks) * 64 + lane];
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt) acc[mt][nt] = mfma16(ah, bq[set][nt], acc[mt][nt]);
                }
            };
            load_b(0, 0); load_b(1, 1); load_b(2, 2);
#pragma unroll 1
            for (int ks = 0; ks < KSTEPS; ks += 4) {
                load_b(ks + 3, 3);
                __builtin_amdgcn_sched_barrier(0);      // (hipcc otherwise sinks the requests to where they are consumed)
                kstep(ks, 0);
                __builtin_amdgcn_sched_barrier(0);
                load_b(ks + 4, 0);
                __builtin_amdgcn_sched_barrier(0);
                kstep(ks + 1, 1);
                __builtin_amdgcn_sched_barrier(0);
                load_b(ks + 5, 1);
                __builtin_amdgcn_sched_barrier(0);
                kstep(ks + 2, 2);
                __builtin_amdgcn_sched_barrier(0);
                load_b(ks + 6, 2);
                __builtin_amdgcn_sched_barrier(0);
                kstep(ks + 3, 3);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // (one register set for the W_ih fragments: the unfused GEMM double-buffers them, here W_hh's 96 registers
            // and the strip's 48 accumulators leave no room -- the SIMD's other wave covers the L2 round trip)
            half8 bh[3], bl[3];
#pragma unroll 1
            for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) {
                    bh[nt] = wp[(size_t)((ks * 3 + nt) * 2 + 0) * 64];
                    bl[nt] = wp[(size_t)((ks * 3 + nt) * 2 + 1) * 64];
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const half8 ah = xs[((0 * MT + mt) * KSTEPS + ks) * 64 + lane];
                    const half8 al = xs[((1 * MT + mt) * KSTEPS + ks) * 64 + lane];
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt) {
                        acc[mt][nt] = mfma16(ah, bh[nt], acc[mt][nt]);
                        acc[mt][nt] = mfma16(al, bh[nt], acc[mt][nt]);
                        acc[mt][nt] = mfma16(ah, bl[nt], acc[mt][nt]);
                    }
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mt][nt][r] = fmaf(acc[mt][nt][r], gi_scale, bv[nt]);
        __syncthreads();      // every wave has read its A fragments: the image may be overwritten from here on

        // ================= recurrence phase: 8 steps on the strip's gi; the next strip's activations arrive under them
        // (branch-free on purpose: behind a conditional request hipcc's waitcnt pass falls back to vmcnt(0) at the store --
        // one HBM round trip per step.  After the last strip the current one is staged again, into an image nobody reads.)
        const int nstrip = strip + 1 < strip1 ? strip + 1 : strip;
        Piece pc[NPIECE];
#pragma unroll
        for (int j = 0; j < kFusedSteps; ++j) {
            const int step = strip * kFusedSteps + j;
            const int cur = (step & (NIMG - 1)) * kHBufBytes;
            const int nxt = ((step + 1) & (NIMG - 1)) * kHBufBytes;
            constexpr int kIssue = NPIECE == 4 ? 1 : 2;      // a piece is requested every kIssue steps, stored 2 steps later
            half8 a[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a[ks] = *reinterpret_cast<const half8 *>(hbuf + cur + ks * kHKStride + rd_off);
            floatx4 ar = floatx4{0.f, 0.f, 0.f, 0.f}, az = ar, anh = ar, anl = ar;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) {
                    ar = mfma16(a[ks], wf[ks][0][sp], ar);
                    az = mfma16(a[ks], wf[ks][1][sp], az);
                }
            }
            if (j % kIssue == 0 && j / kIssue < NPIECE) pc[j / kIssue] = piece_load(nstrip, j / kIssue);
            if (j == MDK_FIN_REQ) head_request(strip);
            // deferred store of the previous step's h (rec_mfma.hpp DS), unconditional: the first step of a launch writes its
            // incoming state (zero, or the resumed h) into its OWN slot, which the next step's store then overwrites
            const unsigned so_prev = ocol(step > s0 ? step - 1 : step);
#pragma unroll
            // (non-temporal: nobody reads h before the next layer, and 1 KB per column of it would otherwise pass through the L2
            // the W_ih fragments want to stay in -- 6.40 -> 6.29 ms per forward, profiles/r5_experiments/README.md)
            for (int q = 0; q < 2; ++q) buf_store_float<2>(hprev[q], orsrc, ovoff + q * 256, so_prev);
            // the piece requested two steps ago has arrived by now: split it and put it into the image (under the MFMAs)
            // (two steps of flight are enough: three or four change nothing -- profiles/r5_experiments/README.md)
            if (j >= 2 && (j - 2) % kIssue == 0 && (j - 2) / kIssue < NPIECE) piece_store((j - 2) / kIssue, pc[(j - 2) / kIssue]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                anh = mfma16(a[ks], wf[ks][2][0], anh);
                if constexpr (!HP) anl = mfma16(a[ks], wf[ks][2][1], anl);
            }
            float rr[2], zz[2], gnv[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = 2 * q + (j & 1);
                const float gr = acc[j >> 1][0][r], gz = acc[j >> 1][1][r];
                gnv[q] = acc[j >> 1][2][r];
                const float tr = gr + (HP ? ar[q] : (ar[2 * q] + ar[2 * q + 1]));
                const float tz = gz + (HP ? az[q] : (az[2 * q] + az[2 * q + 1]));
                rr[q] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(tr * c_sig));
                zz[q] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(tz * c_sig));
            }
#pragma unroll
            for (int i = 0; i < 4 * NS; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // 2 VALU
            }
            __builtin_amdgcn_sched_barrier(0);
            float hn[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float tn;
                if constexpr (HP) tn = anh[q] + bhn;
                else tn = ((anh[2 * q] + anl[2 * q]) + (anh[2 * q + 1] + anl[2 * q + 1])) + bhn;
                const float an = __builtin_fmaf(rr[q], tn, gnv[q]);
                const float e = __builtin_amdgcn_exp2f(an * c_tanh);
                const float n = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
                const float h = __builtin_fmaf(zz[q], hprev[q] - n, n);
                hprev[q] = h;
                hn[q] = h;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                _Float16 hi, lo;
                split_f16(hn[q] * kActScale, hi, lo);
                if constexpr (HP) {
                    *reinterpret_cast<_Float16 *>(hbuf + nxt + wr_off + q * 16) = hi;
                } else {
                    *reinterpret_cast<_Float16 *>(hbuf + nxt + wr_off + (2 * q) * 16) = hi;
                    *reinterpret_cast<_Float16 *>(hbuf + nxt + wr_off + (2 * q + 1) * 16) = lo;
                }
            }
            lds_barrier();
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) buf_store_float<2>(hprev[q], orsrc, ovoff + q * 256, ocol(s_end - 1));      // the last step's h
    if constexpr (HEAD) head_strip(strip1 - 1);                      // (the last step's barrier made its image visible)
}

}  // namespace mdk
