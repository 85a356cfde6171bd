// GRU layer >= 1, projection fused into the recurrence (rec_fused.hpp) -- with the projection ROLLED UNDER the recurrence
// steps instead of alternating with them.
//
// k_rec_fused runs two phases per strip of 8 scan steps: 288 projection MFMAs per wave back to back (the matrix pipe at
// 0.94 of its bound), then 8 recurrence steps in which the pipe is busy 768 of ~1700 cycles (profiles/r4_pmc_step.csv:
// 50-52 % chip-wide): every step is h image -> 48 MFMAs per SIMD -> tanh chain -> LDS publish -> barrier, and nothing
// overlaps the chain.  The projection has no chain at all.  Here it is cut into 16 blocks of one k-step x one HALF strip
// (2 row-tiles x 3 gates x 3 split products = 18 MFMAs) and every recurrence step carries two of them:
//
//   block H  right behind the barrier, in front of the wait for the h image (the pipe works through it while the image
//            arrives from LDS);
//   block T  behind the step's last recurrence MFMA, interleaved with the tanh / blend / publish chain and running on
//            while the wave sits in the barrier.
//
// Accumulators: NONE are added.  The rows of a strip's gi are consumed in order -- row-tile mt by steps 2 mt, 2 mt + 1 --
// so while steps 0..3 consume row-tiles 0, 1 of strip N, row-tiles 2, 3 of the SAME strip are still being projected
// (into the registers strip N - 1 freed after its step 7), and while steps 4..7 consume those, row-tiles 0, 1 of strip
// N + 1 are projected into the registers steps 0..3 have just freed.  The price is W_ih twice per strip from L2 instead
// of once (each half strip streams all of it: 786 KB per work-group and strip, ~50 B per CU and cycle against the 64 the
// L2 delivers) and a second register set for its fragments: a block's 6 fragments are requested one whole step ahead
// (two sets, bH / bT), because on gfx9 every vector-memory operation returns in order -- a wait for a fragment is a wait
// for every older request, among them the HBM reads of the next strip's activations.
// The A image is staged as in k_rec_fused, half a strip at a time: the half that steps 4..7 project is written under
// steps 0..3, the other under steps 4..7 (piece p of the next strip is requested in step 2p and stored in step 2p + 1).
// The epilogue (scale + folded bias) moves to the consumer: fmaf(acc, gi_scale, bias) when a step takes its gi -- the same
// operation on the same operands, so the kernel is BIT-IDENTICAL to k_rec_fused and with it to k_gi_gemm + k_rec_mfma
// (tests/test_parity_gpu.py::test_fused_projection_agrees_bitwise runs all of them).
// fp32-parity mode, bidirectional input (K = 256), 8-window work-groups; HEAD as in rec_fused.hpp.
#pragma once
#include "rec_fused.hpp"

namespace mdk {

#ifndef MDK_ROLL_TV
#define MDK_ROLL_TV 2      // block T: VALU instructions of the chain between two projection MFMAs
#endif
#ifndef MDK_ROLL_TPOST
#define MDK_ROLL_TPOST 0   // block T: 1 = all of it BEHIND the chain and the publish, in front of the barrier (experiment)
#endif
#ifndef MDK_ROLL_TPRE
#define MDK_ROLL_TPRE 2    // block T: projection MFMAs in front of the chain (the wave waits for its last recurrence MFMA there anyway)
#endif

// dynamic LDS: the A image first (offset 0: every one of its addresses is then one lane register + an immediate), the ring
// of h images behind it, the HEAD = 2 delivery table last
__host__ __device__ inline constexpr size_t roll_lds_bytes(int head) {
    return 65536 + (size_t)(head ? 8 : 2) * kHBufBytes + (head ? 8192 : 0) + (head == 2 ? kTileWin * 16 : 0) + 32768;
}

template <int HEAD>
__global__ __launch_bounds__(512, 2) void k_rec_roll(
    const float *__restrict__ act_in, const half8 *__restrict__ wihfrag, const float *__restrict__ bias,
    const half8 *__restrict__ wfrag, const float *__restrict__ b_hn, float *__restrict__ out,
    int n_tiles, int T, int D,
    const float *__restrict__ inv_scale_rec_p, const float *__restrict__ inv_scale_gi_p,
    const float *__restrict__ up_scale_rec_p, float a_scale,
    int reverse_mask, int s0, int ns,
    const half8 *__restrict__ wlin_frag, float lin_inv_scale, float *__restrict__ lpart,
    const float *__restrict__ lin_b, float *__restrict__ probs, int nb, int normalise, SplitPlan spl)
{
    constexpr bool FIN = HEAD == 2;
    constexpr int KSTEPS = 8, DIN = 2, MT = kFusedMT;
    constexpr int NIMG = HEAD ? 8 : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *const xsb = smem;                    // A image [split 2][mt MT][KSTEPS][64 lanes] half8  (rec_fused.hpp)
    unsigned char *const hbuf = smem + 65536;           // NIMG images of h
    struct FinRow { long base; int lo, hi; };
    constexpr int kWlinOff = 65536 + NIMG * kHBufBytes;   // HEAD: this direction's 8 W_lin fragments (every wave multiplies by the same ones)
    FinRow *const ftab = reinterpret_cast<FinRow *>(smem + kWlinOff + (HEAD ? 8192 : 0));
    constexpr int kStageOff = kWlinOff + (HEAD ? 8192 : 0) + (FIN ? kTileWin * 16 : 0);   // raw fp32 pieces in flight: [slot 2][half 2][wave 8][64 lanes] x 16 B
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

    half8 wf[4][3][2];
    {
        const half8 *wp0 = wfrag + ((size_t)(d * 8 + w8) * 24) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int gate = 0; gate < 3; ++gate)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) wf[ks][gate][sp] = wp0[(size_t)((ks * 3 + gate) * 2 + sp) * 64];
    }
    for (int i = tid; i < NIMG * kHBufBytes / 4; i += 512) reinterpret_cast<uint32_t *>(hbuf)[i] = 0u;
    if constexpr (HEAD != 0) {
        // (from L2 at the top of every strip they were eight dependent round trips under register pressure -- and nothing
        // that waits on vmcnt belongs between the steps: the W_ih fragments of the next blocks are in flight there)
        *reinterpret_cast<half8 *>(smem + kWlinOff + tid * 16) = wlin_frag[(size_t)d * 8 * 64 + tid];
    }
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
    const float os = up_scale_rec_p[d];
    const float gi_scale = inv_scale_gi_p[d] * os;
    float bv[3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) bv[nt] = bias[(size_t)d * kG + nt * kH + u] * os;

    const int s_end = s0 + ns;
    // ---- every HBM address of the scan: wave-uniform base + ONE lane offset + a scalar that follows the step (common.hpp)
    // this layer's output: block of column t at t * D * 4096 bytes, this lane's two values 256 bytes apart
    const __amdgpu_buffer_rsrc_t orsrc = make_rsrc(out + act_block(D, tile, T, 0));
    const unsigned ovoff = (unsigned)act_in_block(d, w8, 0, lane) * 4u;
    const int obytes = D * 4096;
    auto ocol = [&](int s) { return (unsigned)((reverse ? (T - 1 - s) : s) * obytes); };     // scan step -> byte offset of its column
    float hprev[2] = {0.f, 0.f};

    const int rd_off = 65536 + g * kHGroupStride + c * 16;
    const int wr_off = 65536 + (w8 >> 1) * kHKStride + (2 * (w8 & 1) + (c >> 3)) * kHGroupStride + (4 * g) * 16 + (c & 7) * 2;

    // ---- staging (rec_fused.hpp): piece `it` of a strip = this thread's 8 floats of scan steps 2 it, 2 it + 1 = row-tile it.
    // Input block of column t at t * 8192 bytes.  Scan step 2 it + th of the strip, th = tid >> 8: forward that is column
    // 8 strip + 2 it + th, reversed T - 1 - (8 strip + 2 it + th) = (T - 2 - 8 strip - 2 it) + (1 - th): the lane part stays
    // non-negative either way.
    const __amdgpu_buffer_rsrc_t irsrc = make_rsrc(act_in + act_block(DIN, tile, T, 0));
    const int pj = tid & 255, pth = tid >> 8;
    const int pgg = pj & 3, pqq = (pj >> 2) & 1, phalf = (pj >> 3) & 1, pchunk = pj >> 4;
    const unsigned pvoff = (unsigned)((pchunk * 16 + pqq * 8 + pgg * 2 + phalf) * 32 + (reverse ? 1 - pth : pth) * 8192);
    const int pk8 = pchunk * 2 + phalf;
    const unsigned pst = (unsigned)(((pk8 >> 2) * 64 + (pk8 & 3) * 16 + 4 * pgg + 2 * pqq + pth) * 16);   // LDS byte offset inside (split, mt)
    const unsigned wvoff = (unsigned)lane * 16u;
    // A piece travels HBM -> LDS by DMA (buffer_load ... lds: no register holds it while it is in flight -- the step has none
    // to spare) into a slot of its own wave, and is split to fp16 hi / lo from there one step later by the thread that
    // asked for it: nobody else reads the slot, so the only synchronisation is the wave's own vmcnt.
    typedef __attribute__((address_space(3))) void lds_void;
    auto piece_load = [&](int strip, int it, int slot) {
        const int s2 = strip * kFusedSteps + 2 * it;
        const unsigned so = (unsigned)((reverse ? (T - 2 - s2) : s2) * 8192);
        // (the instruction offset applies to the LDS address as well as to the global one: the second half's M0 base is 16 short)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(irsrc, (lds_void *)(smem + kStageOff + ((slot * 2 + 0) * 8 + w8) * 1024), 16, pvoff, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(irsrc, (lds_void *)(smem + kStageOff + ((slot * 2 + 1) * 8 + w8) * 1024 - 16), 16, pvoff, so, 16, 0);
    };
    auto piece_store = [&](int it, int slot) {
        const floatx4 v0 = *reinterpret_cast<const floatx4 *>(smem + kStageOff + ((slot * 2 + 0) * 8 + w8) * 1024 + wvoff);
        const floatx4 v1 = *reinterpret_cast<const floatx4 *>(smem + kStageOff + ((slot * 2 + 1) * 8 + w8) * 1024 + wvoff);
        const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        half8 hi, lo;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            _Float16 a, b;
            split_f16(v[i] * a_scale, a, b);
            hi[i] = a; lo[i] = b;
        }
        *reinterpret_cast<half8 *>(xsb + pst + (0 * MT + it) * KSTEPS * 1024) = hi;
        *reinterpret_cast<half8 *>(xsb + pst + (1 * MT + it) * KSTEPS * 1024) = lo;
    };

    const int strip0 = s0 / kFusedSteps, strip1 = s_end / kFusedSteps;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        piece_load(strip0, it, it & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        piece_store(it, it & 1);
    }

#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int gate = 0; gate < 3; ++gate)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) asm volatile("" ::"v"(wf[ks][gate][sp]));
    asm volatile("" ::"v"(bhn));
    __syncthreads();
    if (s0 > 0) {   // resume (rec_fused.hpp): h of scan step s0 - 1 from the output, and its fp16 image
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float h = buf_load_float(orsrc, ovoff + q * 256, ocol(s0 - 1));
            hprev[q] = h;
            _Float16 hi, lo;
            split_f16(h * kActScale, hi, lo);
            unsigned char *img = smem + (s0 & (NIMG - 1)) * kHBufBytes + wr_off;
            *reinterpret_cast<_Float16 *>(img + (2 * q) * 16) = hi;
            *reinterpret_cast<_Float16 *>(img + (2 * q + 1) * 16) = lo;
        }
        __syncthreads();
    }

    // W_ih fragments [D][8 waves][KSTEPS][3 gates][2 hi/lo][64 lanes] half8: fragment f of this wave at f * 1024 bytes
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(wihfrag + ((size_t)(d * 8 + w8) * KSTEPS) * 6 * 64);

    // ---- HEAD (rec_fused.hpp; the partial logits' addresses in the scalar form)
    const __amdgpu_buffer_rsrc_t lorsrc = make_rsrc(lpart + ((size_t)(D - 1 - d) * n_tiles + tile) * T * 40);   // the other direction's
    const __amdgpu_buffer_rsrc_t lprsrc = make_rsrc(lpart + ((size_t)d * n_tiles + tile) * T * 40);             // this direction's
    // (the head's lane constants are recomputed where they are used, from an opaque copy of the lane offset: kept live
    // across the strip they were the registers that spilled -- and a scratch reload is a vmcnt(0) in the middle of the scan)
    auto head_lane = [&](int &hc, int &hg) {
        unsigned lv = wvoff;
        asm volatile("" : "+v"(lv));
        hc = (int)(lv >> 4) & 15;
        hg = (int)(lv >> 8);
    };
    float oth = 0.f;
    float lbs[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (FIN) {
#pragma unroll
        for (int cl = 0; cl < 5; ++cl) lbs[cl] = lin_b[cl];
    }
    auto head_strip = [&](int hs) {
        int c, g;
        head_lane(c, g);
        const int cq = c & 7, qsel = c >> 3;
        const int s = hs * kFusedSteps + w8;
        const int t = reverse ? (T - 1 - s) : s;
        const unsigned char *img = smem + ((w8 + 1) & 7) * kHBufBytes + 65536 + g * kHGroupStride + c * 16;
        const unsigned char *wl = smem + kWlinOff + ((g * 16 + c) * 16);
        floatx4 la = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const half8 a = *reinterpret_cast<const half8 *>(img + ks * kHKStride);
            la = mfma16(a, *reinterpret_cast<const half8 *>(wl + (ks * 2 + 0) * 1024), la);
            la = mfma16(a, *reinterpret_cast<const half8 *>(wl + (ks * 2 + 1) * 1024), la);
        }
        float own[2];
        own[0] = (la[0] + la[1]) * lin_inv_scale;
        own[1] = (la[2] + la[3]) * lin_inv_scale;
        if constexpr (!FIN) {
            if (c < 5) {
                const unsigned vo = (unsigned)(((2 * g) * 5 + c) * 4);
                buf_store_float(own[0], lprsrc, vo, (unsigned)(t * 160));
                buf_store_float(own[1], lprsrc, vo + 20, (unsigned)(t * 160));
            }
        } else {
            const int hl = g * 16 + c;
            const float up = __shfl(own[1], hl - 8);
            float v = qsel ? up : own[0];
            v = d == 0 ? v + oth : oth + v;
            float a[5];
#pragma unroll
            for (int cl = 0; cl < 5; ++cl) a[cl] = __shfl(v, (hl & 56) + cl) + lbs[cl];
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
            int c, g;
            head_lane(c, g);
            const int cq = c & 7, qsel = c >> 3;
            const int s = hs * kFusedSteps + w8;
            const int t = reverse ? (T - 1 - s) : s;
            oth = buf_load_float(lorsrc, (unsigned)(((2 * g + qsel) * 5 + (cq < 5 ? cq : 4)) * 4), (unsigned)(t * 160));
        }
    };

    // ---- the projection, one block = one k-step of one half strip (row-tiles 2 hs, 2 hs + 1)
    floatx4 acc[MT][3];
    half8 bH[3][2], bT[3][2];       // W_ih fragments [gate][hi | lo] of the step's two blocks
    auto load_b = [&](half8 (&b)[3][2], int ks) {
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            // (a block's six fragments are 1 KB apart: four by the 12-bit immediate, two from a second scalar offset)
            const int f0 = nt * 2, f1 = nt * 2 + 1;
            b[nt][0] = buf_load_half8(wrsrc, wvoff + (f0 & 3) * 1024, ks * 6144 + (f0 >> 2) * 4096);
            b[nt][1] = buf_load_half8(wrsrc, wvoff + (f1 & 3) * 1024, ks * 6144 + (f1 >> 2) * 4096);
        }
    };
    auto proj_block = [&](const half8 (&b)[3][2], int hs, int ks, bool first) {
#pragma unroll
        for (int ml = 0; ml < 2; ++ml) {
            const int mt = 2 * hs + ml;
            const half8 ah = *reinterpret_cast<const half8 *>(xsb + wvoff + ((0 * MT + mt) * KSTEPS + ks) * 1024);
            const half8 al = *reinterpret_cast<const half8 *>(xsb + wvoff + ((1 * MT + mt) * KSTEPS + ks) * 1024);
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                floatx4 a0 = acc[mt][nt];
                if (first) a0 = floatx4{0.f, 0.f, 0.f, 0.f};
                a0 = mfma16(ah, b[nt][0], a0);
                a0 = mfma16(al, b[nt][0], a0);
                a0 = mfma16(ah, b[nt][1], a0);
                acc[mt][nt] = a0;
            }
        }
    };
    // first strip: its row-tiles 0, 1 in one go (the k_rec_fused loop over half the rows); 2, 3 roll under steps 0..3
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) acc[mt][nt] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int ks = 0; ks < KSTEPS; ++ks) {
        const unsigned so = (unsigned)ks * 6144u;
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            bH[nt][0] = buf_load_half8(wrsrc, wvoff + (nt * 2 + 0) * 1024, so);
            bH[nt][1] = buf_load_half8(wrsrc, wvoff + (nt * 2 + 1) * 1024, so);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const half8 ah = *reinterpret_cast<const half8 *>(xsb + wvoff + ks * 1024 + ((0 * MT + mt) * KSTEPS) * 1024);
            const half8 al = *reinterpret_cast<const half8 *>(xsb + wvoff + ks * 1024 + ((1 * MT + mt) * KSTEPS) * 1024);
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                acc[mt][nt] = mfma16(ah, bH[nt][0], acc[mt][nt]);
                acc[mt][nt] = mfma16(al, bH[nt][0], acc[mt][nt]);
                acc[mt][nt] = mfma16(ah, bH[nt][1], acc[mt][nt]);
            }
        }
    }
    load_b(bH, 0);
    load_b(bT, 1);

    for (int strip = strip0; strip < strip1; ++strip) {
        const int nstrip = strip + 1 < strip1 ? strip + 1 : strip;     // branch-free: past the end the last strip is staged / projected again
#pragma unroll
        for (int j = 0; j < kFusedSteps; ++j) {
            const int step = strip * kFusedSteps + j;
            constexpr int kSlots = NIMG - 1;
            const int cur = (j & kSlots) * kHBufBytes;            // (strips start at multiples of 8: step & (NIMG - 1) = j & ...)
            const int nxt = ((j + 1) & kSlots) * kHBufBytes;
            const int hs = j < 4 ? 1 : 0;                 // half strip under projection: rows 2, 3 of this strip, then 0, 1 of the next
            const int kA = 2 * (j & 3), kB = kA + 1;
            const int kA1 = 2 * ((j + 1) & 3), kB1 = kA1 + 1;
            half8 a[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a[ks] = *reinterpret_cast<const half8 *>(smem + cur + ks * kHKStride + rd_off);
            __builtin_amdgcn_sched_barrier(0);
            // ================= block H: the pipe works through it while the h image arrives
#ifndef ROLL_DBG_NOH
            proj_block(bH, hs, kA, kA == 0);
#endif
            // ... and the VALU is idle under it: the piece requested a step ago is split and put into the image here
#ifndef ROLL_DBG_NOPIECE
            if ((j & 1) == 1) {
                // hipcc does not order an LDS read behind the DMA that fills it (seen in the ISA: the block's own vmcnt(11)
                // was all that stood in front of this read).  Younger than the request by now: the 6 fragments block T of
                // the previous step asked for, whatever order its section's requests and stores were scheduled in.
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                piece_store(j >> 1, (j >> 1) & 1);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
            load_b(bH, kA1);
            __builtin_amdgcn_sched_barrier(0);
            // ================= the recurrence step (rec_fused.hpp)
            floatx4 ar = floatx4{0.f, 0.f, 0.f, 0.f}, az = ar, anh = ar, anl = ar;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    ar = mfma16(a[ks], wf[ks][0][sp], ar);
                    az = mfma16(a[ks], wf[ks][1][sp], az);
                }
            }
#ifndef ROLL_DBG_NOPIECE
            if ((j & 1) == 0) piece_load(nstrip, j >> 1, (j >> 1) & 1);
#endif
            if (j == MDK_FIN_REQ) head_request(strip);
            // deferred store of the previous step's h (rec_mfma.hpp DS), unconditional: the first step of a launch writes its
            // incoming state into its OWN slot, which the next step's store then overwrites
            {
                const unsigned so = ocol(step > s0 ? step - 1 : step);
#pragma unroll
                for (int q = 0; q < 2; ++q) buf_store_float(hprev[q], orsrc, ovoff + q * 256, so);
            }
            __builtin_amdgcn_sched_barrier(0);
            // (the n tiles take the image from LDS a second time, two k-steps at a time: four live fragments fewer beside the
            // r, z accumulators -- the step's register peak is here)
            {
                half8 an2[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) an2[ks] = *reinterpret_cast<const half8 *>(smem + cur + ks * kHKStride + rd_off);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    anh = mfma16(an2[ks], wf[ks][2][0], anh);
                    anl = mfma16(an2[ks], wf[ks][2][1], anl);
                }
            }
            float rr[2], zz[2], gnv[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = 2 * q + (j & 1);
                // the projection's epilogue, at the consumer: gi = acc * (inv_scale_gi * os) + bias * os  (gi_proj.hpp)
                const float gr = fmaf(acc[j >> 1][0][r], gi_scale, bv[0]);
                const float gz = fmaf(acc[j >> 1][1][r], gi_scale, bv[1]);
                gnv[q] = fmaf(acc[j >> 1][2][r], gi_scale, bv[2]);
                const float tr = gr + (ar[2 * q] + ar[2 * q + 1]);
                const float tz = gz + (az[2 * q] + az[2 * q + 1]);
                rr[q] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(tr * c_sig));
                zz[q] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(tz * c_sig));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // 3 VALU
            }
            __builtin_amdgcn_sched_barrier(0);
            // ================= block T under the chain: tanh, blend, split, publish
#if !defined(ROLL_DBG_NOT) && !MDK_ROLL_TPOST
            proj_block(bT, hs, kB, false);
#endif
            float hn[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float tn = ((anh[2 * q] + anl[2 * q]) + (anh[2 * q + 1] + anl[2 * q + 1])) + bhn;
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
                *reinterpret_cast<_Float16 *>(smem + nxt + wr_off + (2 * q) * 16) = hi;
                *reinterpret_cast<_Float16 *>(smem + nxt + wr_off + (2 * q + 1) * 16) = lo;
            }
#if MDK_ROLL_TPOST
            __builtin_amdgcn_sched_barrier(0);
            proj_block(bT, hs, kB, false);
#elif !defined(ROLL_DBG_NOT)
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);               // the block's 4 A fragments (LDS reads)
            if (MDK_ROLL_TPRE > 0) __builtin_amdgcn_sched_group_barrier(0x008, MDK_ROLL_TPRE, 0);
#pragma unroll
            for (int i = 0; i < 18 - MDK_ROLL_TPRE; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x002, MDK_ROLL_TV, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
            load_b(bT, kB1);
            lds_barrier();
        }
        // HEAD: the ring now holds the images of this strip's 8 steps (step j's in slot (j + 1) & 7; the next strip's first
        // publish, a step away, is what overwrites slot 1)
        if constexpr (HEAD != 0) head_strip(strip);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) buf_store_float(hprev[q], orsrc, ovoff + q * 256, ocol(s_end - 1));
}

}  // namespace mdk
