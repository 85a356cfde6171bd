// Read-level front end of `LatentSpaceLSTM` (reference medaka/architectures/latent_space_lstm.py:
// 154-196 and read_level_modules.py:7-100):
//
//   uint8 (B, P, D, F) -> Embedding(base) + Embedding(strand+1) (+) q/25-1 (+ dwell)     7|8 features
//   -> Conv1d(k=1) -> ReLU -> BatchNorm1d -> Conv1d(k=17, pad 8) -> ReLU -> BatchNorm1d  (per read)
//   -> mean over the non-empty reads                                                     (B, P, 128)
//   [-> Linear(128 -> lstm_size): commutes with the mean and no non-linearity separates it from
//       W_ih of the first LSTM, so the host folds it into that projection (rl_api.hip)]
//
// One fused kernel, nothing per-read is ever written to HBM (the per-read activations would be
// 512 B x B x D x P).  A work-group owns (window b, 96 positions) and walks the reads:
//   1. features + conv1 + ReLU + BN for the tile's positions + 8 halo each side straight into LDS as the
//      fp16 hi/lo A operand of conv2 (zero outside the window = the zero padding of conv2);
//   2. conv2 as an implicit GEMM on the matrix cores: M = 96 positions, N = 128 channels,
//      K = 17 taps x 128 channels = 68 k-steps; the A fragment of tap tau is the same LDS tile
//      read tau rows further down; W2 B-fragments stream from L2 (pre-packed);
//   3. bias + ReLU + BN in registers, accumulated over the reads in registers (empty reads are
//      skipped: their mask is 0);
//   4. mean, store in the layout the LSTM projection GEMM reads.
// fp32 parity through the same fp16 hi/lo split as the GRU kernels (three products, fp32
// accumulate); BatchNorm is folded to y = a*x + c at load time (eval mode).
#pragma once
#include "common.hpp"
#include "layout.hpp"

namespace mdk {

#ifndef MDK_RL_MT
#define MDK_RL_MT 6
#endif
constexpr int kRlMT = MDK_RL_MT;                 // 16-position M-tiles per wave; the W2 fragments of a k-step are loaded
                                                 // once per kRlMT tiles.  Sweep (profiles/run_front_mt.sh, 100 x 10000 x 50):
                                                 // 4: 77.3 / 38.9 ms (fp32 / half), 6: 76.0 / 37.2, 7: 75.9 / 47.9, 8: 91.8 / 46.6;
                                                 // forcing 3 work-groups per CU at 4 (<= 168 VGPRs) measured the same as 2
constexpr int kRlPos = 16 * kRlMT;               // positions per work-group
constexpr int kRlHalo = 8;                       // (17 - 1) / 2
constexpr int kRlRows = kRlPos + 2 * kRlHalo;    // rows of conv1 output per tile
constexpr int kRlRowBytes = 272;                 // 128 channels x 2 B + 16 B pad (bank spread)
constexpr int kRlTaps = 17;
constexpr int kRlC = 128;                        // cnn_size == lstm_size == 128

// mask[b][d] = any non-zero byte of read d in window b (latent_space_lstm.py:164-166).
// Work-group = (window, 256 positions); 16-byte loads; flags OR-ed into a zero-initialised
// int array.  The number of non-empty reads is counted by the consumer (k_rl_front).
static __global__ __launch_bounds__(256) void k_rl_mask(const unsigned char *__restrict__ x, int P, int Dp, int F,
                                                        int *__restrict__ mask)
{
    __shared__ int flags[256];
    const int b = blockIdx.y;
    const size_t row_bytes = (size_t)Dp * F;
    const size_t beg = (size_t)blockIdx.x * 256 * row_bytes;                       // within window b
    const size_t end = min((size_t)P * row_bytes, beg + 256 * row_bytes);
    const unsigned char *xb = x + (size_t)b * P * row_bytes;
    flags[threadIdx.x] = 0;
    __syncthreads();
    // head / tail bytes around the 16-byte aligned body
    const size_t addr0 = (size_t)(xb + beg);
    size_t body_beg = beg + ((16 - (addr0 & 15)) & 15);
    if (body_beg > end) body_beg = end;
    const size_t body_end = body_beg + ((end - body_beg) & ~(size_t)15);
    for (size_t i = beg + threadIdx.x; i < body_beg; i += 256)
        if (xb[i]) flags[(i / F) % Dp] = 1;
    for (size_t i = body_end + threadIdx.x; i < end; i += 256)
        if (xb[i]) flags[(i / F) % Dp] = 1;
    for (size_t i = body_beg + (size_t)threadIdx.x * 16; i < body_end; i += 256 * 16) {
        const uint4 v = *reinterpret_cast<const uint4 *>(xb + i);
        if (v.x | v.y | v.z | v.w) {
            const unsigned int wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if ((wds[k >> 2] >> (8 * (k & 3))) & 0xff) flags[((i + k) / F) % Dp] = 1;
        }
    }
    __syncthreads();
    if (threadIdx.x < Dp && flags[threadIdx.x]) atomicOr(&mask[(size_t)b * Dp + threadIdx.x], 1);
}

struct RlFrontArgs {
    const unsigned char *x;        // [B][P][Dp][F]
    const int *mask;               // [B][Dp]  0/1
    const float *base_emb;         // [A][6]
    const float *strand_emb;       // [3][6]
    const float *w1;               // [128][8]  conv1 weights, rows padded to 8 features
    const float *b1, *a1, *c1;     // [128] conv1 bias, BN1 scale, BN1 shift
    const half8 *w2frag;           // [17][4 kb][4 waves][2 nt][2 hi/lo][64]
    const float *b2, *a2, *c2;     // [128]
    float *pooled;                 // TILED: act_t layout, D = 1;  else natural [B][P][128]
    int B, P, Dp, F, nf, n_alpha;
    float s1, inv2;                // operand scales (powers of two)
};

// TILED: store in the tile-major activation layout (layout.hpp, D = 1; lstm_size 128) instead of
// natural (B, P, 128) rows (lstm_size 384).
// HP: half precision (`TorchModel.half()`): conv2 as ONE fp16 product instead of the three of the
// hi/lo split (the reference's own half mode runs these convolutions in fp16 under autocast).
template <bool TILED, bool HP = false>
__global__ __launch_bounds__(256, 2) void k_rl_front(const RlFrontArgs A)
{
    __shared__ __attribute__((aligned(16))) unsigned char ytile[2 * kRlRows * kRlRowBytes];
    // conv1 is linear in its 7 | 8 inputs and the first six of them are (embedding of base) + (embedding of strand):
    // their contribution + bias is a table over the <= 8 x 3 (base, strand) pairs, built once per work-group with
    // the SAME fma chain the per-position loop used (b1, then features 0..5), so the results are bit-identical
    __shared__ float tab1[8 * 3][kRlC];
    __shared__ float fq[kRlRows], fdw[kRlRows];
    __shared__ int fidx[kRlRows];              // (base, strand) pair of the row, -1 outside the window
    __shared__ int n_reads_s;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * kRlPos;
    const int n = lane & 15, g = lane >> 4;

    if (tid == 128) {
        int cnt = 0;
        for (int d = 0; d < A.Dp; ++d) cnt += A.mask[(size_t)b * A.Dp + d] != 0;
        n_reads_s = cnt;
    }
    for (int e = tid; e < A.n_alpha * 3 * kRlC; e += 256) {
        const int c = e % kRlC, pair = e / kRlC, bi = pair / 3, si = pair % 3;
        float v = A.b1[c];
#pragma unroll
        for (int f = 0; f < 6; ++f) v = fmaf(A.w1[c * 8 + f], A.base_emb[bi * 6 + f] + A.strand_emb[si * 6 + f], v);
        tab1[pair][c] = v;
    }

    // conv1 constants of this thread's two adjacent channels (one 4-byte LDS store per fp16 pair)
    const int ci = 2 * (tid & 63), rsel = tid >> 6;
    float w6[2], w7[2], a1v[2], c1v[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        w6[j] = A.w1[(ci + j) * 8 + 6]; w7[j] = A.w1[(ci + j) * 8 + 7];
        a1v[j] = A.a1[ci + j]; c1v[j] = A.c1[ci + j];
    }
    // epilogue constants of this lane's conv2 / linear output columns: co = 32w + 16nt + n
    float b2v[2], a2v[2], c2v[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int co = 32 * w + 16 * nt + n;
        b2v[nt] = A.b2[co]; a2v[nt] = A.a2[co]; c2v[nt] = A.c2[co];
    }
    floatx4 pool[kRlMT][2];
#pragma unroll
    for (int mt = 0; mt < kRlMT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) pool[mt][nt] = floatx4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    const unsigned char *xb = A.x + (size_t)b * A.P * A.Dp * A.F;
    unsigned char *yhi = ytile, *ylo = ytile + kRlRows * kRlRowBytes;
    constexpr int NS = HP ? 1 : 2;
    constexpr int kSteps = kRlTaps * 4;            // k-steps of conv2: (tap, 32-channel block)
    // W2 fragments of k-step i: [i][4 waves][2 nt][2 hi/lo][64 lanes]
    const half8 *wp = A.w2frag + (size_t)w * 4 * 64 + lane;
    auto load_b = [&](int i, half8 (&dst)[2][NS]) {
        const half8 *wk = wp + (size_t)i * (4 * 4 * 64);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) dst[nt][sp] = wk[(nt * 2 + sp) * 64];
    };
    const int a_lane = n * kRlRowBytes + (8 * g) * 2;

    for (int d = 0; d < A.Dp; ++d) {
        if (A.mask[(size_t)b * A.Dp + d] == 0) continue;   // uniform: empty reads contribute 0
        // ---- 1a. per-position inputs of the tile (+ halo): table row, q-score, dwell
        if (tid < kRlRows) {
            const int pp = p0 - kRlHalo + tid;
            int idx = -1;
            float q = 0.f, dw = 0.f;
            if (pp >= 0 && pp < A.P) {
                const unsigned char *xr = xb + ((size_t)pp * A.Dp + d) * A.F;
                int bi = xr[0];
                if (bi >= A.n_alpha) bi = A.n_alpha - 1;
                int si = (int)(signed char)xr[2] + 1;        // strand -1/0/+1 -> 0/1/2
                si = si < 0 ? 0 : (si > 2 ? 2 : si);
                idx = bi * 3 + si;
                q = (float)xr[1] / 25.0f - 1.0f;
                if (A.nf == 8) dw = (float)xr[4];
            }
            fidx[tid] = idx; fq[tid] = q; fdw[tid] = dw;
        }
        // the W2 fragments of the first two k-steps of this read travel while conv1 runs
        half8 bq[3][2][NS];
        load_b(0, bq[0]);
        load_b(1, bq[1]);
        __syncthreads();
        // ---- 1b. conv1 (k=1) + ReLU + BN1 -> fp16 hi/lo tile
        for (int r = rsel; r < kRlRows; r += 4) {
            const int idx = fidx[r];
            const float q = fq[r], dw = fdw[r];
            const float2 t2 = *reinterpret_cast<const float2 *>(&tab1[idx < 0 ? 0 : idx][ci]);
            half2_t hi2, lo2;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v = fmaf(w6[j], q, j ? t2.y : t2.x);
                v = fmaf(w7[j], dw, v);
                v = fmaxf(v, 0.f);
                v = fmaf(a1v[j], v, c1v[j]);
                if (idx < 0) v = 0.f;                        // zero padding of conv2's input
                _Float16 hi, lo;
                split_f16(v * A.s1, hi, lo);
                hi2[j] = hi; lo2[j] = lo;
            }
            *reinterpret_cast<half2_t *>(yhi + r * kRlRowBytes + ci * 2) = hi2;
            if constexpr (!HP) *reinterpret_cast<half2_t *>(ylo + r * kRlRowBytes + ci * 2) = lo2;
        }
        __syncthreads();
        // ---- 2. conv2 as implicit GEMM: acc[mt][nt] over 17 taps x 4 channel blocks = 68 k-steps.
        // Software pipeline (round 3; the ISA of the round-2 loop waited a full L2 round trip for the four
        // fragment loads of EVERY k-step and an LDS round trip for every M-tile: one wave kept the pipe ~32 % busy,
        // two per SIMD 70 %): the W2 fragments of k-step i+2 are requested before the MFMAs of k-step i (a ring of
        // three register sets: two k-steps = 72 MFMAs of cover for the L2 latency), the A fragments of the next
        // M-tile before the MFMAs of the current one.  Every accumulator still receives its MFMAs in the same
        // order: bit-identical results.
        floatx4 acc[kRlMT][2];
#pragma unroll
        for (int mt = 0; mt < kRlMT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = floatx4{0.f, 0.f, 0.f, 0.f};
        auto a_off = [&](int i) { return a_lane + (i >> 2) * kRlRowBytes + (32 * (i & 3)) * 2; };
        half8 ah = *reinterpret_cast<const half8 *>(yhi + a_off(0)), al = ah;
        if constexpr (!HP) al = *reinterpret_cast<const half8 *>(ylo + a_off(0));
        // k-step i: fragments bb (requested two k-steps ago), request those of k-step i + 2 into bn
        auto kstep = [&](int i, const half8 (&bb)[2][NS], half8 (&bn)[2][NS]) {
            load_b(i + 2 < kSteps ? i + 2 : kSteps - 1, bn);   // (the last two re-request the last block)
            __builtin_amdgcn_sched_barrier(0);                 // the requests stay in front of this k-step's MFMAs
            const int off0 = a_off(i), off1 = a_off(i + 1 < kSteps ? i + 1 : i);
#pragma unroll
            for (int mt = 0; mt < kRlMT; ++mt) {
                // A fragments of the next M-tile (or of the next k-step's first one) before this tile's MFMAs
                const int noff = (mt + 1 < kRlMT) ? off0 + 16 * (mt + 1) * kRlRowBytes : off1;
                const half8 nah = *reinterpret_cast<const half8 *>(yhi + noff);
                half8 nal = nah;                               // (half mode: never used)
                if constexpr (!HP) nal = *reinterpret_cast<const half8 *>(ylo + noff);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    acc[mt][nt] = mfma16(ah, bb[nt][0], acc[mt][nt]);
                    if constexpr (!HP) {
                        acc[mt][nt] = mfma16(al, bb[nt][0], acc[mt][nt]);
                        acc[mt][nt] = mfma16(ah, bb[nt][1], acc[mt][nt]);
                    }
                }
                ah = nah; al = nal;
                __builtin_amdgcn_sched_group_barrier(0x100, NS, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, HP ? 2 : 6, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        static_assert(kSteps == 68, "the three-deep fragment ring below is written for 68 k-steps (66 + 2)");
#pragma unroll 1
        for (int i = 0; i < 66; i += 3) {
            kstep(i, bq[0], bq[2]);
            kstep(i + 1, bq[1], bq[0]);
            kstep(i + 2, bq[2], bq[1]);
        }
        kstep(66, bq[0], bq[2]);
        kstep(67, bq[1], bq[0]);
        __syncthreads();   // everybody is done reading the conv1 tile
        // ---- 3. bias + ReLU + BN2, accumulated over the reads in registers
#pragma unroll
        for (int mt = 0; mt < kRlMT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = fmaf(acc[mt][nt][r], A.inv2, b2v[nt]);
                    v = fmaxf(v, 0.f);
                    pool[mt][nt][r] += fmaf(a2v[nt], v, c2v[nt]);
                }
    }

    // ---- 4. mean over the non-empty reads, store
    const float inv_n = 1.0f / (float)n_reads_s;     // 0 reads -> inf/nan, as the reference's 0/0
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int co = 32 * w + 16 * nt + n;
#pragma unroll
        for (int mt = 0; mt < kRlMT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = p0 + 16 * mt + 4 * g + r;
                if (t < A.P) {
                    if constexpr (TILED)   // window b = tile b>>3, lane group (b&7)>>1, q = b&1
                        A.pooled[act_block(1, b >> 3, A.P, t) + act_in_block(0, co >> 4, b & 1, ((b & 7) >> 1) * 16 + (co & 15))] =
                            pool[mt][nt][r] * inv_n;
                    else
                        A.pooled[((size_t)b * A.P + t) * kRlC + co] = pool[mt][nt][r] * inv_n;
                }
            }
    }
}

}  // namespace mdk
