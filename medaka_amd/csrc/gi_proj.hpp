// Input projections  gi = x W_ih^T + b   for every column and both directions
// (the time-parallel half of nn.GRU, called from reference medaka/architectures/gru.py:66).
//
//  * k_gi_small : layer 0, K = num_features (10).  Exact fp32 FMA on the VALU: 10 FMAs per
//                 output make this a pure HBM-write kernel (1.5 KB out per 40 B in).
//  * k_gi_gemm  : layers >= 1, K = 256 (or 128).  49 % of the network's FLOPs.  fp16x2-split
//                 MFMA GEMM (three products hi*hi + lo*hi + hi*lo into one fp32 accumulator,
//                 operands pre-scaled by powers of two), fp32 in / fp32 out.
// Both write gi in the tile-major register order of the recurrence kernel (layout.hpp); the
// GEMM also reads the previous layer's activations in that order, so both are streaming kernels
// over contiguous blocks.
// The folded bias is b_ih + b_hh for the r and z gates and b_ih for n (b_hn must stay inside
// the r * (.) product, see rec_mfma.hpp).
#pragma once
#include "common.hpp"
#include "layout.hpp"

namespace mdk {

// ------------------------------------------------------------------------------------------
// layer 0.  One 768-thread work-group per (tile, direction, strip of time steps); thread f4 owns float4
// number f4 of every 12 KB gi block of its tile.  With the gate-fastest block order (layout.hpp) the four
// elements of a float4 are consecutive (lane, gate) pairs, so each has its own (window, unit, gate): their
// K weights stay in registers, x rows are tiny broadcast loads.  Every block is written as one contiguous
// 12 KB run.  (Only the unfused / out-of-range fallback path runs this kernel.)
template <int KMAX>
__global__ __launch_bounds__(768) void k_gi_small(
    const float *__restrict__ x,      // [B][T][K] natural layout (the reference's batch tensor)
    const float *__restrict__ w_ih_t, // [D][K][384]  (transposed at load time)
    const float *__restrict__ bias,   // [D][384]     folded bias
    float *__restrict__ gi,           // gi_t
    int B, int T, int K, int n_tiles, int t_per_block, const float *__restrict__ out_scale_p,
    const int *__restrict__ cond, int want)
{
    if (cond != nullptr && ((*cond != 0) != (want != 0))) return;   // see k_pack_x
    const int tile = blockIdx.x;
    const int d = blockIdx.y;
    const float os = out_scale_p[d];
    const int f4 = threadIdx.x;
    float wreg[4][KMAX];
    float b4[4];
    const float *xw[4];
    bool real[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = 4 * f4 + e;                       // element of the block: ((w8*2+q)*64 + lane)*3 + gate
        const int gate = i % 3, lane = (i / 3) & 63, wq = i / 192;
        const int q = wq & 1, w8 = wq >> 1;
        const int j = gate * kH + 16 * w8 + (lane & 15);
        const int win = tile * kTileWin + 2 * (lane >> 4) + q;
        real[e] = win < B;                              // padding windows of the last tile see x = 0
        xw[e] = x + (size_t)(real[e] ? win : 0) * T * K;
        b4[e] = bias[(size_t)d * kG + j];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) wreg[e][k] = (k < K) ? w_ih_t[((size_t)d * K + k) * kG + j] : 0.f;
    }
    const int t0 = blockIdx.z * t_per_block;
    const int t1 = min(T, t0 + t_per_block);
    float *gout = gi + gi_block(d, n_tiles, tile, T, t0, 3) + 4 * f4;
    for (int t = t0; t < t1; ++t, gout += gi_block_floats(3)) {
        float acc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[e] = b4[e];
            if (real[e]) {
                const float *xr = xw[e] + (size_t)t * K;
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (k < K) acc[e] = fmaf(xr[k], wreg[e][k], acc[e]);
            }
            acc[e] *= os;                               // exact: os is a power of two
        }
        *reinterpret_cast<float4 *>(gout) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// ------------------------------------------------------------------------------------------
// layers >= 1.  Work-group = 8 waves (the 4-wave version was instruction-issue bound, matrix pipe
// 31 % busy), M-tile = one window tile x 8 time steps = 64 rows, read as ONE contiguous run of 8
// activation blocks; 64 KB of LDS and <= 128 VGPRs so that TWO work-groups share a CU and one
// loads / stores while the other computes (a 128-row tile with one work-group per CU serialised
// load, compute and the 384 KB store tail: 2.6 ms -> measured below).  MFMA row 4g + 2q + tt of row-tile mt is
// (window 2g+q, t0 + 2*mt + tt), so accumulator register r = 2q + tt of lane g*16+c is exactly
// element `lane` of gi block t0+2mt+tt, sub-block (w8, q, gate): every accumulator register is
// stored by the wave as one contiguous 256-byte run.  The x tile is converted once to fp16
// hi/lo A-fragments in LDS (64 KB at K = 256); W_ih B-fragments stream from L2, pre-packed so
// that each lane issues one 16-byte load per fragment.  One pass per output direction: wave w8
// owns hidden units 16*w8 .. +15 of all three gates (24 accumulator tiles).
constexpr int kGemmSteps = 8;                  // time steps per work-group: M-tile = 8 windows x 8 steps
constexpr int kGemmMT = kGemmSteps / 2;        // 16-row MFMA tiles per M-tile

// HP: half-precision mode, one fp16 product instead of the three of the hi/lo split.
// NG = gate tiles per hidden unit (3 GRU, 4 LSTM): N = NG * 128 columns per direction.
// MT = 16-row MFMA tiles per work-group: M-tile = 8 windows x 2*MT time steps.
//   MT = 4 (64 rows, 64 KB of LDS, <= 128 VGPRs, two work-groups per CU): every work-group streams ALL of W_ih
//          (786 KB of B fragments) from L2 for its 64 rows = 12 KB per row, three times the HBM traffic of the row;
//          round 2's counters put the kernel at 13.7 TB/s of L2 hits -- bound by L2, not by HBM or the pipe;
//   MT = 8 (128 rows, 128 KB of LDS, one work-group per CU): 6 KB of L2 per row, bit-identical -- and measured 7 %
//          SLOWER at 1000 x 10000 (13.1 vs 12.2 ms; round 3, profiles/r3_experiments/README.md): with one work-group
//          per CU nothing overlaps the staging of the next tile, and L2 was not the limiter after all.  Not instantiated.
template <int KSTEPS, bool HP, int NG = 3, int MT = kGemmMT>   // K = 32 * KSTEPS = D_in * 128
__global__ __launch_bounds__(512, MT <= 4 ? 4 : 2) void k_gi_gemm(
    const float *__restrict__ act_in,  // act_t of the previous layer (|x| < 1: GRU outputs)
    const half8 *__restrict__ wfrag,   // [D][8 waves][KSTEPS][NG gates][2 hi/lo][64 lanes]
    const float *__restrict__ bias,    // [D][NG*128]
    float *__restrict__ gi,            // gi_t
    int n_tiles, int T, int D, const float *__restrict__ inv_scale_p,
    const float *__restrict__ out_scale_p, float a_scale,   // a_scale: power-of-two operand scale of act_in
    int strip0,                                              // first 8-step strip of this launch
    const int *__restrict__ cond, int want,                  // run only if (*cond != 0) == want (cond may be null)
    int t_end)                                               // columns >= t_end are neither read nor written
{
    if (cond != nullptr && ((*cond != 0) != (want != 0))) return;
    constexpr int DIN = KSTEPS / 4;            // directions of the input activations
    constexpr int NP = DIN * 128;              // 8-float pieces per activation block
    constexpr int STEPS = 2 * MT;              // time steps per work-group
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half8 *xs = reinterpret_cast<half8 *>(smem);   // [split 2][mt MT][KSTEPS][64 lanes]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid, tile fastest
    const int tile = blockIdx.x % n_tiles;
    const int t0 = strip0 * kGemmSteps + (blockIdx.x / n_tiles) * STEPS;

    // ---- stage: 16 blocks x NP pieces; thread-local piece j -> (g, q) fastest (LDS bank spread)
    {
        const float *src0 = act_in + act_block(DIN, tile, T, t0);
        for (int P = tid; P < STEPS * NP; P += 512) {
            const int tau = P / NP, j = P % NP;
            const int g = j & 3, q = (j >> 2) & 1, half = (j >> 3) & 1, chunk = j >> 4;
            const int piece = chunk * 16 + q * 8 + g * 2 + half;
            float v[8];
            if (t0 + tau < t_end) {
                const float *src = src0 + (size_t)tau * (DIN * 1024) + piece * 8;
                const float4 v0 = *reinterpret_cast<const float4 *>(src);
                const float4 v1 = *reinterpret_cast<const float4 *>(src + 4);
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
                split_f16(v[i] * a_scale, a, b);
                hi[i] = a; lo[i] = b;
            }
            const int row = 4 * g + 2 * q + (tau & 1), mt = tau >> 1;
            const int k8 = chunk * 2 + half, ks = k8 >> 2;
            const int slot = (k8 & 3) * 16 + row;      // A-fragment lane that consumes it
            xs[((0 * MT + mt) * KSTEPS + ks) * 64 + slot] = hi;
            if constexpr (!HP) xs[((1 * MT + mt) * KSTEPS + ks) * 64 + slot] = lo;
        }
    }
    __syncthreads();

    for (int d = 0; d < D; ++d) {
        floatx4 acc[MT][NG];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NG; ++nt) acc[mt][nt] = floatx4{0.f, 0.f, 0.f, 0.f};

        const half8 *wp = wfrag + ((size_t)(d * 8 + w8) * KSTEPS) * (2 * NG) * 64 + lane;
        // W_ih B-fragments come from L2 (all work-groups stream the same 786 KB): the fragments of k-step ks + 1 are
        // requested before the 36 MFMAs of k-step ks are issued (two register sets, the loop runs two k-steps per
        // trip so that the set index is static; 60 -> 122 VGPRs, still four waves per SIMD) -- before, every k-step
        // began by waiting one L2 round trip.  Bit-identical; measured -4 % at 1000 x 2256 (the split scan's shape,
        // 2.73 -> 2.63 ms) and nothing at 1000 x 10000: the other work-group of the CU was already covering most of
        // that wait, and the kernel sits on L2 (10.5 TB/s of fragment reads), HBM (3.8 TB/s) and a 57 % busy,
        // power-limited matrix pipe at once.
        // (four gate tiles in the fp32-parity split -- the LSTM projection -- and the K = 128 instantiation have no room for the
        // second set inside 128 registers: with it hipcc spilled 67 / 79 / 3 VGPRs to scratch -- every reload a vmcnt(0) in front
        // of the MFMAs.  ONE set there: the CU's other work-group covers the L2 round trip, as it did before round 3.)
        constexpr int NSET = (!HP && (NG == 4 || KSTEPS == 4)) ? 1 : 2;
        half8 bh[NSET][NG], bl[NSET][NG];
        auto load_b = [&](int ks, int set) {
#pragma unroll
            for (int nt = 0; nt < NG; ++nt) {
                bh[set][nt] = wp[(size_t)((ks * NG + nt) * 2 + 0) * 64];
                if constexpr (!HP) bl[set][nt] = wp[(size_t)((ks * NG + nt) * 2 + 1) * 64];
            }
        };
        auto kstep = [&](int ks, int set) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const half8 ah = xs[((0 * MT + mt) * KSTEPS + ks) * 64 + lane];
                half8 al;
                if constexpr (!HP) al = xs[((1 * MT + mt) * KSTEPS + ks) * 64 + lane];
#pragma unroll
                for (int nt = 0; nt < NG; ++nt) {
                    acc[mt][nt] = mfma16(ah, bh[set][nt], acc[mt][nt]);
                    if constexpr (!HP) {
                        acc[mt][nt] = mfma16(al, bh[set][nt], acc[mt][nt]);
                        acc[mt][nt] = mfma16(ah, bl[set][nt], acc[mt][nt]);
                    }
                }
            }
        };
        static_assert(KSTEPS % 2 == 0, "two k-steps per trip");
        if constexpr (NSET == 1) {
#pragma unroll 1
            for (int ks = 0; ks < KSTEPS; ++ks) {
                load_b(ks, 0);
                kstep(ks, 0);
            }
        } else {
            load_b(0, 0);
#pragma unroll 1
            for (int ks = 0; ks < KSTEPS; ks += 2) {
                load_b(ks + 1, NSET - 1);
                __builtin_amdgcn_sched_barrier(0);      // (hipcc otherwise sinks the requests to the end of the trip)
                kstep(ks, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 2 < KSTEPS) load_b(ks + 2, 0);
                __builtin_amdgcn_sched_barrier(0);
                kstep(ks + 1, NSET - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- epilogue: scale back, add folded bias, 256-byte runs per accumulator register
        const float os = out_scale_p[d];
        const float inv_scale = inv_scale_p[d] * os;   // powers of two: exact
        float bv[NG];
#pragma unroll
        for (int nt = 0; nt < NG; ++nt) bv[nt] = bias[(size_t)d * (NG * kH) + nt * kH + 16 * w8 + (lane & 15)] * os;
        float *gblk = gi + gi_block(d, n_tiles, tile, T, t0, NG);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = r >> 1, tt = r & 1;
                const int t = t0 + 2 * mt + tt;
                if (t < t_end) {
                    float *dst = gblk + (size_t)(2 * mt + tt) * gi_block_floats(NG) + gi_in_block(w8, q, 0, lane, NG);
                    typename FloatRun<NG>::vec_t v;
#pragma unroll
                    for (int nt = 0; nt < NG; ++nt) v[nt] = fmaf(acc[mt][nt][r], inv_scale, bv[nt]);
                    store_run<NG>(dst, v);      // the NG gates of (unit, window) are adjacent (layout.hpp)
                }
            }
        }
    }
}

}  // namespace mdk
