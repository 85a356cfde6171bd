// Splitting the scan: the latency-bound batches of the reference (100-200 windows of 10 000 columns) become the
// throughput-bound batch the chip is good at.
//
// A window is T dependent steps per layer and direction, and a batch of 200 windows has work for 100 of the 256 CUs
// (rec_mfma.hpp: one work-group per 4 windows and direction).  But a GRU forgets: started from h = 0 a few hundred
// columns before column a, both layers arrive at column a in the state the full scan has there, to within rounding
// (measured on the engine: profiles/r3_experiments/scan_split/merge_probe.txt) -- the property the reference itself
// relies on when it cuts contigs into 10 000-column windows that overlap by 1000 and stitches them at the middle of
// the overlap (medaka/medaka.py:266-272, medaka/stitch.py:169-194).  So every window is cut into S chunks, each
// extended by a margin of G columns on both sides, and the S*B "virtual windows" go through the unchanged kernels
// as ONE batch of T/S + 2G columns.
//
// Nothing is taken on trust: every junction is CERTIFIED on the device before the call returns.  For both layers and
// both directions, the state the warm-started chunk has at the junction (and again G/2 columns further along its scan)
// is compared with the state its neighbour carried there through its whole chunk; if any of these differ by more
// than kSplitEps (2^-19; 2^-12 in half-precision mode) the call is repeated as the plain sequential scan and the model stays sequential from then on.
//
//   virtual window v = k*B + w  = columns [start[k], start[k] + Tv) of window w; its columns [core0[k], core0[k+1])
//   are the ones that are delivered.
#pragma once
#include "common.hpp"
#include "layout.hpp"

namespace mdk {

constexpr int kMaxSplit = 16;
constexpr int kSplitFlagWords = 2 + 8 * (kMaxSplit - 1);   // certificate words on the device
// Largest junction difference that certifies, on h in [-1, 1].  Two scans that have merged still differ by the rounding
// noise of their different histories: measured 1e-7 .. 5e-7 in fp32-parity mode (fp16 hi/lo operands, 22 bits) and
// 1e-5 .. 2e-4 in half-precision mode (11 bits) -- profiles/r3_experiments/scan_split/check_split.txt; scans that have
// NOT merged (weights x3: memory longer than the margin) show 1e-5 / 5e-4 and more.
constexpr float kSplitEps = 1.9073486328125e-06f;     // 2^-19
constexpr float kSplitEpsHalf = 2.44140625e-04f;      // 2^-12

struct SplitPlan {
    int S = 1;                    // chunks per window (1 = not split)
    int B = 0, T = 0;             // the real batch
    int Tv = 0;                   // columns of a virtual window
    int G = 0;                    // margin
    int start[kMaxSplit];         // first real column of chunk k
    int core0[kMaxSplit + 1];     // real columns [core0[k], core0[k+1]) are delivered from chunk k
};

// x (B, T, F) -> xv (S*B, Tv, F): rows of Tv*F floats, copied as float2 (every row starts on a multiple of 2*F floats
// only when F is even: the odd case falls back to scalar copies through `vec` = 1)
static __global__ __launch_bounds__(256) void k_split_gather(const float *__restrict__ x, float *__restrict__ xv,
                                                             SplitPlan p, int F, int vec) {
    const long row_elems = (long)p.Tv * F / vec;
    const long total = (long)p.S * p.B * row_elems;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long v = i / row_elems, e = i - v * row_elems;
        const int k = (int)(v / p.B), w = (int)(v - (long)k * p.B);
        const size_t src = ((size_t)w * p.T + p.start[k]) * F;
        if (vec == 2)
            reinterpret_cast<float2 *>(xv)[i] = reinterpret_cast<const float2 *>(x + src)[e];
        else
            xv[i] = x[src + e];
    }
}

// pv (S*B, Tv, C) -> probs (B, T, C): the core columns of every chunk
static __global__ __launch_bounds__(256) void k_split_scatter(const float *__restrict__ pv, float *__restrict__ probs,
                                                              SplitPlan p, int C) {
    const long total = (long)p.B * p.T * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long col = i / C;
        const int c = (int)(i - col * C);
        const int w = (int)(col / p.T), t = (int)(col - (long)w * p.T);
        int k = (int)((long)t * p.S / p.T);             // cores are [T*k/S, T*(k+1)/S): at most one off
        while (k + 1 < p.S && t >= p.core0[k + 1]) ++k;
        while (k > 0 && t < p.core0[k]) --k;
        probs[i] = pv[(((size_t)k * p.B + w) * p.Tv + (t - p.start[k])) * C + c];
    }
}

// h of (virtual window v, local column t, direction d, unit u) in an activation buffer (layout.hpp act_t)
__device__ __forceinline__ float split_h(const float *__restrict__ act, int Tv, int v, int t, int d, int u) {
    const int tile = v >> 3, wt = v & 7;
    const int lane = (wt >> 1) * 16 + (u & 15);
    return act[act_block(2, tile, Tv, t) + act_in_block(d, u >> 4, wt & 1, lane)];
}

// Certificate.  One thread per (junction j, layer, direction, point, window, unit).  `flag[0]` |= 1 when a pair
// differs by more than eps (or is not a number); flag[1] = bits of the largest difference seen, flag[2 + y] the
// largest of certificate point y = ((j * 2 + layer) * 2 + direction) * 2 + point.
//   direction 0 scans t upwards: at real column a = core0[j+1], chunk j carried its state to a - 1 (and on into its
//   right margin); chunk j+1 arrives there warm.  Direction 1 scans downwards: chunk j+1 carried its state to a (and on
//   into its left margin), chunk j arrives there warm.  Point 1 is G/2 columns further along the scan: both chunks have
//   run on from the junction, and the differences must still be below eps there (not further out: in the outer half of
//   a margin layer 1 is fed by a layer 0 that is itself still warming up in the other direction, by design).
static __global__ __launch_bounds__(128) void k_split_verify(const float *__restrict__ act0, const float *__restrict__ act1,
                                                             SplitPlan p, float eps, unsigned *__restrict__ flag) {
    const int u = threadIdx.x;
    const int w = blockIdx.x;
    int z = blockIdx.y;                       // ((j * 2 + layer) * 2 + d) * 2 + point
    const int point = z & 1; z >>= 1;
    const int d = z & 1; z >>= 1;
    const int layer = z & 1; z >>= 1;
    const int j = z;
    const float *act = layer ? act1 : act0;
    const int a = p.core0[j + 1];
    const int t = d == 0 ? (a - 1 + point * (p.G / 2)) : (a - point * (p.G / 2));
    const float h0 = split_h(act, p.Tv, j * p.B + w, t - p.start[j], d, u);
    const float h1 = split_h(act, p.Tv, (j + 1) * p.B + w, t - p.start[j + 1], d, u);
    float delta = fabsf(h0 - h1);
    if (!(delta <= 4.0f)) delta = __builtin_inff();      // not a number
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) delta = fmaxf(delta, __shfl_xor(delta, o));
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&flag[1], __float_as_uint(delta));
        atomicMax(&flag[2 + blockIdx.y], __float_as_uint(delta));      // per (junction, layer, direction, point)
        if (!(delta <= eps)) atomicOr(&flag[0], 1u);
    }
}

}  // namespace mdk
