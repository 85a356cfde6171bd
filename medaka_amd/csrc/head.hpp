// Classifier head and the small element-wise models.
//   k_linear_softmax : nn.Linear(D*H -> 5) + softmax(dim=-1)      (reference gru.py:67-71)
//   k_majority       : MajorityVoteModel.forward                   (majority_vote_model.py:37-53)
// Both are HBM-streaming kernels: one 1 KB row in, 20 B out.
#pragma once
#include "common.hpp"
#include "layout.hpp"

namespace mdk {

// 16 lanes share one row: lane j holds float4 chunks j, j+16, j+32, j+48 of the row, the five
// partial dot products are xor-reduced over the 16 lanes, lanes 0..4 each write one class.
template <int NCH>   // row width = 64 * NCH floats  (NCH = 4 for bidirectional H=128)
__global__ __launch_bounds__(256) void k_linear_softmax(
    const float *__restrict__ hin,    // [M][64*NCH]
    const float *__restrict__ lin_w,  // [5][64*NCH]
    const float *__restrict__ lin_b,  // [5]
    float *__restrict__ probs,        // [M][5]
    long M, int normalise)
{
    constexpr int W = 64 * NCH;
    const int lane = threadIdx.x & 63;
    const int sub = lane & 15;
    const int rsub = lane >> 4;
    float4 wv[5][NCH];
#pragma unroll
    for (int cl = 0; cl < 5; ++cl)
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
            wv[cl][ch] = *reinterpret_cast<const float4 *>(lin_w + cl * W + (ch * 16 + sub) * 4);
    float bv[5];
#pragma unroll
    for (int cl = 0; cl < 5; ++cl) bv[cl] = lin_b[cl];

    const long wave_global = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long n_waves = (long)gridDim.x * (blockDim.x >> 6);
    for (long r4 = wave_global * 4; r4 < M; r4 += n_waves * 4) {
        const long row = r4 + rsub;
        const bool ok = row < M;
        const float *hr = hin + (ok ? row : (M - 1)) * W;
        float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const float4 v = *reinterpret_cast<const float4 *>(hr + (ch * 16 + sub) * 4);
#pragma unroll
            for (int cl = 0; cl < 5; ++cl) {
                acc[cl] = fmaf(v.x, wv[cl][ch].x, acc[cl]);
                acc[cl] = fmaf(v.y, wv[cl][ch].y, acc[cl]);
                acc[cl] = fmaf(v.z, wv[cl][ch].z, acc[cl]);
                acc[cl] = fmaf(v.w, wv[cl][ch].w, acc[cl]);
            }
        }
#pragma unroll
        for (int cl = 0; cl < 5; ++cl) {
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) acc[cl] += __shfl_xor(acc[cl], off, 16);
            acc[cl] += bv[cl];
        }
        float res[5];
        if (normalise) {
            float mx = acc[0];
#pragma unroll
            for (int cl = 1; cl < 5; ++cl) mx = fmaxf(mx, acc[cl]);
            float sum = 0.f;
#pragma unroll
            for (int cl = 0; cl < 5; ++cl) { res[cl] = __expf(acc[cl] - mx); sum += res[cl]; }
#pragma unroll
            for (int cl = 0; cl < 5; ++cl) res[cl] = res[cl] / sum;
        } else {
#pragma unroll
            for (int cl = 0; cl < 5; ++cl) res[cl] = acc[cl];
        }
        if (ok && sub < 5) {
            float v = res[0];
            v = sub == 1 ? res[1] : v;
            v = sub == 2 ? res[2] : v;
            v = sub == 3 ? res[3] : v;
            v = sub == 4 ? res[4] : v;
            probs[row * 5 + sub] = v;
        }
    }
}

// Same head over the tile-major activation layout (layout.hpp) written by the recurrence kernel.
// One wave per (tile, t) block = the rows of 8 windows.  Lane l reads float4 number i*64 + l of
// the block for i = 0..DIN*4-1: it always sees window (g, q) = ((l>>2)&3, (l>>4)&1) and features
// 16*(2i + (l>>5)) + 4*(l&3) .. +3; the 8 lanes of a window are xor-reduced (masks 1, 2, 32).
// Probabilities go out in the reference's natural (B, T, 5) order.
// SPLIT (split scan, scan_split.hpp): the B windows are virtual ones -- chunk k = win / sp.B of real window win % sp.B,
// column t = real column sp.start[k] + t; only a chunk's own columns [core0[k], core0[k+1]) are computed and they go
// straight to their place in the real (sp.B, sp.T, 5) result.
template <int DIN, bool SPLIT = false>
__global__ __launch_bounds__(256) void k_head_tiled(
    const float *__restrict__ act,    // act_t of the last layer
    const float *__restrict__ lin_w,  // [5][DIN*128]
    const float *__restrict__ lin_b,  // [5]
    float *__restrict__ probs,        // [B][T][5]
    int B, int T, int n_tiles, int normalise,
    int t0, int nt,                   // columns [t0, t0 + nt) of every window
    SplitPlan sp)
{
    constexpr int F = DIN * 128;
    __shared__ __attribute__((aligned(16))) float wl[5 * F];
    for (int i = threadIdx.x; i < 5 * F; i += blockDim.x) wl[i] = lin_w[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int c4 = lane & 3, g = (lane >> 2) & 3, q = (lane >> 4) & 1, hi = lane >> 5;
    float bv[5];
#pragma unroll
    for (int cl = 0; cl < 5; ++cl) bv[cl] = lin_b[cl];
    const long n_blocks = (long)n_tiles * nt;
    const long wave_global = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long n_waves = (long)gridDim.x * (blockDim.x >> 6);
    for (long blk = wave_global; blk < n_blocks; blk += n_waves) {
        const int tile = (int)(blk / nt), t = t0 + (int)(blk % nt);
        const int win = tile * kTileWin + 2 * g + q;
        float *dst = probs + ((size_t)win * T + t) * 5;
        bool deliver = win < B;
        if constexpr (SPLIT) {
            const int k = win / sp.B, tr = sp.start[min(k, sp.S - 1)] + t;
            deliver = deliver && tr >= sp.core0[min(k, sp.S - 1)] && tr < sp.core0[min(k, sp.S - 1) + 1];
            dst = probs + ((size_t)(win - k * sp.B) * sp.T + tr) * 5;
            if (!__any(deliver)) continue;        // a margin column of all 8 windows
        }
        const float *src = act + ((size_t)tile * T + t) * (DIN * 1024) + 4 * lane;
        float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < DIN * 4; ++i) {
            const float4 v = *reinterpret_cast<const float4 *>(src + i * 256);
            const int f0 = 16 * (2 * i + hi) + 4 * c4;
#pragma unroll
            for (int cl = 0; cl < 5; ++cl) {
                const float4 wv = *reinterpret_cast<const float4 *>(&wl[cl * F + f0]);
                acc[cl] = fmaf(v.x, wv.x, acc[cl]);
                acc[cl] = fmaf(v.y, wv.y, acc[cl]);
                acc[cl] = fmaf(v.z, wv.z, acc[cl]);
                acc[cl] = fmaf(v.w, wv.w, acc[cl]);
            }
        }
#pragma unroll
        for (int cl = 0; cl < 5; ++cl) {
            acc[cl] += __shfl_xor(acc[cl], 1);
            acc[cl] += __shfl_xor(acc[cl], 2);
            acc[cl] += __shfl_xor(acc[cl], 32);
            acc[cl] += bv[cl];
        }
        if (c4 == 0 && hi == 0 && deliver) {
            float res[5];
            if (normalise) {
                float mx = acc[0];
#pragma unroll
                for (int cl = 1; cl < 5; ++cl) mx = fmaxf(mx, acc[cl]);
                float sum = 0.f;
#pragma unroll
                for (int cl = 0; cl < 5; ++cl) { res[cl] = __expf(acc[cl] - mx); sum += res[cl]; }
#pragma unroll
                for (int cl = 0; cl < 5; ++cl) res[cl] = res[cl] / sum;
            } else {
#pragma unroll
                for (int cl = 0; cl < 5; ++cl) res[cl] = acc[cl];
            }
#pragma unroll
            for (int cl = 0; cl < 5; ++cl) dst[cl] = res[cl];
        }
    }
}

// Head of a last layer whose Linear ran inside the recurrence kernel (rec_fused.hpp HEAD): lpart holds, per direction,
// the 5 partial logits of the 8 windows of every (tile, t) block.  logits = part_0 (+ part_1) + bias, softmax as above,
// probabilities out in the reference's natural (B, T, 5) order; SPLIT as in k_head_tiled.  One thread per (block, window).
template <bool SPLIT>
static __global__ __launch_bounds__(256) void k_head_combine(
    const float *__restrict__ lpart,  // [D][n_tiles][T][8][5]
    const float *__restrict__ lin_b,  // [5]
    float *__restrict__ probs, int B, int T, int n_tiles, int D, int normalise, int t0, int nt, SplitPlan sp)
{
    const long total = (long)n_tiles * nt * kTileWin;
    const size_t dir_stride = (size_t)n_tiles * T * 40;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int w = (int)(i & 7);
        const long blk = i >> 3;
        const int tile = (int)(blk / nt), t = t0 + (int)(blk % nt);
        const int win = tile * kTileWin + w;
        if (win >= B) continue;
        float *dst = probs + ((size_t)win * T + t) * 5;
        if constexpr (SPLIT) {
            const int k = win / sp.B, tr = sp.start[k] + t;
            if (tr < sp.core0[k] || tr >= sp.core0[k + 1]) continue;
            dst = probs + ((size_t)(win - k * sp.B) * sp.T + tr) * 5;
        }
        const float *src = lpart + ((size_t)tile * T + t) * 40 + w * 5;
        float acc[5];
#pragma unroll
        for (int cl = 0; cl < 5; ++cl) {
            float v = src[cl];
            if (D == 2) v += src[dir_stride + cl];
            acc[cl] = v + lin_b[cl];
        }
        float res[5];
        if (normalise) {
            float mx = acc[0];
#pragma unroll
            for (int cl = 1; cl < 5; ++cl) mx = fmaxf(mx, acc[cl]);
            float sum = 0.f;
#pragma unroll
            for (int cl = 0; cl < 5; ++cl) { res[cl] = __expf(acc[cl] - mx); sum += res[cl]; }
#pragma unroll
            for (int cl = 0; cl < 5; ++cl) res[cl] = res[cl] / sum;
        } else {
#pragma unroll
            for (int cl = 0; cl < 5; ++cl) res[cl] = acc[cl];
        }
#pragma unroll
        for (int cl = 0; cl < 5; ++cl) dst[cl] = res[cl];
    }
}

// channels a c g t A C G T d D -> classes [d+D, a+A, c+C, g+G, t+T]; class 0 += 1 - sum
static __global__ __launch_bounds__(256) void k_majority(const float *__restrict__ x,
                                                  float *__restrict__ probs, long n_cols)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cols) return;
    const float *r = x + i * 10;
    float p[5];
    p[0] = r[8] + r[9];
#pragma unroll
    for (int c = 0; c < 4; ++c) p[1 + c] = r[c] + r[4 + c];
    float s = p[0];
#pragma unroll
    for (int c = 1; c < 5; ++c) s += p[c];
    p[0] += 1.0f - s;
#pragma unroll
    for (int c = 0; c < 5; ++c) probs[i * 5 + c] = p[c];
}

// CountsFeatureEncoder(normalise='total') on the device (reference medaka/features.py:907-911):
//     feature = (counts / np.maximum(1, depth)).astype(float32)
// numpy divides the integer arrays in float64 and then rounds to float32; the same two IEEE
// roundings are done here (v_div f64, v_cvt_f32_f64), so the result is bit-identical.
static __global__ __launch_bounds__(256) void k_normalise_counts(const unsigned short *__restrict__ counts,
                                                                 const unsigned int *__restrict__ depth,
                                                                 float *__restrict__ x, long n_cols, int F)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cols * F) return;
    const unsigned int d = depth[i / F];
    x[i] = (float)((double)counts[i] / (double)(d > 1u ? d : 1u));
}

// argmax decode (reference medaka/labels.py:1061-1065): most probable class -- the FIRST maximum,
// as numpy.argmax -- and its probability; 5 bytes per column leave the device instead of 20.
static __global__ __launch_bounds__(256) void k_decode(const float *__restrict__ probs, unsigned char *__restrict__ cls,
                                                       float *__restrict__ pmax, long n_cols, int C)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cols) return;
    const float *r = probs + i * C;
    int best = 0;
    float bv = r[0];
    for (int c = 1; c < C; ++c) {
        const float v = r[c];
        if (bv == bv && (v > bv || v != v)) { bv = v; best = c; }   // numpy.argmax: the first NaN wins
    }
    cls[i] = (unsigned char)best;
    pmax[i] = bv;
}

// The last result chunks of a split host call, written to the caller's (page-locked, device-visible) buffer by the CUs.
// The scan's second half produces probabilities about as fast as one DMA queue ships them as 2-D copies (half precision:
// 40 MB in 1.1 ms), so when the last recurrence kernel ends the copies of its last launches are still queued: 0.45 ms of
// tail on a 5.7 ms call (profiles/r6_experiments/README.md section 1).  At that moment the CUs have nothing to do -- a copy
// kernel that talks to host memory BESIDE a recurrence stalls it (profiles/r4_experiments/README.md), behind the last one it
// stalls nothing -- and a kernel writes pinned host memory at the full PCIe rate (54 GB/s, profiles/r2_host_path_probe.txt).
// Block (w, j): window w, range r = j / S of chunk k = j % S: the real columns core_k /\ (start[k] + [t0[r], t0[r] + nt[r])).
struct TailRanges { int n; int t0[4]; int nt[4]; };
static __global__ __launch_bounds__(256) void k_tail_to_host(const float *__restrict__ probs, float *__restrict__ host, SplitPlan sp,
                                                             TailRanges tr, int C) {
    const int w = blockIdx.x, r = blockIdx.y / sp.S, k = blockIdx.y % sp.S;
    const int a = max(sp.core0[k], sp.start[k] + tr.t0[r]), b = min(sp.core0[k + 1], sp.start[k] + tr.t0[r] + tr.nt[r]);
    if (a >= b) return;
    const size_t base = ((size_t)w * sp.T + a) * C;
    const int n = (b - a) * C;
    for (int i = threadIdx.x; i < n; i += blockDim.x) host[base + i] = probs[base + i];
}

}  // namespace mdk
