// HBM layouts of the intermediates (DESIGN.md "data layout").
//
// Windows are grouped in tiles of 8.  Both intermediates are stored tile-major and, inside a
// (tile, t) block, in the register order of the recurrence kernel, so that every wave-level
// load/store of the serial kernel is one contiguous 256-byte run (the natural [window][t][f]
// layout made each of them four 64-byte runs 15 MB apart, and the kernel was bound by the number
// of cache lines its vector-memory instructions touched, not by bytes):
//
//   gi_t  [dir][tile][t][w8 8][q 2][lane 64][gate NG]  fp32   NG = 3 (GRU): 12 KB per block
//         (gate fastest: the recurrence reads the NG pre-activations of its (unit, window) with ONE 12- or
//          16-byte load per lane and the GEMM stores them with one: three dword loads 256 B apart cost 4.4 %
//          of the recurrence, profiles/r2_ablation.txt mask 32)
//   act_t [tile][t][dir][w8 8][q 2][lane 64]            fp32   D*1024 floats per block
//
// with  lane = g*16 + c,  window-in-tile = 2*g + q,  hidden unit = 16*w8 + c.
// Feature f of an activation row (f = dir*128 + unit) lives in chunk f>>4 = dir*8 + w8.
#pragma once
#include <stddef.h>

namespace mdk {

constexpr int kTileWin = 8;          // windows per tile
// NG = gate tiles per hidden unit: 3 for the GRU (r, z, n), 4 for the LSTM (i, f, g, o)
__host__ __device__ inline constexpr int gi_block_floats(int NG) { return 8 * 2 * NG * 64; }

__host__ __device__ inline size_t gi_block(int dir, int n_tiles, int tile, int T, int t, int NG) {
    return (((size_t)dir * n_tiles + tile) * T + t) * gi_block_floats(NG);
}
__host__ __device__ inline int gi_in_block(int w8, int q, int gate, int lane, int NG) {
    return ((w8 * 2 + q) * 64 + lane) * NG + gate;
}
__host__ __device__ inline size_t act_block(int D, int tile, int T, int t) {
    return ((size_t)tile * T + t) * (size_t)(D * 1024);
}
__host__ __device__ inline int act_in_block(int dir, int w8, int q, int lane) {
    return ((dir * 8 + w8) * 2 + q) * 64 + lane;
}

// Split scan (scan_split.hpp): a batch of B windows of T columns runs as S*B virtual windows of Tv columns.
constexpr int kMaxSplit = 16;
struct SplitPlan {
    int S = 1;                    // chunks per window (1 = not split)
    int B = 0, T = 0;             // the real batch
    int Tv = 0;                   // columns of a virtual window
    int G = 0;                    // margin
    int start[kMaxSplit];         // first real column of chunk k
    int core0[kMaxSplit + 1];     // real columns [core0[k], core0[k+1]) are delivered from chunk k
};

}  // namespace mdk
