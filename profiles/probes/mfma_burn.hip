// A co-tenant for the recurrence: 4-wave work-groups (one wave per SIMD, ~20 VGPRs, no LDS) that issue back-to-back
// MFMAs for `seconds`; prints the MFMA rate it got.  Run next to `bench.py --device-only` to see how much matrix-pipe
// time the latency-bound recurrence leaves to a co-resident kernel, and what that costs the recurrence.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_burn mfma_burn.hip && ./mfma_burn <seconds> <blocks>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_burn(float *out, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (i + 1)); }
    floatx4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k & 3], 0, 0, 0);
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 5.0;
    const int blocks = argc > 2 ? atoi(argv[2]) : 256;
    float *out; hipMalloc(&out, blocks * 256 * 4);
    const int iters = 20000;                                   // 320 000 MFMAs per wave per launch
    auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    double el = 0;
    do {
        hipLaunchKernelGGL(k_burn, dim3(blocks), dim3(256), 0, nullptr, out, iters);
        hipDeviceSynchronize();
        ++launches;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < seconds);
    const double mfmas = (double)launches * iters * 16 * blocks * 4;            // wave-level MFMAs
    printf("burn: %d blocks x 4 waves for %.1f s: %.3e MFMAs/s = %.1f %% of the chip's 16-cycle issue rate at 2.4 GHz\n", blocks, el,
           mfmas / el, 100.0 * (mfmas / el) / (1024 * 2.4e9 / 16));
    return 0;
}
