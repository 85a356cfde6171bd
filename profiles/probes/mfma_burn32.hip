// Sustained chip-wide rate of v_mfma_f32_32x32x16_f16 vs v_mfma_f32_16x16x32_f16 (both 16 KFLOP x 2 / 1 per issue slot),
// run for seconds on all SIMDs: is the larger shape cheaper in power (fewer operand reads per FLOP)?
//   ./mfma_burn32 <seconds> <shape 16|32> <waves per SIMD 1|2>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int SHAPE>
__global__ __launch_bounds__(512) void k_burn(float *out, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * ((threadIdx.x * 7 + i * 13) % 97)); b[i] = (_Float16)(0.002f * ((threadIdx.x * 3 + i) % 89)); }
    float s = 0.f;
    if constexpr (SHAPE == 16) {
        floatx4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k & 3], 0, 0, 0);
        s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    } else {
        floatx16 acc[2];
        for (int k = 0; k < 2; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k & 1], 0, 0, 0);   // 8 x 32 KFLOP = 16 x 16 KFLOP
        s = acc[0][0] + acc[1][5];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    const int shape = argc > 2 ? atoi(argv[2]) : 16, wps = argc > 3 ? atoi(argv[3]) : 1;
    const int blocks = 256, threads = 256 * wps;
    float *out; hipMalloc(&out, blocks * threads * 4);
    const int iters = 20000;
    auto t0 = std::chrono::steady_clock::now();
    long launches = 0; double el = 0;
    do {
        if (shape == 16) hipLaunchKernelGGL(k_burn<16>, dim3(blocks), dim3(threads), 0, nullptr, out, iters);
        else hipLaunchKernelGGL(k_burn<32>, dim3(blocks), dim3(threads), 0, nullptr, out, iters);
        hipDeviceSynchronize(); ++launches;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < seconds);
    const double flop = (double)launches * iters * 16 * 16384.0 * blocks * 4 * wps;      // 16 x (16 x 16 x 32 x 2 FLOP) per iteration and wave
    printf("shape %dx%d, %d wave(s)/SIMD, %.1f s: %.0f TFLOP/s = %.1f %% of 2500\n", shape, shape, wps, el, flop / el / 1e12, 100.0 * flop / el / 2.5e15);
    return 0;
}
