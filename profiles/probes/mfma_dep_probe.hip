// Cost of DEPENDENT v_mfma_f32_16x16x32_f16 chains on gfx950: cycles per MFMA when consecutive MFMAs accumulate into
// the same register (distance 1), alternate between 2, 3, 4, 6, 8 accumulators; one wave per SIMD and two.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_dep_probe mfma_dep_probe.hip && ./mfma_dep_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512) void k_chain(float *out, unsigned long long *cyc, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (i + 1)); }
    floatx4 acc[NACC];
    for (int k = 0; k < NACC; ++k) acc[k] = floatx4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 48 / NACC; ++rep)
#pragma unroll
            for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
static void run(int threads, int blocks) {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, blocks * 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_chain<NACC>), dim3(blocks), dim3(threads), 0, nullptr, out, cyc, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_chain<NACC>), dim3(blocks), dim3(threads), 0, nullptr, out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double per_wave = 48.0 * iters;                       // MFMAs per wave
    const int waves_per_simd = threads / 256;
    printf("accumulators %d, %d wave(s)/SIMD, %3d CUs: %.1f ns per MFMA per SIMD (%.2f us per 48), s_memtime ticks per MFMA per wave %.2f\n",
           NACC, waves_per_simd, blocks, ms * 1e6 / (per_wave * waves_per_simd), ms * 1e3 / iters, (double)c / per_wave);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int blocks : {1, 100, 256}) {
        for (int threads : {256, 512}) {
            run<1>(threads, blocks); run<2>(threads, blocks); run<3>(threads, blocks); run<4>(threads, blocks);
            run<6>(threads, blocks); run<8>(threads, blocks); run<12>(threads, blocks);
        }
    }
    return 0;
}
