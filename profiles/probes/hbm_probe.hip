// HBM streaming probe: what write / read / copy rates does this MI355X sustain for the footprints
// the engine uses (8.2 GB = gi of one layer at B=200, 30.7 GB at B=1000)?  Used to bound k_gi_gemm.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_probe hbm_probe.hip && ./hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_fill(float4 *p, size_t n) {
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
__global__ __launch_bounds__(256) void k_fill_nt(float4 *p, size_t n) {
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 w = {1.f, 2.f, 3.f, 4.f};
        __builtin_nontemporal_store(w, reinterpret_cast<f4 *>(p + i));
    }
}
__global__ __launch_bounds__(256) void k_read(const float4 *p, size_t n, float *out) {
    float4 a = make_float4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 v = p[i]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (a.x + a.y + a.z + a.w == 12345.f) *out = 1.f;
}
__global__ __launch_bounds__(256) void k_copy(const float4 *p, float4 *q, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) q[i] = p[i];
}

int main() {
    float *out; CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (double gb : {2.0, 8.2, 30.7}) {
        const size_t n = (size_t)(gb * 1e9 / 16);
        float4 *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
        for (int grid : {2048, 16384}) {
            float ms[4] = {0, 0, 0, 0};
            for (int rep = 0; rep < 3; ++rep) {   // last repetition is kept
                CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, a, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[0], e0, e1));
                CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_fill_nt, dim3(grid), dim3(256), 0, 0, a, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[1], e0, e1));
                CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[2], e0, e1));
                CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[3], e0, e1));
            }
            printf("%.1f GB grid %5d: fill %.2f TB/s  fill_nt %.2f TB/s  read %.2f TB/s  copy %.2f TB/s (r+w)\n", gb, grid,
                   gb / ms[0], gb / ms[1], gb / ms[2], 2 * gb / ms[3]);
        }
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
