// Host <-> device transfer probe for the predict_on_batch staging design (DESIGN.md "host path").
// Sizes are one BASELINE configs[1] batch: x = 200 x 10000 x 10 fp32 (80 MB), probs = 200 x 10000 x 5 fp32 (40 MB).
//   hipcc --offload-arch=gfx950 -O3 -pthread -o host_path_probe host_path_probe.hip && ./host_path_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

__global__ __launch_bounds__(256) void k_copy16(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

static void par_memcpy(char *dst, const char *src, size_t n, int nt) {
    std::vector<std::thread> th;
    const size_t per = (n / nt + 4095) / 4096 * 4096;
    for (int t = 0; t < nt; ++t) {
        const size_t o = (size_t)t * per;
        if (o >= n) break;
        const size_t len = std::min(per, n - o);
        th.emplace_back([=] { memcpy(dst + o, src + o, len); });
    }
    for (auto &t : th) t.join();
}

int main() {
    const size_t NX = (size_t)200 * 10000 * 10 * 4, NP = (size_t)200 * 10000 * 5 * 4;
    printf("host threads: %u\n", std::thread::hardware_concurrency());
    char *dx, *dp;
    CK(hipMalloc(&dx, NX)); CK(hipMalloc(&dp, NP));
    char *pg_x = (char *)malloc(NX), *pg_p = (char *)malloc(NP);
    memset(pg_x, 1, NX); memset(pg_p, 2, NP);
    char *pin_x, *pin_p;
    double t0 = now_ms();
    CK(hipHostMalloc((void **)&pin_x, NX, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&pin_p, NP, hipHostMallocDefault));
    printf("hipHostMalloc 80+40 MB: %.2f ms\n", now_ms() - t0);
    memset(pin_x, 1, NX); memset(pin_p, 2, NP);
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));

    auto timeit = [&](const char *what, size_t bytes, auto fn) {
        double best = 1e30;
        for (int r = 0; r < 4; ++r) {
            CK(hipDeviceSynchronize());
            const double a = now_ms();
            fn();
            CK(hipDeviceSynchronize());
            best = std::min(best, now_ms() - a);
        }
        printf("%-62s %8.3f ms  %7.2f GB/s\n", what, best, bytes / best / 1e6);
    };
    timeit("H2D 80 MB pageable hipMemcpyAsync", NX, [&] { CK(hipMemcpyAsync(dx, pg_x, NX, hipMemcpyHostToDevice, s)); });
    timeit("D2H 40 MB pageable hipMemcpyAsync", NP, [&] { CK(hipMemcpyAsync(pg_p, dp, NP, hipMemcpyDeviceToHost, s)); });
    timeit("H2D 80 MB pinned hipMemcpyAsync", NX, [&] { CK(hipMemcpyAsync(dx, pin_x, NX, hipMemcpyHostToDevice, s)); });
    timeit("D2H 40 MB pinned hipMemcpyAsync", NP, [&] { CK(hipMemcpyAsync(pin_p, dp, NP, hipMemcpyDeviceToHost, s)); });
    timeit("H2D 80 MB + D2H 40 MB pinned, two streams (full duplex)", NX + NP, [&] {
        CK(hipMemcpyAsync(dx, pin_x, NX, hipMemcpyHostToDevice, s));
        CK(hipMemcpyAsync(pin_p, dp, NP, hipMemcpyDeviceToHost, s2)); });
    for (int chunks : {8, 32}) {
        char buf[128];
        snprintf(buf, sizeof buf, "H2D 80 MB pinned in %d chunks", chunks);
        timeit(buf, NX, [&] { for (int c = 0; c < chunks; ++c) CK(hipMemcpyAsync(dx + NX / chunks * c, pin_x + NX / chunks * c, NX / chunks, hipMemcpyHostToDevice, s)); });
    }
    // strided time slabs: 200 rows of (T/8 columns x 40 B) with the natural pitch
    {
        const size_t pitch = 10000 * 40, width = 1250 * 40;
        timeit("H2D 2D pinned -> device, 8 slabs of 200 x 50 KB (pitch 400 KB)", NX, [&] {
            for (int c = 0; c < 8; ++c) CK(hipMemcpy2DAsync(dx + c * width, pitch, pin_x + c * width, pitch, width, 200, hipMemcpyHostToDevice, s)); });
        timeit("H2D 2D pageable -> device, same slabs", NX, [&] {
            for (int c = 0; c < 8; ++c) CK(hipMemcpy2DAsync(dx + c * width, pitch, pg_x + c * width, pitch, width, 200, hipMemcpyHostToDevice, s)); });
        const size_t ppitch = 10000 * 20, pwidth = 1250 * 20;
        timeit("D2H 2D device -> pinned, 8 slabs of 200 x 25 KB (pitch 200 KB)", NP, [&] {
            for (int c = 0; c < 8; ++c) CK(hipMemcpy2DAsync(pin_p + c * pwidth, ppitch, dp + c * pwidth, ppitch, pwidth, 200, hipMemcpyDeviceToHost, s)); });
        timeit("D2H 2D device -> pageable, same slabs", NP, [&] {
            for (int c = 0; c < 8; ++c) CK(hipMemcpy2DAsync(pg_p + c * pwidth, ppitch, dp + c * pwidth, ppitch, pwidth, 200, hipMemcpyDeviceToHost, s)); });
    }
    // kernels reading / writing pinned host memory directly (zero copy over PCIe)
    {
        char *dev_view_x, *dev_view_p;
        CK(hipHostGetDevicePointer((void **)&dev_view_x, pin_x, 0));
        CK(hipHostGetDevicePointer((void **)&dev_view_p, pin_p, 0));
        for (int grid : {64, 256, 1024}) {
            char buf[128];
            snprintf(buf, sizeof buf, "kernel reads pinned host 80 MB -> HBM, grid %d", grid);
            timeit(buf, NX, [&] { hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, s, (const float4 *)dev_view_x, (float4 *)dx, NX / 16); });
            snprintf(buf, sizeof buf, "kernel writes HBM 40 MB -> pinned host, grid %d", grid);
            timeit(buf, NP, [&] { hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, s, (const float4 *)dp, (float4 *)dev_view_p, NP / 16); });
        }
    }
    // registering the caller's pageable buffer
    for (int r = 0; r < 2; ++r) {
        double a = now_ms();
        CK(hipHostRegister(pg_x, NX, hipHostRegisterDefault));
        double b = now_ms();
        CK(hipMemcpyAsync(dx, pg_x, NX, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
        double c = now_ms();
        CK(hipHostUnregister(pg_x));
        double d = now_ms();
        printf("hipHostRegister 80 MB: %.2f ms, copy %.2f ms, unregister %.2f ms\n", b - a, c - b, d - c);
    }
    // CPU staging copies pageable -> pinned
    for (int nt : {1, 2, 4, 8, 16}) {
        double best = 1e30;
        for (int r = 0; r < 4; ++r) { double a = now_ms(); par_memcpy(pin_x, pg_x, NX, nt); best = std::min(best, now_ms() - a); }
        printf("memcpy pageable -> pinned 80 MB, %2d threads (spawn included): %7.3f ms %6.2f GB/s\n", nt, best, NX / best / 1e6);
    }
    for (int nt : {1, 4}) {
        double best = 1e30;
        for (int r = 0; r < 4; ++r) { double a = now_ms(); par_memcpy(pg_p, pin_p, NP, nt); best = std::min(best, now_ms() - a); }
        printf("memcpy pinned -> pageable 40 MB, %2d threads: %7.3f ms %6.2f GB/s\n", nt, best, NP / best / 1e6);
    }
    // first-touch cost of a fresh pageable output buffer (what np.empty / torch.empty hands over)
    {
        double a = now_ms();
        char *fresh = (char *)malloc(NP);
        CK(hipMemcpyAsync(fresh, dp, NP, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        printf("D2H 40 MB into a FRESH malloc (page faults included): %.2f ms\n", now_ms() - a);
        double b = now_ms();
        par_memcpy(fresh, pin_p, NP, 1);
        printf("memcpy pinned -> same buffer again: %.2f ms\n", now_ms() - b);
        free(fresh);
        a = now_ms();
        fresh = (char *)malloc(NP);
        par_memcpy(fresh, pin_p, NP, 4);
        printf("memcpy pinned -> FRESH malloc 40 MB, 4 threads: %.2f ms\n", now_ms() - a);
        free(fresh);
    }
    return 0;
}
