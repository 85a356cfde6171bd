// 2-D DMA copies of SHORT rows between page-locked host memory and the device: what a column slab of a (200, 10000, F)
// batch costs as a function of its width (round 4: can the split scan's host path stream slabs by DMA?).
//   hipcc --offload-arch=gfx950 -O3 -o dma2d_probe dma2d_probe.hip && ./dma2d_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const int B = 200, T = 10000;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int E : {40, 20}) {                       // bytes per column: x (10 floats) in, probabilities (5 floats) out
        const size_t N = (size_t)B * T * E;
        char *dev, *pin;
        CK(hipMalloc(&dev, N));
        CK(hipHostMalloc((void **)&pin, N, hipHostMallocDefault));
        memset(pin, 1, N);
        for (int cols : {36, 72, 144, 280, 568, 1128, 2000, 10000}) {
            for (int dir = 0; dir < 2; ++dir) {
                if ((E == 40) != (dir == 0)) continue;          // x travels in, probabilities out
                // n_ranges ranges of `cols` columns per window, spread over the window: one 2-D copy each
                const int n_ranges = cols >= 10000 ? 1 : (cols > 1000 ? 5 : 10);
                double best = 1e30;
                for (int rep = 0; rep < 5; ++rep) {
                    CK(hipStreamSynchronize(s));
                    const double t0 = now_ms();
                    for (int r = 0; r < n_ranges; ++r) {
                        const size_t off = (size_t)(r * (T / n_ranges)) * E;
                        if (dir == 0) CK(hipMemcpy2DAsync(dev + off, (size_t)T * E, pin + off, (size_t)T * E, (size_t)cols * E, B, hipMemcpyHostToDevice, s));
                        else CK(hipMemcpy2DAsync(pin + off, (size_t)T * E, dev + off, (size_t)T * E, (size_t)cols * E, B, hipMemcpyDeviceToHost, s));
                    }
                    const double t1 = now_ms();
                    CK(hipStreamSynchronize(s));
                    const double t2 = now_ms();
                    if (t2 - t0 < best) { best = t2 - t0; if (rep == 4 || true) {} }
                    if (rep == 4) printf("%s rows of %6d B (%4d columns) x %4d rows: %.3f ms = %.1f GB/s, %.3f us per row  (enqueue %.3f ms)\n",
                                         dir == 0 ? "H2D" : "D2H", cols * E, cols, n_ranges * B, best, 1e-6 * n_ranges * B * cols * E / best,
                                         1e3 * best / (n_ranges * B), t1 - t0);
                }
            }
        }
        CK(hipFree(dev)); CK(hipHostFree(pin));
    }
    return 0;
}
