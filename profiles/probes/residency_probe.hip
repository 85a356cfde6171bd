// How many 512-thread work-groups of a given register / LDS footprint does an MI355X keep resident at once?
// (round 3: the two-tile recurrence ran in TWO rounds above ~208 work-groups although the one-tile kernel, with the
// same grid and a similar footprint, runs 250 in one round.)  Every work-group records where it ran (XCC_ID, HW_ID)
// and when (s_memrealtime at entry / exit) and holds its CU for `ticks` of the 100 MHz clock.
//   hipcc --offload-arch=gfx950 -O2 -o residency_probe residency_probe.hip && ./residency_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

struct Rec { unsigned long long t0, t1; unsigned xcc, hwid; };

template <int VREG, int LDS>
__global__ __launch_bounds__(512, 2) void k_probe(Rec *out, unsigned long long ticks) {
    __shared__ unsigned char lds[LDS];
    lds[threadIdx.x] = (unsigned char)threadIdx.x;
    // reserve registers up to v<VREG>
    if constexpr (VREG == 191) asm volatile("v_mov_b32 v191, 0" ::: "v191");
    if constexpr (VREG == 207) asm volatile("v_mov_b32 v207, 0" ::: "v207");
    if constexpr (VREG == 227) asm volatile("v_mov_b32 v227, 0" ::: "v227");
    if constexpr (VREG == 239) asm volatile("v_mov_b32 v239, 0" ::: "v239");
    if constexpr (VREG == 247) asm volatile("v_mov_b32 v247, 0" ::: "v247");
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    __syncthreads();
    if (threadIdx.x == 0) {
        Rec r;
        r.t0 = t0; r.t1 = __builtin_amdgcn_s_memrealtime();
        r.xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;           // HW_REG_XCC_ID[3:0]
        r.hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);                // HW_REG_HW_ID
        out[blockIdx.y * gridDim.x + blockIdx.x] = r;
        if (lds[7] == 255) out[0].t0 = 0;
    }
}

template <int VREG, int LDS>
static void run(int gx, int gy) {
    const int n = gx * gy;
    Rec *d = nullptr;
    hipMalloc(&d, n * sizeof(Rec));
    hipLaunchKernelGGL((k_probe<VREG, LDS>), dim3(gx, gy), dim3(512), 0, nullptr, d, 100000ull);   // 1 ms
    hipDeviceSynchronize();
    std::vector<Rec> r(n);
    hipMemcpy(r.data(), d, n * sizeof(Rec), hipMemcpyDeviceToHost);
    hipFree(d);
    unsigned long long first = ~0ull, last = 0;
    for (auto &x : r) { first = std::min(first, x.t0); last = std::max(last, x.t1); }
    int late = 0, per_xcc[16] = {0}, late_xcc[16] = {0};
    for (auto &x : r) { per_xcc[x.xcc & 15]++; if (x.t0 - first > 50000ull) { late++; late_xcc[x.xcc & 15]++; } }   // started > 0.5 ms late
    // distinct (xcc, se, cu) used
    std::vector<unsigned> cus;
    for (auto &x : r) cus.push_back((x.xcc << 16) | (x.hwid & 0xff00));   // cu_id[11:8], sh_id[12], se_id[15:13]
    std::sort(cus.begin(), cus.end());
    const int distinct = (int)(std::unique(cus.begin(), cus.end()) - cus.begin());
    printf("VGPR<=%3d LDS %5d grid %3d x %d = %3d WGs: span %.2f ms, %3d started late (second round), distinct CUs used %3d | per XCC:",
           VREG + 1, LDS, gx, gy, n, (last - first) / 100000.0, late, distinct);
    for (int i = 0; i < 8; ++i) printf(" %d(%d)", per_xcc[i], late_xcc[i]);
    printf("\n");
}

int main() {
    for (int gx : {100, 104, 107, 113, 120, 125, 128}) run<227, 17408>(gx, 2);
    for (int gx : {107, 125, 128}) run<191, 8704>(gx, 2);
    for (int gx : {107, 125}) run<227, 8704>(gx, 2);
    for (int gx : {107, 125}) run<191, 17408>(gx, 2);
    for (int gx : {107, 125}) run<207, 17408>(gx, 2);
    for (int gx : {125}) { run<239, 8704>(gx, 2); run<247, 8704>(gx, 2); }
    run<227, 17408>(214, 1); run<227, 17408>(250, 1); run<191, 8704>(250, 1);
    return 0;
}
