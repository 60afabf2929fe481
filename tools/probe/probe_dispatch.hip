// Where does the dispatcher put the workgroups of a "two per CU" launch (256 threads, 80 KiB LDS, <= 256 VGPRs), and when do they start?
// Each workgroup records {XCC_ID, HW_ID, start, end} (wall clock, 100 MHz) and spins ~8 us.  Prints, per XCD, the CU (SE/SH/CU id) of the first
// 80 workgroups in dispatch order and whether co-resident pairs are consecutive blocks or blocks 32 apart.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/probe_dispatch.hip -o /tmp/probe_dispatch && /tmp/probe_dispatch [grid]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 2) void k(unsigned long long* out, int spin) {
    extern __shared__ char smem[];
    const unsigned long long t0 = wall_clock64();
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
    unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
    smem[threadIdx.x] = (char)threadIdx.x;
    while (wall_clock64() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = xcc; out[blockIdx.x * 4 + 1] = hw; out[blockIdx.x * 4 + 2] = t0; out[blockIdx.x * 4 + 3] = wall_clock64();
    }
}
int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 780;
    unsigned long long* d; hipMalloc(&d, grid * 32);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(grid), dim3(256), 81920, 0, d, 800); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), d, grid * 32, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull;
    for (int b = 0; b < grid; ++b) tmin = h[b * 4 + 2] < tmin ? h[b * 4 + 2] : tmin;
    printf("block xcc se sh cu simd wave start_us end_us\n");
    for (int b = 0; b < grid && b < 8 * 100; ++b) {
        if ((b & 7) != 0) continue;                                // XCD 0's blocks in dispatch order
        unsigned hw = (unsigned)h[b * 4 + 1];
        printf("%4d %u %u %u %2u %u %2u %7.2f %7.2f\n", b, (unsigned)h[b * 4] & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15,
               (h[b * 4 + 2] - tmin) / 100.0, (h[b * 4 + 3] - tmin) / 100.0);
    }
    // co-residency: for each block, which other blocks overlapped in time on the same (xcc, se, sh, cu)
    std::map<unsigned, std::vector<int>> by_cu;
    for (int b = 0; b < grid; ++b) by_cu[((unsigned)h[b * 4] & 15) << 16 | (((unsigned)h[b * 4 + 1] >> 8) & 0xff)].push_back(b);
    int hist[6] = {0, 0, 0, 0, 0, 0};
    for (auto& kv : by_cu) {
        auto& v = kv.second;
        for (size_t i = 0; i + 1 < v.size(); i += 2) {
            const int dlt = (v[i + 1] - v[i]) >> 3;
            hist[dlt == 1 ? 0 : dlt == 32 ? 1 : dlt < 32 ? 2 : 3]++;
        }
    }
    printf("CUs seen: %zu; first-pair index distance (in per-XCD dispatch order): ==1: %d, ==32: %d, other<32: %d, other: %d\n", by_cu.size(), hist[0], hist[1], hist[2], hist[3]);
    return 0;
}
