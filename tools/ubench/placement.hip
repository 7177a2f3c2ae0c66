// placement.hip -- where do the two wavefronts of 1024 two-wave workgroups land?  (config 2's launch shape: 128 threads, ~29 KB of LDS,
// 88 VGPRs: four workgroups per CU.)  Every wavefront records HW_ID and XCC_ID while all are resident; the host counts, per SIMD, how
// many wave-0 ("team A") and wave-1 ("team B") wavefronts it holds.   hipcc --offload-arch=gfx950 -O2 -o placement placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <tuple>
#include <vector>
__global__ void __launch_bounds__(128) k(unsigned *out, int spin)
{
    extern __shared__ unsigned lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 3)" : "=s"(xcc));
    lds[threadIdx.x] = hw;
    unsigned long long t0 = clock64();
    while (clock64() - t0 < (unsigned long long)spin) { lds[threadIdx.x] += 1; }   // stay resident until all have started
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
}
int main()
{
    const int WG = 1024;
    unsigned *d; hipMalloc(&d, WG * 4 * sizeof(unsigned));
    hipLaunchKernelGGL(k, dim3(WG), dim3(128), 29 * 1024, 0, d, 2000000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(WG * 4); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<std::tuple<unsigned, unsigned, unsigned>, std::pair<int, int>> simd;   // (xcc, se/sh/cu, simd) -> (#A, #B)
    int same = 0;
    for (int b = 0; b < WG; b++) {
        unsigned key[2];
        for (int w = 0; w < 2; w++) {
            unsigned hw = h[(b * 2 + w) * 2], xcc = h[(b * 2 + w) * 2 + 1];
            unsigned cu = (hw >> 8) & 0x7f, sd = (hw >> 4) & 3;
            auto &e = simd[{xcc, cu, sd}];
            (w ? e.second : e.first)++;
            key[w] = (xcc << 16) | (cu << 2) | sd;
        }
        same += key[0] == key[1];
    }
    std::map<std::pair<int, int>, int> hist;
    for (auto &e : simd) hist[e.second]++;
    printf("%zu SIMDs hold wavefronts; workgroups with both wavefronts on ONE SIMD: %d of %d\n", simd.size(), same, WG);
    for (auto &e : hist) printf("  SIMDs with %d team-A and %d team-B wavefronts: %d\n", e.first.first, e.first.second, e.second);
    return 0;
}
