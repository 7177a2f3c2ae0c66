// simd_map.hip -- where do the wavefronts of a 256-thread workgroup land?  Prints, for wave
// index w = 0..3 of every workgroup, the histogram of the hardware SIMD it ran on (HW_ID
// bits 5:4), plus the number of distinct SIMDs per workgroup.
// hipcc -O2 --offload-arch=gfx950 simd_map.hip -o simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) probe(unsigned *out, int spin)
{
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4); // HW_REG_HW_ID
    double v = threadIdx.x;
    for (int i = 0; i < spin; i++) v = v * 1.0000001 + 1e-9; // stay resident for a while
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = hw | (v == 12345.0 ? 1u << 31 : 0);
}

int main()
{
    const int nb = 4096;
    unsigned *d;
    hipMalloc(&d, nb * 4 * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 4);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int hist[4][4] = {}, distinct[5] = {};
    for (int b = 0; b < nb; b++) {
        unsigned seen = 0;
        for (int w = 0; w < 4; w++) {
            const int simd = (h[b * 4 + w] >> 4) & 3;
            hist[w][simd]++;
            seen |= 1u << simd;
        }
        distinct[__builtin_popcount(seen)]++;
    }
    for (int w = 0; w < 4; w++)
        printf("wave %d: SIMD0 %d SIMD1 %d SIMD2 %d SIMD3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("workgroups by number of distinct SIMDs: 1:%d 2:%d 3:%d 4:%d\n", distinct[1], distinct[2], distinct[3], distinct[4]);
    printf("first workgroups (cu,simd per wave):");
    for (int b = 0; b < 8; b++) {
        printf("  [");
        for (int w = 0; w < 4; w++) printf(" cu%u/s%u", (h[b * 4 + w] >> 8) & 15, (h[b * 4 + w] >> 4) & 3);
        printf(" ]");
    }
    printf("\n");
    return 0;
}
