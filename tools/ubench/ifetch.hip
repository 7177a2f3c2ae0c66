// ifetch.hip -- is a long straight-line kernel bound by instruction fetch?  Every wave runs a
// 2048-instruction unrolled body (8 KB of 4-byte VOP2 or 16 KB of 8-byte VOP3 instructions),
// 64 times; 4 waves per SIMD.  "staggered": each wave first spins for a different time, so
// that the 32 waves behind one instruction cache sit at different addresses of the body.
// hipcc -O2 --offload-arch=gfx950 ifetch.hip -o ifetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define R8(x) x x x x x x x x
#define R64(x) R8(R8(x))
#define R2048(x) R8(R8(R8(x))) R8(R8(R8(x))) R8(R8(R8(x))) R8(R8(R8(x)))

template <int OP, int STAG> __global__ void __launch_bounds__(1024) k(float *out, float cf, long long *cyc)
{
    float f0 = threadIdx.x, f1 = 1, f2 = 2, f3 = 3, f4 = 4, f5 = 5, f6 = 6, f7 = 7;
    if (STAG) { // desynchronise: wave w of block b spins (w*37 + b*11) % 64 * ~200 cycles
        const int w = threadIdx.x >> 6;
        int n = ((w * 37 + blockIdx.x * 11) % 64) * 6;
        for (int i = 0; i < n; i++) __builtin_amdgcn_s_sleep(8);
    }
    const long long t0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < 64; it++) {
        if (OP == 0) asm volatile(R2048("v_add_f32_e32 %0, %8, %0\n v_add_f32_e32 %1, %8, %1\n v_add_f32_e32 %2, %8, %2\n v_add_f32_e32 %3, %8, %3\n v_add_f32_e32 %4, %8, %4\n v_add_f32_e32 %5, %8, %5\n v_add_f32_e32 %6, %8, %6\n v_add_f32_e32 %7, %8, %7\n") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "s"(cf));
        if (OP == 1) asm volatile(R2048("v_fma_f32 %0, %8, %0, %0\n v_fma_f32 %1, %8, %1, %1\n v_fma_f32 %2, %8, %2, %2\n v_fma_f32 %3, %8, %3, %3\n v_fma_f32 %4, %8, %4, %4\n v_fma_f32 %5, %8, %5, %5\n v_fma_f32 %6, %8, %6, %6\n v_fma_f32 %7, %8, %7, %7\n") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "s"(cf));
    }
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP, int STAG> void run(const char *name, int bytes)
{
    const int blocks = 256, threads = 1024;
    float *out; long long *cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads); hipMalloc(&cyc, sizeof(long long) * blocks * 16);
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL((k<OP, STAG>), dim3(blocks), dim3(threads), 0, 0, out, 1.0000001f, cyc);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 16); hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 16, hipMemcpyDeviceToHost);
    double avr = 0; for (int i = 0; i < blocks * 16; i++) avr += h[i]; avr /= blocks * 16;
    const double n = 64.0 * 2048 * 8;                 // instructions per wave
    const double ns_wave = avr * 10.0 / n;            // one wave's time per instruction
    printf("%-46s %6.3f ns per instr per wave, %6.3f per SIMD (4 waves); %5.1f B/ns per CU pair\n", name, ns_wave, ns_wave / 4,
           8 * 4 * bytes / ns_wave);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0, 0>("v_add_f32_e32 (4 B), waves in step", 4);
    run<0, 1>("v_add_f32_e32 (4 B), waves staggered", 4);
    run<1, 0>("v_fma_f32 (8 B), waves in step", 8);
    run<1, 1>("v_fma_f32 (8 B), waves staggered", 8);
    return 0;
}
