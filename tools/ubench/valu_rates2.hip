// issue cost of single VALU ops at full chip load (1024 blocks x 1024 threads), via inline asm
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 400
#define NACC 16
template <int OP> __global__ void k(float *out, float cf, double cd) {
    float f[NACC]; double d[NACC]; float2 p[NACC];
    for (int j = 0; j < NACC; j++) { f[j] = threadIdx.x * 1e-3f + j; d[j] = f[j]; p[j] = make_float2(f[j], f[j] + 1); }
#pragma unroll 1
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
        for (int j = 0; j < NACC; j++) {
            if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[j]) : "v"(cf));
            if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[j]) : "v"(cd));
            if (OP == 2) asm volatile("v_min_f32 %0, |%0|, %1" : "+v"(f[j]) : "v"(cf));
            if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j]) : "v"(p[(j+1)%NACC]));
            if (OP == 4) asm volatile("v_mov_b32 %0, %1" : "+v"(f[j]) : "v"(cf));
            if (OP == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[j]) : "v"(cf));
            if (OP == 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[j]) : "v"(cf));
            if (OP == 7) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(f[j]) : "v"(d[j]));
            if (OP == 8) asm volatile("v_and_b32 %0, %0, %1" : "+v"(f[j]) : "v"(cf));
            if (OP == 9) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(f[j]), "v"(cf) : "vcc");
            if (OP == 10) asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(d[j]), "v"(cd) : "vcc");
        }
    }
    float s = 0; for (int j = 0; j < NACC; j++) s += f[j] + (float)d[j] + p[j].x + p[j].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char *name) {
    int blocks = 1024, threads = 1024;
    float *out; hipMalloc(&out, sizeof(float) * blocks * threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 1.0000001f, 1.0000001);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 1.0000001f, 1.0000001);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = (double)blocks * 16 * ITER * 4 * NACC / 1024.0;
    printf("%-16s %.3f ms  -> %.3f ns per wave-instr per SIMD (%.2f cycles at 2.1 GHz)\n", name, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.1);
    hipFree(out);
}
int main() {
    run<0>("v_add_f32"); run<1>("v_add_f64"); run<2>("v_min_f32 |x|"); run<3>("v_pk_add_f32"); run<4>("v_mov_b32");
    run<5>("v_cndmask_b32"); run<6>("v_fma_f32"); run<7>("v_cvt_f32_f64"); run<8>("v_and_b32"); run<9>("v_cmp_lt_f32"); run<10>("v_cmp_lt_f64");
    return 0;
}
