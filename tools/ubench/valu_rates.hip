// micro-benchmark: issue cost (cycles per wave-instruction) of the VALU ops the sweep
// kernel uses, measured with s_memtime, W waves per SIMD (one block per CU, 256*W threads)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 96
#define NACC 24
#define ITER 200
template <int OP> __device__ __forceinline__ void body(double (&a)[NACC], float (&f)[NACC], double c, float cf) {
#pragma unroll
    for (int r = 0; r < REP / NACC; r++) {
#pragma unroll
        for (int j = 0; j < NACC; j++) {
            if (OP == 0) a[j] = a[j] + c;                       // v_add_f64
            if (OP == 1) a[j] = fma(a[j], c, c);                // v_fma_f64
            if (OP == 2) a[j] = fmin(fabs(a[j]), c);            // v_min_f64
            if (OP == 3) f[j] = f[j] + cf;                      // v_add_f32
            if (OP == 4) f[j] = fmaf(f[j], cf, cf);             // v_fma_f32
            if (OP == 5) f[j] = (float)a[j] + f[j];             // v_cvt_f32_f64 + add_f32
            if (OP == 6) a[j] = __builtin_amdgcn_rcp(a[j]);     // v_rcp_f64
            if (OP == 7) a[j] = __builtin_rint(a[j] * c);          // v_rndne_f64
            if (OP == 8) { int lo = __builtin_amdgcn_readlane(__double2loint(a[j]), j); f[j] += __int_as_float(lo); } // readlane + cvt + add
            if (OP == 9) a[j] = a[j] * c;                       // v_mul_f64
        }
    }
}
template <int OP> __global__ void k(double *out, long long *cyc, double c, float cf) {
    double a[NACC]; float f[NACC];
    for (int j = 0; j < NACC; j++) { a[j] = threadIdx.x * 1e-3 + j; f[j] = (float)a[j]; }
    long long r0 = __builtin_amdgcn_s_memrealtime();
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < ITER; it++) body<OP>(a, f, c, cf);
    long long t1 = __builtin_amdgcn_s_memtime();
    long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0; for (int j = 0; j < NACC; j++) s += a[j] + f[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[2*blockIdx.x] = t1 - t0; cyc[2*blockIdx.x+1] = r1 - r0; }
}
template <int OP> void run(const char *name, int wavesPerSimd, int blocks = 256) {
    int threads = 256 * wavesPerSimd;
    double *out; long long *cyc;
    hipMalloc(&out, sizeof(double) * blocks * threads); hipMalloc(&cyc, sizeof(long long) * blocks * 2);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0000001, 1.0000001f);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0000001, 1.0000001f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0000001, 1.0000001f); hipEventRecord(e1);
    hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks*2); hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 2, hipMemcpyDeviceToHost);
    double avg = 0, avr = 0; for (int i = 0; i < blocks; i++) { avg += h[2*i]; avr += h[2*i+1]; } avg /= blocks; avr /= blocks;
    // s_memtime ticks at 100 MHz? report ticks per (instruction * waves on the SIMD)
    printf("%-24s blocks=%4d waves/blk=%2d memtime=%.0f realtime(100MHz)=%.0f -> memtime clock %.1f MHz, kernel %.3f ms, ns per op per wave %.3f\n",
           name, blocks, wavesPerSimd*4, avg, avr, avg / avr * 100.0, ms, avr * 10.0 / ((double)ITER * REP));
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int b : {1, 256, 512, 1024}) for (int w : {1, 4}) { run<0>("v_add_f64", w, b); run<3>("v_pk_add_f32", w, b); }
    return 0;
}
