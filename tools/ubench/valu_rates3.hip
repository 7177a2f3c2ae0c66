// valu_rates3.hip -- issue cost of the exact instruction forms the fp32 screening uses, measured
// with inline asm (16 independent chains, 2 waves per SIMD, one 512-thread block per CU).
// hipcc -O2 --offload-arch=gfx950 valu_rates3.hip -o valu_rates3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 400
#define X16(m) m(0) m(1) m(2) m(3) m(4) m(5) m(6) m(7) m(8) m(9) m(10) m(11) m(12) m(13) m(14) m(15)

template <int OP> __global__ void __launch_bounds__(1024) k(float *out, float cf, long long *cyc)
{
    float f[16], g[16]; unsigned u[16];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[16];
    for (int j = 0; j < 16; j++) { f[j] = threadIdx.x * 1e-3f + j; g[j] = f[j] * 0.5f; u[j] = j; p[j] = f2{f[j], g[j]}; }
    const long long t0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < ITER; it++) {
#define SUB(j) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(f[j]) : "s"(cf));
#define MUL(j) asm volatile("v_mul_f32_e32 %0, %1, %1" : "=v"(f[j]) : "v"(g[j]));
#define FMAC(j) asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(f[j]) : "v"(g[j]));
#define FMA3(j) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(f[j]) : "v"(g[j]));
#define CMPADDC(j) asm volatile("v_cmp_nle_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(u[j]) : "s"(cf), "v"(f[j]) : "vcc");
#define CMP(j) asm volatile("v_cmp_nle_f32_e32 vcc, %0, %1" : : "s"(cf), "v"(f[j]) : "vcc");
#define ADDC(j) asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(u[j]) : : "vcc");
#define RND(j) asm volatile("v_rndne_f32_e32 %0, %1" : "=v"(f[j]) : "v"(g[j]));
#define PKFMA(j) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[j]) : "v"(p[(j + 1) & 15]));
#define PKADD(j) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[j]) : "v"(p[(j + 1) & 15]));
#define PKMUL(j) asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(p[j]) : "v"(p[(j + 1) & 15]));
#define DOT2(j) asm volatile("v_dot2_f32_f16 %0, %1, %1, %0" : "+v"(f[j]) : "v"(g[j]));
#define DOT2C(j) asm volatile("v_dot2c_f32_f16_e32 %0, %1, %1" : "+v"(f[j]) : "v"(g[j]));
#define PKADDH(j) asm volatile("v_pk_add_f16 %0, %1, %0" : "+v"(f[j]) : "v"(g[j]));
#define MINABS(j) asm volatile("v_min_f32_e64 %0, |%1|, %0" : "+v"(f[j]) : "v"(g[j]));
#define PAIR(j) asm volatile("v_sub_f32_e32 %0, %3, %1\n\tv_mul_f32_e32 %0, %0, %0\n\tv_sub_f32_e32 %2, %3, %1\n\tv_fmac_f32_e32 %0, %2, %2\n\tv_sub_f32_e32 %2, %3, %1\n\tv_fmac_f32_e32 %0, %2, %2" : "=&v"(f[j]), "+v"(g[j]), "=&v"(p[j].x) : "s"(cf));
        if (OP == 0) { X16(SUB) } if (OP == 1) { X16(MUL) } if (OP == 2) { X16(FMAC) } if (OP == 3) { X16(FMA3) }
        if (OP == 4) { X16(CMPADDC) } if (OP == 5) { X16(CMP) } if (OP == 6) { X16(ADDC) } if (OP == 7) { X16(RND) }
        if (OP == 8) { X16(PKFMA) } if (OP == 9) { X16(PKADD) } if (OP == 10) { X16(PKMUL) } if (OP == 11) { X16(DOT2) }
#define SCREENP(j) asm volatile("v_pk_sub_i16 %0, %3, %1\n\tv_dot2_i32_i16 %0, %0, %0, 0 clamp\n\tv_cvt_f32_i32_e32 %0, %0\n\tv_mul_f32_e32 %0, %3, %0\n\tv_fma_mix_f32 %0, %2, %2, %0 op_sel_hi:[1,1,0]\n\tv_cmp_nle_f32_e32 vcc, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc" : "=&v"(f[j]), "+v"(u[j]) : "v"(g[j]), "s"(cf) : "vcc");
#define PKSUBI(j) asm volatile("v_pk_sub_i16 %0, %1, %0" : "+v"(u[j]) : "v"(u[(j + 1) & 15]));
#define DOT2I(j) asm volatile("v_dot2_i32_i16 %0, %1, %1, %0 clamp" : "+v"(u[j]) : "v"(u[(j + 1) & 15]));
#define CVTI(j) asm volatile("v_cvt_f32_i32_e32 %0, %1" : "=v"(f[j]) : "v"(u[j]));
#define FMAMIX(j) asm volatile("v_fma_mix_f32 %0, %1, %1, %0 op_sel_hi:[1,1,0]" : "+v"(f[j]) : "v"(g[j]));
#define SUBU(j) asm volatile("v_sub_u32_e32 %0, %1, %0" : "+v"(u[j]) : "v"(u[(j + 1) & 15]));
#define MUL24(j) asm volatile("v_mul_i32_i24_e32 %0, %1, %1" : "=v"(u[j]) : "v"(u[(j + 1) & 15]));
#define ASHR(j) asm volatile("v_ashrrev_i32_e32 %0, 16, %1" : "=v"(u[j]) : "v"(u[(j + 1) & 15]));
#define RFL(j) asm volatile("v_readfirstlane_b32 s20, %0\n\tv_add_f32_e32 %0, s20, %0" : "+v"(f[j]) : : "s20");
#define RFL2(j) asm volatile("v_readfirstlane_b32 s20, %0\n\tv_readfirstlane_b32 s21, %1\n\tv_add_f32_e32 %0, s20, %0\n\tv_add_f32_e32 %1, s21, %1" : "+v"(f[j]), "+v"(g[j]) : : "s20", "s21");
#define RDL(j) asm volatile("v_readlane_b32 s20, %0, 5\n\tv_add_f32_e32 %0, s20, %0" : "+v"(f[j]) : : "s20");
#define WRL(j) asm volatile("v_writelane_b32 %0, %1, 7" : "+v"(f[j]) : "s"(cf));
#define F64ADD(j) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[j]) : "v"(d[(j + 1) & 15]));
        if (OP == 17) { X16(SCREENP) }
        if (OP == 25) { X16(RFL) } if (OP == 26) { X16(RFL2) } if (OP == 27) { X16(RDL) } if (OP == 28) { X16(WRL) }
        if (OP == 18) { X16(PKSUBI) } if (OP == 19) { X16(DOT2I) } if (OP == 20) { X16(CVTI) } if (OP == 21) { X16(FMAMIX) }
        if (OP == 22) { X16(SUBU) } if (OP == 23) { X16(MUL24) } if (OP == 24) { X16(ASHR) }
        if (OP == 12) { X16(PKADDH) } if (OP == 13) { X16(MINABS) } if (OP == 14) { X16(PAIR) } if (OP == 15) { X16(DOT2C) }
    }
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    float s = 0; for (int j = 0; j < 16; j++) s += f[j] + g[j] + u[j] + p[j].x + p[j].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP> void run(const char *name, int per, int threads = 512)
{
    const int blocks = 256;
    const int wps = threads / 256; // waves per SIMD
    float *out; long long *cyc;
    hipMalloc(&out, sizeof(float) * blocks * 1024); hipMalloc(&cyc, sizeof(long long) * blocks);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 1.0000001f, cyc);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 1.0000001f, cyc);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double avr = 0; for (int i = 0; i < blocks; i++) avr += h[i]; avr /= blocks;
    // 2 waves per SIMD share it: SIMD time per wave-instruction = elapsed / (2 * ITER * 16 * per)
    const double ns = avr * 10.0 / ((double)wps * ITER * 16 * per);
    printf("%-44s %6.3f ns per wave-instr per SIMD  (%.2f cycles at 2.4 GHz, %.2f at 2.1)\n", name, ns, ns * 2.4, ns * 2.1);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0>("v_sub_f32_e32 v, s, v", 1); run<1>("v_mul_f32_e32 v, v, v", 1); run<2>("v_fmac_f32_e32", 1);
    run<3>("v_fma_f32 (VOP3)", 1); run<4>("v_cmp_nle_e32 + v_addc_co_e32", 2); run<5>("v_cmp_nle_f32_e32", 1);
    run<6>("v_addc_co_u32_e32", 1); run<7>("v_rndne_f32_e32", 1); run<8>("v_pk_fma_f32", 1); run<9>("v_pk_add_f32", 1);
    run<10>("v_pk_mul_f32", 1); run<11>("v_dot2_f32_f16", 1); run<15>("v_dot2c_f32_f16_e32", 1); run<12>("v_pk_add_f16", 1);
    run<13>("v_min_f32_e64 |v|", 1); run<14>("interior pair-eval (6 instr)", 6);
    printf("-- 2 waves per SIMD, the screen's instructions\n");
    run<18>("v_pk_sub_i16", 1); run<19>("v_dot2_i32_i16 clamp", 1); run<20>("v_cvt_f32_i32_e32", 1); run<21>("v_fma_mix_f32", 1);
    run<22>("v_sub_u32_e32", 1); run<23>("v_mul_i32_i24_e32", 1); run<24>("v_ashrrev_i32_e32", 1); run<0>("v_sub_f32_e32", 1);
    run<12>("v_pk_add_f16", 1); run<11>("v_dot2_f32_f16", 1);
    printf("-- lane operations (each followed by a VALU use of the scalar), 2 and 4 waves per SIMD\n");
    run<25>("v_readfirstlane + v_add using it", 2); run<26>("2 x v_readfirstlane + 2 x v_add", 4); run<27>("v_readlane + v_add using it", 2);
    run<28>("v_writelane", 1);
    run<25>("v_readfirstlane + v_add using it", 2, 1024); run<27>("v_readlane + v_add using it", 2, 1024); run<28>("v_writelane", 1, 1024);
    for (int t : {512}) {
        printf("-- %d wave(s) per SIMD\n", t / 256);
        run<17>("screen: 7 VALU (cmp_e32 + addc)", 7, t);
        run<0>("v_sub_f32_e32", 1, t); run<11>("v_dot2_f32_f16", 1, t); run<3>("v_fma_f32", 1, t);
    }
    return 0;
}
