// screen_only.hip -- the screen of the screened sweep kernel (mx_screen_lds / mx_screen of
// csrc/smcx_sweep_mx.hip, the very code) alone in a loop, four resp. two waves per SIMD: how long
// does one pass over 64 slots x 2 probes take when nothing else competes for the SIMD?
// hipcc -O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950 screen_only.hip -o screen_only
#include "../../montecarlo-surfacer_amd/csrc/smcx_sweep_mx.hip"
#include <cstdio>

using namespace smcx;

// rolled form: a loop over the 16 groups of four slots with run-time register indexing
template <int S>
__device__ __forceinline__ void screen_rolled(const unsigned (&xy)[S], const unsigned (&zw)[S / 2][64], int lane,
                                              unsigned axy, unsigned azz, unsigned bxy, unsigned bzz, float u2,
                                              float thr, unsigned (&ca)[2], unsigned (&cb)[2])
{
#pragma unroll 1
    for (int k0 = 0; k0 < S; k0 += 4) {
        float qa[4], qb[4];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const mx_h2 zz = __builtin_bit_cast(mx_h2, zw[k0 / 2 + p][lane]);
            const mx_h2 da = __builtin_bit_cast(mx_h2, azz) - zz;
            const mx_h2 db = __builtin_bit_cast(mx_h2, bzz) - zz;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int j = 2 * p + h;
                const float fa = (float)(h ? da.y : da.x), fb = (float)(h ? db.y : db.x);
                const unsigned p_xy = xy[k0 + j];
                qa[j] = __builtin_fmaf(fa, fa, mx_qxy(axy, p_xy, u2));
                qb[j] = __builtin_fmaf(fb, fb, mx_qxy(bxy, p_xy, u2));
            }
        }
        const int w = k0 >> 5;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            unsigned ta = w ? ca[1] : ca[0], tb = w ? cb[1] : cb[0];
            asm("v_cmp_nle_f32_e32 vcc, %2, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(ta) : "v"(qa[j]), "s"(thr) : "vcc");
            asm("v_cmp_nle_f32_e32 vcc, %2, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(tb) : "v"(qb[j]), "s"(thr) : "vcc");
            if (w) { ca[1] = ta; cb[1] = tb; } else { ca[0] = ta; cb[0] = tb; }
        }
    }
}

// ablations of the unrolled screen (z fp32 in registers): which instruction costs the time?
//  MODE 0: full   1: no cmp/addc (sum q)   2: cmp/addc only on a precomputed q   3: no dot2/cvt (xy as float sub/mul)
template <int MODE, int S = 64, int WV = 2>
__global__ void __launch_bounds__(64, WV) screen_ablate(unsigned *out, int iters, unsigned axy0, unsigned bxy0, float u2, float thr)
{
    const int lane = threadIdx.x;
    unsigned xy[S]; float z[S];
#pragma unroll
    for (int k = 0; k < S; k++) { xy[k] = (unsigned)(lane * 2654435761u + k * 40503u); z[k] = (float)((lane * 7 + k * 13) % 97) - 48.f; }
    unsigned acc = 0; float facc = 0.f;
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        const unsigned axy = (unsigned)__builtin_amdgcn_readfirstlane((int)(axy0 + it * 977u));
        const unsigned bxy = (unsigned)__builtin_amdgcn_readfirstlane((int)(bxy0 + it * 131u));
        const float az = 1.5f + it, bz = -2.5f - it;
        unsigned ca[2] = {0u, 0u}, cb[2] = {0u, 0u};
#pragma unroll
        for (int k = 0; k < S; k++) {
            float qa, qb;
            if (MODE == 2) { qa = z[k] + az; qb = z[k] + bz; }
            else if (MODE == 3) {
                const float xa = __builtin_bit_cast(float, xy[k]) - __builtin_bit_cast(float, axy), xb = __builtin_bit_cast(float, xy[k]) - __builtin_bit_cast(float, bxy);
                const float da = az - z[k], db = bz - z[k];
                qa = __builtin_fmaf(da, da, u2 * (xa * xa)); qb = __builtin_fmaf(db, db, u2 * (xb * xb));
            } else {
                const float da = az - z[k], db = bz - z[k];
                qa = __builtin_fmaf(da, da, mx_qxy(axy, xy[k], u2)); qb = __builtin_fmaf(db, db, mx_qxy(bxy, xy[k], u2));
            }
            if (MODE == 1) { facc += qa; facc += qb; }
            else {
                asm("v_cmp_nle_f32_e32 vcc, %2, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(ca[k >> 5]) : "v"(qa), "s"(thr) : "vcc");
                asm("v_cmp_nle_f32_e32 vcc, %2, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(cb[k >> 5]) : "v"(qb), "s"(thr) : "vcc");
            }
        }
        acc ^= ca[0] ^ ca[1] ^ cb[0] ^ cb[1];
    }
    out[blockIdx.x * 64 + lane] = acc + (unsigned)facc;
}
template <int MODE, int S = 64, int WV = 2> void run_ablate(const char *name)
{
    const int blocks = 1024 * WV, iters = 4096;
    unsigned *out; hipMalloc(&out, blocks * 64 * sizeof(unsigned));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 2; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((screen_ablate<MODE, S, WV>), dim3(blocks), dim3(64), 0, 0, out, iters, 12345u, 777u, 2.5e-7f, 9.4f);
        hipEventRecord(e1); hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s S=%d, %d waves/SIMD: %7.2f ms -> %6.1f ns SIMD time per pass, %5.2f ns per slot and probe\n", name, S, WV, ms, ms * 1e6 / iters / WV, ms * 1e6 / iters / WV / (2 * S));
}

// rolled form on two 32-wide register vectors: dynamic extractelement -> indexed VGPR reads
typedef unsigned u32x32 __attribute__((ext_vector_type(32)));
__device__ __forceinline__ void screen_rolled_vec(const u32x32 &xv, int base, const unsigned (&zw)[32][64], int lane,
                                                  unsigned axy, unsigned azz, unsigned bxy, unsigned bzz, float u2,
                                                  float thr, unsigned &ca, unsigned &cb)
{
#pragma unroll 1
    for (int k0 = 0; k0 < 32; k0 += 4) {
        float qa[4], qb[4];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const mx_h2 zz = __builtin_bit_cast(mx_h2, zw[(base + k0) / 2 + p][lane]);
            const mx_h2 da = __builtin_bit_cast(mx_h2, azz) - zz;
            const mx_h2 db = __builtin_bit_cast(mx_h2, bzz) - zz;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int j = 2 * p + h;
                const float fa = (float)(h ? da.y : da.x), fb = (float)(h ? db.y : db.x);
                const unsigned p_xy = xv[k0 + j];
                qa[j] = __builtin_fmaf(fa, fa, mx_qxy(axy, p_xy, u2));
                qb[j] = __builtin_fmaf(fb, fb, mx_qxy(bxy, p_xy, u2));
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            asm("v_cmp_nle_f32_e32 vcc, %2, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(ca) : "v"(qa[j]), "s"(thr) : "vcc");
            asm("v_cmp_nle_f32_e32 vcc, %2, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(cb) : "v"(qb[j]), "s"(thr) : "vcc");
        }
    }
}

__global__ void __launch_bounds__(64, 4) screen_vec(unsigned *out, int iters, unsigned axy0, unsigned bxy0, float u2, float thr)
{
    __shared__ unsigned zl[32][64];
    const int lane = threadIdx.x;
    u32x32 xa, xb;
#pragma unroll
    for (int k = 0; k < 32; k++) { xa[k] = (unsigned)(lane * 2654435761u + k * 40503u); xb[k] = (unsigned)(lane * 40503u + k * 2654435761u); }
    for (int j = 0; j < 32; j++) zl[j][lane] = 0x50005000u + (unsigned)((lane + j) & 0xff);
    unsigned acc = 0;
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        const unsigned axy = (unsigned)__builtin_amdgcn_readfirstlane((int)(axy0 + it * 977u));
        const unsigned bxy = (unsigned)__builtin_amdgcn_readfirstlane((int)(bxy0 + it * 131u));
        unsigned ca[2] = {0u, 0u}, cb[2] = {0u, 0u};
        screen_rolled_vec(xa, 0, zl, lane, axy, 0x50005000u, bxy, 0x51005100u, u2, thr, ca[0], cb[0]);
        screen_rolled_vec(xb, 32, zl, lane, axy, 0x50005000u, bxy, 0x51005100u, u2, thr, ca[1], cb[1]);
        acc ^= ca[0] ^ ca[1] ^ cb[0] ^ cb[1];
    }
    out[blockIdx.x * 64 + lane] = acc;
}

template <bool ZL, bool ROLLED = false>
__global__ void __launch_bounds__(64, ZL ? 4 : 2) screen_only(unsigned *out, int iters, unsigned axy0, unsigned bxy0,
                                                              float u2, float thr)
{
    constexpr int S = 64;
    __shared__ unsigned zl[1][S / 2][64];
    const int lane = threadIdx.x;
    unsigned xy[S];
    float z[ZL ? 1 : S];
#pragma unroll
    for (int k = 0; k < S; k++) {
        xy[k] = (unsigned)(lane * 2654435761u + k * 40503u);
        if constexpr (!ZL) z[k] = (float)((lane * 7 + k * 13) % 97) - 48.f;
    }
    for (int j = 0; j < S / 2; j++) zl[0][j][lane] = 0x50005000u + (unsigned)((lane + j) & 0xff);
    unsigned acc = 0;
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        const unsigned axy = (unsigned)__builtin_amdgcn_readfirstlane((int)(axy0 + it * 977u));
        const unsigned bxy = (unsigned)__builtin_amdgcn_readfirstlane((int)(bxy0 + it * 131u));
        unsigned ca[2] = {0u, 0u}, cb[2] = {0u, 0u};
        if constexpr (ROLLED) screen_rolled<S>(xy, zl[0], lane, axy, 0x50005000u, bxy, 0x51005100u, u2, thr, ca, cb);
        else if constexpr (ZL) mx_screen_lds<S>(xy, zl[0], lane, axy, 0x50005000u, bxy, 0x51005100u, u2, thr, ca, cb);
        else mx_screen<S>(xy, z, axy, 1.5f, bxy, -2.5f, u2, thr, ca, cb);
        acc ^= ca[0] ^ ca[1] ^ cb[0] ^ cb[1];
    }
    out[blockIdx.x * 64 + lane] = acc;
}

template <bool ZL, bool ROLLED = false> void run(const char *name)
{
    const int blocks = 1024 * (ZL ? 4 : 2), iters = 4096;
    unsigned *out; hipMalloc(&out, blocks * 64 * sizeof(unsigned));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 2; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((screen_only<ZL, ROLLED>), dim3(blocks), dim3(64), 0, 0, out, iters, 12345u, 777u, 2.5e-7f, 9.4f);
        hipEventRecord(e1); hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = ZL ? 4 : 2;
    printf("%-34s %d waves/SIMD: %7.2f ms for %d passes -> %6.1f ns SIMD time per pass, %5.2f ns per slot and probe\n", name,
           (int)waves, ms, iters, ms * 1e6 / iters / waves, ms * 1e6 / iters / waves / 128);
    hipFree(out);
}

void run_vec()
{
    const int blocks = 4096, iters = 4096;
    unsigned *out; hipMalloc(&out, blocks * 64 * sizeof(unsigned));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 2; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(screen_vec, dim3(blocks), dim3(64), 0, 0, out, iters, 12345u, 777u, 2.5e-7f, 9.4f);
        hipEventRecord(e1); hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s 4 waves/SIMD: %7.2f ms for %d passes -> %6.1f ns SIMD time per pass, %5.2f ns per slot and probe\n",
           "screen, LDS z, rolled on vectors", ms, iters, ms * 1e6 / iters / 4, ms * 1e6 / iters / 4 / 128);
}

int main()
{
    run_ablate<0, 32, 1>("ablate: full"); run_ablate<0, 32, 2>("ablate: full"); run_ablate<0, 32, 4>("ablate: full"); run_ablate<0, 16, 8>("ablate: full");
    run_ablate<1, 32, 4>("ablate: no cmp/addc"); run_ablate<2, 32, 4>("ablate: cmp/addc only"); run_ablate<3, 32, 4>("ablate: float xy");
    run_ablate<0>("ablate: full"); run_ablate<1>("ablate: no cmp/addc"); run_ablate<2>("ablate: cmp/addc only"); run_ablate<3>("ablate: float xy (no pk_sub/dot2/cvt)");
    run_vec();
    run<true>("screen, z as fp16 in LDS");
    run<false>("screen, z as fp32 in registers");
    run<true, true>("screen, LDS z, rolled loop");
    return 0;
}
