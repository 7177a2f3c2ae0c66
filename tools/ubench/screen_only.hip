// screen_only.hip -- the screen of the screened sweep kernel (mx_screen_lds / mx_screen of
// csrc/smcx_sweep_mx.hip, the very code) alone in a loop: how long does one pass over S slots x
// 2 probes take when nothing else competes for the SIMD, at 1..8 waves per SIMD?
// (Earlier versions of this file also held ablations and rolled-loop forms of the screen; their
// results are in profiles/r01_screened_kernel.log.)
// hipcc -O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950 screen_only.hip -o screen_only
#include "../../montecarlo-surfacer_amd/csrc/smcx_sweep_mx.hip"
#include <cstdio>

using namespace smcx;

template <bool ZL, int S, int WV>
__global__ void __launch_bounds__(64, WV) screen_only(unsigned *out, int iters, unsigned axy0, unsigned bxy0,
                                                      float u2, float thr)
{
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
        unsigned ca[(S + 31) / 32] = {}, cb[(S + 31) / 32] = {};
        if constexpr (ZL) mx_screen_lds<S>(xy, zl[0], lane, axy, 0x50005000u, bxy, 0x51005100u, u2, thr, ca, cb);
        else mx_screen<S>(xy, z, axy, 1.5f, bxy, -2.5f, u2, thr, ca, cb);
        for (int w = 0; w < (S + 31) / 32; w++) acc ^= ca[w] ^ cb[w];
    }
    out[blockIdx.x * 64 + lane] = acc;
}

template <bool ZL, int S, int WV> void run(const char *name)
{
    const int blocks = 1024 * WV, iters = 4096;
    unsigned *out; hipMalloc(&out, blocks * 64 * sizeof(unsigned));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int r = 0; r < 2; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((screen_only<ZL, S, WV>), dim3(blocks), dim3(64), 0, 0, out, iters, 12345u, 777u, 2.5e-7f, 9.4f);
        hipEventRecord(e1); hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-30s S=%2d, %d waves/SIMD: %7.2f ms for %d passes -> %7.1f ns SIMD time per pass, %5.2f ns per slot and probe\n",
           name, S, WV, ms, iters, ms * 1e6 / iters / WV, ms * 1e6 / iters / WV / (2 * S));
    hipFree(out);
}

int main()
{
    run<true, 64, 4>("z as fp16 in LDS");
    run<false, 64, 2>("z as fp32 in registers");
    run<false, 32, 1>("z as fp32 in registers"); run<false, 32, 2>("z as fp32 in registers");
    run<false, 32, 4>("z as fp32 in registers"); run<false, 16, 8>("z as fp32 in registers");
    return 0;
}
