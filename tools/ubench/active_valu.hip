// active_valu.hip -- what does SQ_ACTIVE_INST_VALU count?  ("0.61 / 0.73 / 0.99")
// Four kernels, each 4096 wavefronts (4 per SIMD on MI355X) running ITER x 32 independent VALU instructions of ONE form:
//   k_fast  v_add_f32 d, a, b        (1.9 SIMD cycles per wave-instruction at 4 waves/SIMD, profiles/r02_issue_costs.txt)
//   k_slow  v_alignbit_b32 d,a,b,31  (3.4)
//   k_f64   v_fma_f64                (3.4)
//   k_mix   the sweep kernel's rough mix: 2 fast : 3 slow : 2 fp64 per 7
// Every wavefront stores its s_memtime / s_memrealtime deltas; the host prints cycles per wave-instruction and the
// clock.  Run under `rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES` and divide
// SQ_ACTIVE_INST_VALU by SQ_INSTS_VALU: if the fast form reads ~0.5 quad-cycles per instruction the counter
// measures pipe occupancy; if it reads ~1.0 for every form it counts whole 4-cycle issue slots.
//   hipcc -O2 --offload-arch=gfx950 active_valu.hip -o active_valu
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define ITER 20000
#define R4(x) x x x x
#define R32(x) R4(R4(x)) R4(R4(x))

#define KERNEL(name, body)                                                                             \
    __global__ void __launch_bounds__(64, 4) name(unsigned long long *out)                              \
    {                                                                                                  \
        unsigned long long t0, t1, r0, r1;                                                             \
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)); \
        asm volatile("s_mov_b32 s40, 0\n\t"                                                           \
                     "1:\n\t" body                                                                      \
                     "s_add_u32 s40, s40, 1\n\t"                                                       \
                     "s_cmp_lt_u32 s40, %0\n\t"                                                        \
                     "s_cbranch_scc1 1b\n\t"                                                           \
                     :: "n"(ITER)                                                                      \
                     : "s40", "scc", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28",   \
                       "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43",   \
                       "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58",   \
                       "v59", "v60", "v61", "v62", "v63", "v127"); /* v127: 128 VGPRs = four waves per SIMD */                                            \
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)); \
        if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }    \
    }

#define FAST8 "v_add_f32 v16, v48, v49\n\tv_add_f32 v17, v50, v51\n\tv_add_f32 v18, v52, v53\n\tv_add_f32 v19, v54, v55\n\t" \
              "v_add_f32 v20, v56, v57\n\tv_add_f32 v21, v58, v59\n\tv_add_f32 v22, v60, v61\n\tv_add_f32 v23, v62, v63\n\t"
#define SLOW8 "v_alignbit_b32 v24, v48, v49, 31\n\tv_alignbit_b32 v25, v50, v51, 31\n\tv_alignbit_b32 v26, v52, v53, 31\n\tv_alignbit_b32 v27, v54, v55, 31\n\t" \
              "v_alignbit_b32 v28, v56, v57, 31\n\tv_alignbit_b32 v29, v58, v59, 31\n\tv_alignbit_b32 v30, v60, v61, 31\n\tv_alignbit_b32 v31, v62, v63, 31\n\t"
#define F648 "v_fma_f64 v[32:33], v[48:49], v[50:51], v[52:53]\n\tv_fma_f64 v[34:35], v[50:51], v[52:53], v[54:55]\n\t"  \
             "v_fma_f64 v[36:37], v[52:53], v[54:55], v[56:57]\n\tv_fma_f64 v[38:39], v[54:55], v[56:57], v[58:59]\n\t"  \
             "v_fma_f64 v[40:41], v[56:57], v[58:59], v[60:61]\n\tv_fma_f64 v[42:43], v[58:59], v[60:61], v[62:63]\n\t"  \
             "v_fma_f64 v[44:45], v[60:61], v[62:63], v[48:49]\n\tv_fma_f64 v[46:47], v[62:63], v[48:49], v[50:51]\n\t"

KERNEL(k_fast, FAST8 FAST8 FAST8 FAST8)
KERNEL(k_slow, SLOW8 SLOW8 SLOW8 SLOW8)
KERNEL(k_f64, F648 F648 F648 F648)
KERNEL(k_mix, FAST8 SLOW8 F648 SLOW8)   // 8 fast : 16 slow : 8 fp64

int main()
{
    const int nw = 4096;
    unsigned long long *d;
    hipMalloc(&d, 2 * nw * sizeof(unsigned long long));
    std::vector<unsigned long long> h(2 * nw);
    void (*ks[4])(unsigned long long *) = {k_fast, k_slow, k_f64, k_mix};
    const char *names[4] = {"k_fast (v_add_f32)", "k_slow (v_alignbit_b32)", "k_f64 (v_fma_f64)", "k_mix (8 fast, 16 slow, 8 fp64)"};
    for (int k = 0; k < 4; k++) {
        for (int rep = 0; rep < 2; rep++) { // the second launch is the one reported
            hipLaunchKernelGGL(ks[k], dim3(nw), dim3(64), 0, 0, d);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::vector<double> cyc, ghz;
        for (int w = 0; w < nw; w++) { cyc.push_back((double)h[2 * w]); ghz.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 0.1); }
        std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
        const double instr = 32.0 * ITER;
        printf("%-34s median wave %.0f cycles, %.2f GHz: %.2f SIMD cycles per wave-instruction at 4 waves/SIMD\n", names[k],
               cyc[nw / 2], ghz[nw / 2], cyc[nw / 2] / (4.0 * instr));
    }
    hipFree(d);
    return 0;
}
