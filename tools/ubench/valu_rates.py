#!/usr/bin/env python3
"""valu_rates.py OUTDIR -- what ONE instruction of each form costs a gfx950 SIMD, by itself: for each form a kernel whose loop
body is 128 INDEPENDENT copies of it (32 rotating destination registers, fixed sources), run with 1, 2 and 4 wavefronts per SIMD
on all 256 CUs, timed with s_memtime inside the wavefront.  Printed: SIMD cycles per wave-instruction (cycles a wavefront took / (its instructions x wavefronts
per SIMD)); 'a|b' forms alternate a and b.  Round 4 priced the kinds inside the sweep kernel with 40 padded instructions each (v_mov 1.8,
v_alignbit 3.2, v_add_f64 3.4, s_add 2.0); the phase table of round 5 split the kernel's 32-bit VALU by opcode on that basis.
This table says which forms are which (generates, compiles with hipcc and runs: for a GPU box, through gpurun)."""
import os
import subprocess
import sys

out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/valu_rates"
os.makedirs(out, exist_ok=True)

# (name, instruction with {d} = rotating destination VGPR index, {d2} = rotating even pair)
FORMS = [
    ("v_mov_b32 v,v", "v_mov_b32 v{d}, v2"),
    ("v_mov_b32 v,s", "v_mov_b32 v{d}, s10"),
    ("v_add_u32 v,v,v", "v_add_u32 v{d}, v2, v3"),
    ("v_add_u32 v,s,v", "v_add_u32 v{d}, s10, v3"),
    ("v_sub_u32 v,v,v", "v_sub_u32 v{d}, v2, v3"),
    ("v_and_b32 v,v,v", "v_and_b32 v{d}, v2, v3"),
    ("v_lshlrev_b32 v,imm,v", "v_lshlrev_b32 v{d}, 3, v2"),
    ("v_lshlrev_b32 v,s,v", "v_lshlrev_b32 v{d}, s10, v2"),
    ("v_max_i32 v,v,v", "v_max_i32 v{d}, v2, v3"),
    ("v_min_u32 v,v,v", "v_min_u32 v{d}, v2, v3"),
    ("v_or_b32 v,v,v", "v_or_b32 v{d}, v2, v3"),
    ("v_xor_b32 v,v,v", "v_xor_b32 v{d}, v2, v3"),
    ("v_lshlrev_b32 v,v,v", "v_lshlrev_b32 v{d}, v2, v3"),
    ("v_lshrrev_b32 v,v,v", "v_lshrrev_b32 v{d}, v2, v3"),
    ("v_ashrrev_i32 v,v,v", "v_ashrrev_i32 v{d}, v2, v3"),
    ("v_subrev_u32 v,v,v", "v_subrev_u32 v{d}, v2, v3"),
    ("v_mov_b32 v,0 (inline constant)", "v_mov_b32 v{d}, 0"),
    ("v_mov_b32 v,literal", "v_mov_b32 v{d}, 0x12345"),
    ("v_add_u32 v,1,v (inline constant)", "v_add_u32 v{d}, 1, v3"),
    ("v_and_b32 v,literal,v", "v_and_b32 v{d}, 0xff00ff, v3"),
    ("v_add_co_u32 v,vcc,v,v", "v_add_co_u32 v{d}, vcc, v2, v3"),
    ("v_addc_co_u32 v,vcc,v,v,vcc", "v_addc_co_u32 v{d}, vcc, v2, v3, vcc"),
    ("v_mul_f32 v,v,v", "v_mul_f32 v{d}, v2, v3"),
    ("v_fmac_f32 v,v,v", "v_fmac_f32 v{d}, v2, v3"),
    ("v_cvt_f32_i32 v,v", "v_cvt_f32_i32 v{d}, v2"),
    ("v_not_b32 v,v", "v_not_b32 v{d}, v2"),
    ("v_bfrev_b32 v,v", "v_bfrev_b32 v{d}, v2"),
    ("v_mul_u32_u24 v,v,v", "v_mul_u32_u24 v{d}, v2, v3"),
    ("v_mul_i32_i24 v,v,v", "v_mul_i32_i24 v{d}, v2, v3"),
    ("v_add_u32 v,v,v alternating with v_dot4", "v_add_u32 v{d}, v2, v3|v_dot4_i32_i8 v{d}, v2, v2, v3"),
    ("v_sub,v_dot4,v_alignbit (one slot of the screen)", "v_sub_u32 v{d}, v2, v3|v_dot4_i32_i8 v{d}, v4, v4, v5|v_alignbit_b32 v{d}, v6, v7, 31"),
    ("v_add_u32 alternating with s_add_u32", "v_add_u32 v{d}, v2, v3|s_add_u32 s20, s10, s11"),
    ("v_dot4 alternating with s_add_u32", "v_dot4_i32_i8 v{d}, v2, v2, v3|s_add_u32 s20, s10, s11"),
    ("v_add_f64 alternating with s_add_u32", "v_add_f64 v[{d2}:{d2p}], v[2:3], v[4:5]|s_add_u32 s20, s10, s11"),
    ("v_add_f64 alternating with v_add_u32", "v_add_f64 v[{d2}:{d2p}], v[2:3], v[4:5]|v_add_u32 v9, v8, v8"),
    ("v_dot4_i32_i8 v,v,v,s", "v_dot4_i32_i8 v{d}, v2, v2, s10"),
    ("v_dot4_i32_i8 v,v,v,v", "v_dot4_i32_i8 v{d}, v2, v2, v3"),
    ("v_dot2_i32_i16 v,v,v,v", "v_dot2_i32_i16 v{d}, v2, v2, v3"),
    ("v_alignbit_b32 v,v,v,31", "v_alignbit_b32 v{d}, v2, v3, 31"),
    ("v_alignbit_b32 d,d,v,31 (own dst)", "v_alignbit_b32 v{d}, v{d}, v3, 31"),
    ("v_and_or_b32", "v_and_or_b32 v{d}, v2, v3, v4"),
    ("v_lshl_add_u32", "v_lshl_add_u32 v{d}, v2, 3, v4"),
    ("v_lshl_or_b32", "v_lshl_or_b32 v{d}, v2, 1, v4"),
    ("v_add3_u32", "v_add3_u32 v{d}, v2, v3, v4"),
    ("v_bfe_u32", "v_bfe_u32 v{d}, v2, 3, 5"),
    ("v_perm_b32", "v_perm_b32 v{d}, v2, v3, v4"),
    ("v_ffbl_b32", "v_ffbl_b32 v{d}, v2"),
    ("v_mul_lo_u32", "v_mul_lo_u32 v{d}, v2, v3"),
    ("v_mad_u32_u24", "v_mad_u32_u24 v{d}, v2, v3, v4"),
    ("v_cmp_lt_i32 vcc (VOP2 encoding)", "v_cmp_lt_i32 vcc, v2, v3"),
    ("v_cmp_lt_i32 s[20:21] (VOP3)", "v_cmp_lt_i32 s[20:21], v2, v3"),
    ("v_cmp_lt_i32 vcc, s, v", "v_cmp_lt_i32 vcc, s10, v3"),
    ("v_cndmask_b32 vcc", "v_cndmask_b32 v{d}, v2, v3, vcc"),
    ("v_cndmask_b32_e64 vcc (VOP3 encoding)", "v_cndmask_b32_e64 v{d}, v2, v3, vcc"),
    ("v_cndmask_b32 vcc alternating with v_mov", "v_cndmask_b32 v{d}, v2, v3, vcc|v_mov_b32 v{d}, v2"),
    ("v_cndmask_b32 s[22:23]", "v_cndmask_b32 v{d}, v2, v3, s[22:23]"),
    ("v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 v{d}, s10, v3"),
    ("v_readlane_b32", "v_readlane_b32 s20, v2, 5"),
    ("v_readfirstlane_b32", "v_readfirstlane_b32 s20, v2"),
    ("v_writelane_b32", "v_writelane_b32 v{d}, s10, 5"),
    ("v_mov_b32 dpp row_shr:1", "v_mov_b32_dpp v{d}, v2 row_shr:1 row_mask:0xf bank_mask:0xf"),
    ("v_add_u32 dpp row_shr:1", "v_add_u32_dpp v{d}, v2, v3 row_shr:1 row_mask:0xf bank_mask:0xf"),
    ("v_mov_b64", "v_mov_b64 v[{d2}:{d2p}], v[2:3]"),
    ("v_lshlrev_b64", "v_lshlrev_b64 v[{d2}:{d2p}], 3, v[2:3]"),
    ("v_add_f64", "v_add_f64 v[{d2}:{d2p}], v[2:3], v[4:5]"),
    ("v_mul_f64", "v_mul_f64 v[{d2}:{d2p}], v[2:3], v[4:5]"),
    ("v_fma_f64", "v_fma_f64 v[{d2}:{d2p}], v[2:3], v[4:5], v[6:7]"),
    ("v_fma_f64 v,v,s,v", "v_fma_f64 v[{d2}:{d2p}], v[2:3], s[12:13], v[6:7]"),
    ("v_rcp_f64", "v_rcp_f64 v[{d2}:{d2p}], v[2:3]"),
    ("v_cvt_f64_i32", "v_cvt_f64_i32 v[{d2}:{d2p}], v2"),
    ("v_rndne_f64", "v_rndne_f64 v[{d2}:{d2p}], v[2:3]"),
    ("v_fma_f32", "v_fma_f32 v{d}, v2, v3, v4"),
    ("v_pk_fma_f32", "v_pk_fma_f32 v[{d2}:{d2p}], v[2:3], v[4:5], v[6:7]"),
    ("v_add_f32", "v_add_f32 v{d}, v2, v3"),
    ("s_add_u32", "s_add_u32 s20, s10, s11"),
    ("s_and_b64", "s_and_b64 s[20:21], s[12:13], s[14:15]"),
    ("s_cmp_lt_u32", "s_cmp_lt_u32 s10, s11"),
    ("s_nop 0", "s_nop 0"),
    ("ds_bpermute_b32", "ds_bpermute_b32 v{d}, v8, v2"),
    ("ds_read_b32", "ds_read_b32 v{d}, v8"),
    ("ds_read_b64", "ds_read_b64 v[{d2}:{d2p}], v8"),
]

N_BODY = 128
src = ["#include <hip/hip_runtime.h>", "#include <cstdio>", "#include <cstdint>", "#include <vector>",
       "#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, \"%s: %s\\n\", #x, hipGetErrorString(e_)); exit(1); } } while (0)"]
clob = ", ".join('"v%d"' % i for i in range(2, 44)) + ', "s10", "s11", "s12", "s13", "s14", "s15", "s20", "s21", "s22", "s23", "s24", "vcc", "scc", "memory"'
for k, (name, ins) in enumerate(FORMS):
    body = []
    for i in range(N_BODY):
        d = 10 + (i % 32)
        d2 = 10 + 2 * (i % 16)
        alts = ins.split("|")
        body.append(alts[i % len(alts)].format(d=d, d2=d2, d2p=d2 + 1))
        if ins.startswith("ds_") and i % 8 == 7:
            body.append("s_waitcnt lgkmcnt(0)")
    init = ["v_mov_b32 v%d, %d" % (i, i) for i in range(2, 10)] + ["v_lshlrev_b32 v8, 2, v8", "s_mov_b32 s10, 3", "s_mov_b32 s11, 5",
            "s_mov_b64 s[12:13], 0x3ff0", "s_mov_b64 s[14:15], -1", "s_mov_b64 s[22:23], 0x55", "s_mov_b64 vcc, 0x33", "v_cvt_f64_i32 v[2:3], v2", "v_cvt_f64_i32 v[4:5], v4",
            "v_cvt_f64_i32 v[6:7], v6"]
    if "f64" not in ins and "_b64 v" not in ins and "pk_" not in ins:
        init = init[:-3]
    text = "\\n\\t".join(init + ["s_mov_b32 s24, %0", "s_waitcnt lgkmcnt(0)", "1:"] + body +
                         ["s_sub_u32 s24, s24, 1", "s_cmp_lg_u32 s24, 0", "s_cbranch_scc1 1b", "s_waitcnt vmcnt(0) lgkmcnt(0)"])
    src.append("__global__ void __launch_bounds__(256) k%d(int loops, unsigned long long *t)\n{\n    __shared__ unsigned lds[1024];\n"
               "    lds[threadIdx.x] = threadIdx.x;\n    __syncthreads();\n"
               "    const unsigned long long t0 = __builtin_readcyclecounter();\n"
               "    asm volatile(\"%s\" :: \"s\"(loops) : %s);\n"
               "    const unsigned long long t1 = __builtin_readcyclecounter();\n"
               "    if ((threadIdx.x & 63) == 0) t[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;\n    if (loops < 0) t[0] = lds[5];\n}\n" % (k, text, clob))
src.append("typedef void (*kern_t)(int, unsigned long long *);")
src.append("static const struct { const char *name; kern_t k; } K[] = {")
for k, (name, ins) in enumerate(FORMS):
    src.append('    {"%s", k%d},' % (name, k))
src.append("};")
src.append(r"""
int main()
{
    unsigned long long *t;
    const int maxw = 256 * 4 * 4;
    CHECK(hipMalloc(&t, maxw * sizeof(*t)));
    std::vector<unsigned long long> h(maxw);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int loops = 2000, nbody = %d;
    printf("%%-36s %%10s %%10s %%10s   (s_memtime ticks of a SIMD per wave-instruction at 1, 2, 4 wavefronts per SIMD = ticks a wavefront took / (instructions x wavefronts per SIMD))\n", "form", "1/SIMD", "2/SIMD", "4/SIMD");
    for (unsigned k = 0; k < sizeof(K) / sizeof(K[0]); k++) {
        printf("%%-36s", K[k].name);
        double ns4 = 0;
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int grid = 256 * wps; // workgroups of 4 wavefronts: one per SIMD of a CU
            K[k].k<<<grid, 256>>>(10, t); // warm
            CHECK(hipEventRecord(e0));
            K[k].k<<<grid, 256>>>(loops, t);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(h.data(), t, grid * 4 * sizeof(*t), hipMemcpyDeviceToHost));
            double s = 0; for (int i = 0; i < grid * 4; i++) s += (double)h[i];
            const double ticks = s / (grid * 4);
            printf(" %%10.2f", ticks / ((double)loops * nbody * wps));
            if (wps == 4) ns4 = ms * 1e6 / ((double)loops * nbody * wps);
        }
        printf("   %%.3f ns per wave-instruction per SIMD at 4 (event clock, incl. launch)\n", ns4);
        fflush(stdout);
    }
    return 0;
}
""" % N_BODY)
path = os.path.join(out, "valu_rates.hip")
open(path, "w").write("\n".join(src))
exe = os.path.join(out, "valu_rates")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", path, "-o", exe])
if "--build-only" not in sys.argv:
    subprocess.check_call([exe])
