#!/usr/bin/env python3
"""gen_issue.py -- writes issue_costs.hip: issue cost of single VALU instruction forms on gfx950 in
SHADER CYCLES (s_memtime), not ns, at 1, 2, 4 and 8 waves per SIMD, with explicit register
numbers so that encoding (VOP1/VOP2/VOP3/VOP3P/VOPC), number of VGPR source operands, VGPR bank
(index % 4) of the sources, SGPR/constant operands and dependent chains can be varied one at a time.

Every kernel is ONE inline-asm block: 32 independent instructions per loop iteration (destinations
v16..v47, sources v48..v79 unless the variant says otherwise), s_memtime around the loop.
    python3 gen_issue.py > issue_costs.hip && hipcc -O2 --offload-arch=gfx950 issue_costs.hip -o issue_costs
"""
import sys

V = []  # (name, [32 instruction strings], instr per line)


def rep(name, fmt, n=32, per=1):
    """fmt may use {d} = v(16+j), {a},{b},{c} = sources chosen by the lambda"""
    V.append((name, [fmt(j) for j in range(n)], per))


def d(j): return "v%d" % (16 + j)
def s(j, k=0): return "v%d" % (48 + (j + 8 * k) % 32)


def bank(j, b, k=0):
    """a source register of bank b (index % 4 == b), distinct per (j, k) as far as possible"""
    base = 48 + 4 * ((j + 3 * k) % 8)
    return "v%d" % (base + b)


# ---- encodings of the same operation ----------------------------------------------------------
rep("v_add_f32_e32 d,a,b (VOP2)", lambda j: "v_add_f32_e32 %s, %s, %s" % (d(j), s(j), s(j, 1)))
rep("v_add_f32_e64 d,a,b (VOP3)", lambda j: "v_add_f32_e64 %s, %s, %s" % (d(j), s(j), s(j, 1)))
rep("v_add_f32_e32 d,s,b (SGPR src0)", lambda j: "v_add_f32_e32 %s, s22, %s" % (d(j), s(j, 1)))
rep("v_add_f32_e32 d,1.0,b (inline const)", lambda j: "v_add_f32_e32 %s, 1.0, %s" % (d(j), s(j, 1)))
rep("v_add_f32_e32 d,lit,b (32-bit literal)", lambda j: "v_add_f32_e32 %s, 0x40490fdb, %s" % (d(j), s(j, 1)))
rep("v_mov_b32_e32 d,a (VOP1)", lambda j: "v_mov_b32_e32 %s, %s" % (d(j), s(j)))
rep("v_mov_b32_e64 d,a (VOP3)", lambda j: "v_mov_b32_e64 %s, %s" % (d(j), s(j)))
rep("v_and_b32_e32 d,a,b", lambda j: "v_and_b32_e32 %s, %s, %s" % (d(j), s(j), s(j, 1)))
rep("v_sub_u32_e32 d,a,b", lambda j: "v_sub_u32_e32 %s, %s, %s" % (d(j), s(j), s(j, 1)))
rep("v_sub_u32_e32 d,s,b", lambda j: "v_sub_u32_e32 %s, s22, %s" % (d(j), s(j, 1)))
rep("v_sub_u16_e32 d,a,b", lambda j: "v_sub_u16_e32 %s, %s, %s" % (d(j), s(j), s(j, 1)))
rep("v_fmac_f32_e32 d,a,b", lambda j: "v_fmac_f32_e32 %s, %s, %s" % (d(j), s(j), s(j, 1)))
rep("v_mul_i32_i24_e32 d,a,b", lambda j: "v_mul_i32_i24_e32 %s, %s, %s" % (d(j), s(j), s(j, 1)))
rep("v_mul_i32_i24_e32 d,a,a", lambda j: "v_mul_i32_i24_e32 %s, %s, %s" % (d(j), s(j), s(j)))
rep("v_lshrrev_b32_e32 d,31,a", lambda j: "v_lshrrev_b32_e32 %s, 31, %s" % (d(j), s(j)))
rep("v_cvt_f32_i32_e32 d,a", lambda j: "v_cvt_f32_i32_e32 %s, %s" % (d(j), s(j)))
rep("v_cvt_f32_f16_e32 d,a", lambda j: "v_cvt_f32_f16_e32 %s, %s" % (d(j), s(j)))
rep("v_rndne_f32_e32 d,a", lambda j: "v_rndne_f32_e32 %s, %s" % (d(j), s(j)))
# ---- three-operand VOP3: how many VGPR sources, which banks -----------------------------------
rep("v_fma_f32 d,a,b,c (3 VGPR, mixed banks)", lambda j: "v_fma_f32 %s, %s, %s, %s" % (d(j), s(j), s(j, 1), s(j, 2)))
rep("v_fma_f32 d,a,b,c (3 VGPR, banks 0,1,2)", lambda j: "v_fma_f32 %s, %s, %s, %s" % (d(j), bank(j, 0), bank(j, 1, 1), bank(j, 2, 2)))
rep("v_fma_f32 d,a,b,c (3 VGPR, all bank 0)", lambda j: "v_fma_f32 %s, %s, %s, %s" % (d(j), bank(j, 0), bank(j, 0, 1), bank(j, 0, 2)))
rep("v_fma_f32 d,a,a,c (2 distinct VGPR)", lambda j: "v_fma_f32 %s, %s, %s, %s" % (d(j), s(j), s(j), s(j, 2)))
rep("v_fma_f32 d,a,a,a (1 distinct VGPR)", lambda j: "v_fma_f32 %s, %s, %s, %s" % (d(j), s(j), s(j), s(j)))
rep("v_fma_f32 d,s,b,c (SGPR + 2 VGPR)", lambda j: "v_fma_f32 %s, s22, %s, %s" % (d(j), s(j, 1), s(j, 2)))
rep("v_fma_f32 d,s,b,1.0 (SGPR + VGPR + const)", lambda j: "v_fma_f32 %s, s22, %s, 1.0" % (d(j), s(j, 1)))
rep("v_fma_f32 d,a,b,d (accumulate in place)", lambda j: "v_fma_f32 %s, %s, %s, %s" % (d(j), s(j), s(j, 1), d(j)))
rep("v_alignbit_b32 d,a,b,31", lambda j: "v_alignbit_b32 %s, %s, %s, 31" % (d(j), s(j), s(j, 1)))
rep("v_alignbit_b32 d,d,b,31 (in place)", lambda j: "v_alignbit_b32 %s, %s, %s, 31" % (d(j), d(j), s(j, 1)))
rep("v_lshl_or_b32 d,a,1,b", lambda j: "v_lshl_or_b32 %s, %s, 1, %s" % (d(j), s(j), s(j, 1)))
rep("v_lshl_add_u32 d,a,1,b", lambda j: "v_lshl_add_u32 %s, %s, 1, %s" % (d(j), s(j), s(j, 1)))
rep("v_add3_u32 d,a,b,c", lambda j: "v_add3_u32 %s, %s, %s, %s" % (d(j), s(j), s(j, 1), s(j, 2)))
rep("v_and_or_b32 d,a,b,c", lambda j: "v_and_or_b32 %s, %s, %s, %s" % (d(j), s(j), s(j, 1), s(j, 2)))
rep("v_bfe_i32 d,a,0,16", lambda j: "v_bfe_i32 %s, %s, 0, 16" % (d(j), s(j)))
rep("v_mad_i32_i24 d,a,a,c", lambda j: "v_mad_i32_i24 %s, %s, %s, %s" % (d(j), s(j), s(j), s(j, 2)))
rep("v_mad_u32_u24 d,a,b,c", lambda j: "v_mad_u32_u24 %s, %s, %s, %s" % (d(j), s(j), s(j, 1), s(j, 2)))
rep("v_mad_i32_i16 d,a,a,c", lambda j: "v_mad_i32_i16 %s, %s, %s, %s" % (d(j), s(j), s(j), s(j, 2)))
rep("v_mad_i32_i16 d,a,a,c op_sel hi", lambda j: "v_mad_i32_i16 %s, %s, %s, %s op_sel:[1,1,0,0]" % (d(j), s(j), s(j), s(j, 2)))
rep("v_mad_i32_i16 d,a,a,d clamp (in place)", lambda j: "v_mad_i32_i16 %s, %s, %s, %s clamp" % (d(j), s(j), s(j), d(j)))
rep("v_mad_u64_u32 (64-bit)", lambda j: "v_mad_u64_u32 v[%d:%d], s[30:31], %s, %s, v[%d:%d]" % (16 + 2 * (j % 16), 17 + 2 * (j % 16), s(j), s(j, 1), 48 + 2 * (j % 16), 49 + 2 * (j % 16)))
# ---- packed (VOP3P) ----------------------------------------------------------------------------
rep("v_pk_sub_i16 d,a,b", lambda j: "v_pk_sub_i16 %s, %s, %s" % (d(j), s(j), s(j, 1)))
rep("v_pk_sub_i16 d,s,b", lambda j: "v_pk_sub_i16 %s, s22, %s" % (d(j), s(j, 1)))
rep("v_pk_sub_u16 d,s,b", lambda j: "v_pk_sub_u16 %s, s22, %s" % (d(j), s(j, 1)))
rep("v_pk_add_f16 d,s,b", lambda j: "v_pk_add_f16 %s, s22, %s" % (d(j), s(j, 1)))
rep("v_pk_mul_lo_u16 d,a,a", lambda j: "v_pk_mul_lo_u16 %s, %s, %s" % (d(j), s(j), s(j)))
rep("v_pk_mad_i16 d,a,a,c", lambda j: "v_pk_mad_i16 %s, %s, %s, %s" % (d(j), s(j), s(j), s(j, 2)))
rep("v_pk_fma_f16 d,a,a,c", lambda j: "v_pk_fma_f16 %s, %s, %s, %s" % (d(j), s(j), s(j), s(j, 2)))
rep("v_pk_fma_f32 d,a,b,c", lambda j: "v_pk_fma_f32 v[%d:%d], v[%d:%d], v[%d:%d], v[%d:%d]" % (
    16 + 2 * (j % 16), 17 + 2 * (j % 16), 48 + 2 * (j % 16), 49 + 2 * (j % 16), 48 + 2 * ((j + 5) % 16), 49 + 2 * ((j + 5) % 16),
    48 + 2 * ((j + 9) % 16), 49 + 2 * ((j + 9) % 16)))
rep("v_dot2_i32_i16 d,a,a,0", lambda j: "v_dot2_i32_i16 %s, %s, %s, 0" % (d(j), s(j), s(j)))
rep("v_dot2_i32_i16 d,a,a,0 clamp", lambda j: "v_dot2_i32_i16 %s, %s, %s, 0 clamp" % (d(j), s(j), s(j)))
rep("v_dot2_i32_i16 d,a,a,s", lambda j: "v_dot2_i32_i16 %s, %s, %s, s22" % (d(j), s(j), s(j)))
rep("v_dot2_i32_i16 d,a,a,c", lambda j: "v_dot2_i32_i16 %s, %s, %s, %s" % (d(j), s(j), s(j), s(j, 2)))
rep("v_dot2_i32_i16 d,a,b,c", lambda j: "v_dot2_i32_i16 %s, %s, %s, %s" % (d(j), s(j), s(j, 1), s(j, 2)))
rep("v_dot2c_i32_i16_e32 d,a,a (VOP2)", lambda j: "v_dot2c_i32_i16_e32 %s, %s, %s" % (d(j), s(j), s(j)))
rep("v_dot2_f32_f16 d,a,a,c", lambda j: "v_dot2_f32_f16 %s, %s, %s, %s" % (d(j), s(j), s(j), s(j, 2)))
rep("v_dot2c_f32_f16_e32 d,a,a (VOP2)", lambda j: "v_dot2c_f32_f16_e32 %s, %s, %s" % (d(j), s(j), s(j)))
rep("v_dot4_i32_i8 d,a,a,c", lambda j: "v_dot4_i32_i8 %s, %s, %s, %s" % (d(j), s(j), s(j), s(j, 2)))
rep("v_dot4c_i32_i8_e32 d,a,a (VOP2)", lambda j: "v_dot4c_i32_i8_e32 %s, %s, %s" % (d(j), s(j), s(j)))
rep("v_fma_mix_f32 d,a,a,c hi", lambda j: "v_fma_mix_f32 %s, %s, %s, %s op_sel:[1,1,0] op_sel_hi:[1,1,0]" % (d(j), s(j), s(j), s(j, 2)))
# ---- compares ------------------------------------------------------------------------------------
rep("v_cmp_lt_i32_e32 vcc,a,b", lambda j: "v_cmp_lt_i32_e32 vcc, %s, %s" % (s(j), s(j, 1)))
rep("v_cmp_lt_i32_e64 s[24:25],a,b", lambda j: "v_cmp_lt_i32_e64 s[24:25], %s, %s" % (s(j), s(j, 1)))
rep("v_cmp_lt_f32_e32 vcc,a,b", lambda j: "v_cmp_lt_f32_e32 vcc, %s, %s" % (s(j), s(j, 1)))
rep("v_cmp + v_addc_co (pair)", lambda j: "v_cmp_lt_i32_e32 vcc, %s, %s\n\tv_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (s(j), s(j, 1), d(j), d(j), d(j)), per=2)
rep("v_cndmask_b32_e32 d,a,b,vcc", lambda j: "v_cndmask_b32_e32 %s, %s, %s, vcc" % (d(j), s(j), s(j, 1)))
rep("v_cndmask_b32_e64 d,a,b,s[28:29]", lambda j: "v_cndmask_b32_e64 %s, %s, %s, s[28:29]" % (d(j), s(j), s(j, 1)))
rep("v_cndmask_b32_e64 d,0,1,s[28:29]", lambda j: "v_cndmask_b32_e64 %s, 0, 1, s[28:29]" % d(j))
rep("v_cndmask_b32_e64 d,0,b,s[28:29] (mask = lanes 0-31)", lambda j: "v_cndmask_b32_e64 %s, 0, %s, s[40:41]" % (d(j), s(j, 1)))
rep("v_cmp_lt_i32 vcc + v_cndmask vcc (pair)", lambda j: "v_cmp_lt_i32_e32 vcc, %s, %s\n\tv_cndmask_b32_e32 %s, %s, %s, vcc" % (s(j), s(j, 1), d(j), s(j), s(j, 1)), per=2)
rep("v_bfi_b32 d,m,a,b", lambda j: "v_bfi_b32 %s, %s, %s, %s" % (d(j), s(j), s(j, 1), s(j, 2)))
rep("v_perm_b32 d,a,b,sel", lambda j: "v_perm_b32 %s, %s, %s, %s" % (d(j), s(j), s(j, 1), s(j, 2)))
rep("v_max_i32_e32 d,a,b", lambda j: "v_max_i32_e32 %s, %s, %s" % (d(j), s(j), s(j, 1)))
rep("v_min_f32_e32 d,a,b", lambda j: "v_min_f32_e32 %s, %s, %s" % (d(j), s(j), s(j, 1)))
rep("v_mul_f32_e32 d,a,b", lambda j: "v_mul_f32_e32 %s, %s, %s" % (d(j), s(j), s(j, 1)))
rep("v_xor_b32_e32 d,a,b", lambda j: "v_xor_b32_e32 %s, %s, %s" % (d(j), s(j), s(j, 1)))
rep("v_or_b32_e32 d,a,b", lambda j: "v_or_b32_e32 %s, %s, %s" % (d(j), s(j), s(j, 1)))
rep("v_add_u32_e32 d,a,b", lambda j: "v_add_u32_e32 %s, %s, %s" % (d(j), s(j), s(j, 1)))
rep("v_ashrrev_i32_e32 d,31,a", lambda j: "v_ashrrev_i32_e32 %s, 31, %s" % (d(j), s(j)))
rep("v_lshlrev_b32_e32 d,3,a", lambda j: "v_lshlrev_b32_e32 %s, 3, %s" % (d(j), s(j)))
rep("v_cvt_i32_f32_e32 d,a", lambda j: "v_cvt_i32_f32_e32 %s, %s" % (d(j), s(j)))
rep("v_med3_i32 d,a,b,c", lambda j: "v_med3_i32 %s, %s, %s, %s" % (d(j), s(j), s(j, 1), s(j, 2)))
rep("v_bfrev_b32_e32 d,a", lambda j: "v_bfrev_b32_e32 %s, %s" % (d(j), s(j)))
rep("v_ffbl_b32_e32 d,a", lambda j: "v_ffbl_b32_e32 %s, %s" % (d(j), s(j)))
rep("v_mov_b64_e32 d,a", lambda j: "v_mov_b64_e32 v[%d:%d], v[%d:%d]" % (16 + 2 * (j % 16), 17 + 2 * (j % 16), 48 + 2 * (j % 16), 49 + 2 * (j % 16)))
rep("v_mov_b64_e32 d,s", lambda j: "v_mov_b64_e32 v[%d:%d], s[22:23]" % (16 + 2 * (j % 16), 17 + 2 * (j % 16)))
rep("v_mov_b32_e32 d,s", lambda j: "v_mov_b32_e32 %s, s22" % d(j))
rep("v_mov_b32_e32 d,1.0", lambda j: "v_mov_b32_e32 %s, 1.0" % d(j))
rep("v_writelane_b32 d,s,7", lambda j: "v_writelane_b32 %s, s22, 7" % d(j))
rep("v_cvt_i32_f64_e32 d,a", lambda j: "v_cvt_i32_f64_e32 %s, v[%d:%d]" % (d(j), 48 + 2 * (j % 16), 49 + 2 * (j % 16)))
rep("v_cvt_f64_i32_e32 d,a", lambda j: "v_cvt_f64_i32_e32 v[%d:%d], %s" % (16 + 2 * (j % 16), 17 + 2 * (j % 16), s(j)))
rep("v_cvt_f32_f64_e32 d,a", lambda j: "v_cvt_f32_f64_e32 %s, v[%d:%d]" % (d(j), 48 + 2 * (j % 16), 49 + 2 * (j % 16)))
rep("select by exec: s_mov exec lo; v_mov; s_mov exec hi; v_mov; s_mov exec all (5 instr)", lambda j: "s_mov_b64 exec, s[40:41]\n\tv_mov_b32_e32 %s, %s\n\ts_mov_b64 exec, s[42:43]\n\tv_mov_b32_e32 %s, %s\n\ts_mov_b64 exec, -1" % (d(j), s(j), d(j), s(j, 1)), n=16, per=5)
rep("s_and_saveexec + v_mov + s_or exec (3 instr)", lambda j: "s_and_saveexec_b64 s[44:45], s[40:41]\n\tv_mov_b32_e32 %s, %s\n\ts_or_b64 exec, exec, s[44:45]" % (d(j), s(j)), n=16, per=3)
# ---- fp64 ---------------------------------------------------------------------------------------
def dd(j): return "v[%d:%d]" % (16 + 2 * (j % 16), 17 + 2 * (j % 16))
def sd(j, k=0): return "v[%d:%d]" % (48 + 2 * ((j + 5 * k) % 16), 49 + 2 * ((j + 5 * k) % 16))
rep("v_add_f64 d,a,b", lambda j: "v_add_f64 %s, %s, %s" % (dd(j), sd(j), sd(j, 1)))
rep("v_mul_f64 d,a,b", lambda j: "v_mul_f64 %s, %s, %s" % (dd(j), sd(j), sd(j, 1)))
rep("v_fma_f64 d,a,b,c", lambda j: "v_fma_f64 %s, %s, %s, %s" % (dd(j), sd(j), sd(j, 1), sd(j, 2)))
rep("v_fma_f64 d,s,b,c", lambda j: "v_fma_f64 %s, s[22:23], %s, %s" % (dd(j), sd(j, 1), sd(j, 2)))
rep("v_min_f64 d,|a|,b", lambda j: "v_min_f64 %s, |%s|, %s" % (dd(j), sd(j), sd(j, 1)))
rep("v_cmp_lt_f64_e32 vcc,a,b", lambda j: "v_cmp_lt_f64_e32 vcc, %s, %s" % (sd(j), sd(j, 1)))
rep("v_rcp_f64_e32 d,a", lambda j: "v_rcp_f64_e32 %s, %s" % (dd(j), sd(j)))
rep("v_rndne_f64_e32 d,a", lambda j: "v_rndne_f64_e32 %s, %s" % (dd(j), sd(j)))
# ---- lane / scalar traffic ----------------------------------------------------------------------
rep("v_readfirstlane_b32 s,a", lambda j: "v_readfirstlane_b32 s%d, %s" % (36 + j % 12, s(j)))
rep("v_readlane_b32 s,a,5", lambda j: "v_readlane_b32 s%d, %s, 5" % (36 + j % 12, s(j)))
rep("v_mov_b32_dpp quad_perm", lambda j: "v_mov_b32_dpp %s, %s quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" % (d(j), s(j)))
rep("v_add_f32_dpp row_ror:8", lambda j: "v_add_f32_dpp %s, %s, %s row_ror:8 row_mask:0xf bank_mask:0xf" % (d(j), s(j), s(j, 1)))
rep("v_permlane32_swap d,a", lambda j: "v_permlane32_swap_b32_e32 %s, %s" % (d(j), s(j)))
rep("s_nop 0", lambda j: "s_nop 0")
rep("s_add_u32 (SALU)", lambda j: "s_add_u32 s%d, s%d, s22" % (36 + j % 12, 36 + j % 12))
# ---- dependent chains ---------------------------------------------------------------------------
rep("v_add_f32_e32 chain of 32 (dependent)", lambda j: "v_add_f32_e32 v16, v16, %s" % s(j))
rep("v_fma_f32 chain of 32 (dependent)", lambda j: "v_fma_f32 v16, v16, %s, %s" % (s(j), s(j, 1)))
rep("v_add_f64 chain of 32 (dependent)", lambda j: "v_add_f64 v[16:17], v[16:17], %s" % sd(j))
rep("v_add_f32 4 chains (dep. distance 4)", lambda j: "v_add_f32_e32 v%d, v%d, %s" % (16 + j % 4, 16 + j % 4, s(j)))
rep("v_add_f32 8 chains (dep. distance 8)", lambda j: "v_add_f32_e32 v%d, v%d, %s" % (16 + j % 8, 16 + j % 8, s(j)))

# ---- whole screens: one slot and both probes per group of instructions --------------------------
# current product (z fp16 in LDS form, without the ds_read): per 2 slots x 2 probes = 26 VALU
def screen_now(j):
    x0, x1, zz = s(2 * j), s(2 * j + 1), s(j, 3)
    t = ["v%d" % (16 + (8 * j + k) % 32) for k in range(8)]
    return "\n\t".join([
        "v_pk_add_f16 %s, s22, %s neg_lo:[0,1] neg_hi:[0,1]" % (t[0], zz),
        "v_pk_add_f16 %s, s23, %s neg_lo:[0,1] neg_hi:[0,1]" % (t[1], zz),
        "v_pk_sub_i16 %s, s24, %s" % (t[2], x0), "v_dot2_i32_i16 %s, %s, %s, 0 clamp" % (t[2], t[2], t[2]),
        "v_cvt_f32_i32_e32 %s, %s" % (t[2], t[2]), "v_fma_f32 %s, s26, %s, %s" % (t[2], t[2], "v80"),
        "v_fma_mix_f32 %s, %s, %s, %s op_sel_hi:[1,1,0]" % (t[2], t[0], t[0], t[2]),
        "v_alignbit_b32 v81, v81, %s, 31" % t[2],
        "v_pk_sub_i16 %s, s25, %s" % (t[3], x0), "v_dot2_i32_i16 %s, %s, %s, 0 clamp" % (t[3], t[3], t[3]),
        "v_cvt_f32_i32_e32 %s, %s" % (t[3], t[3]), "v_fma_f32 %s, s26, %s, %s" % (t[3], t[3], "v80"),
        "v_fma_mix_f32 %s, %s, %s, %s op_sel_hi:[1,1,0]" % (t[3], t[1], t[1], t[3]),
        "v_alignbit_b32 v82, v82, %s, 31" % t[3],
        "v_pk_sub_i16 %s, s24, %s" % (t[4], x1), "v_dot2_i32_i16 %s, %s, %s, 0 clamp" % (t[4], t[4], t[4]),
        "v_cvt_f32_i32_e32 %s, %s" % (t[4], t[4]), "v_fma_f32 %s, s26, %s, %s" % (t[4], t[4], "v80"),
        "v_fma_mix_f32 %s, %s, %s, %s op_sel:[1,1,0] op_sel_hi:[1,1,0]" % (t[4], t[0], t[0], t[4]),
        "v_alignbit_b32 v81, v81, %s, 31" % t[4],
        "v_pk_sub_i16 %s, s25, %s" % (t[5], x1), "v_dot2_i32_i16 %s, %s, %s, 0 clamp" % (t[5], t[5], t[5]),
        "v_cvt_f32_i32_e32 %s, %s" % (t[5], t[5]), "v_fma_f32 %s, s26, %s, %s" % (t[5], t[5], "v80"),
        "v_fma_mix_f32 %s, %s, %s, %s op_sel:[1,1,0] op_sel_hi:[1,1,0]" % (t[5], t[1], t[1], t[5]),
        "v_alignbit_b32 v82, v82, %s, 31" % t[5]])


rep("SCREEN now: int16 xy, fp16 z (26 VALU / 2 slots x 2 probes)", screen_now, n=4, per=26)


# all-integer screen: z as int16 pairs, dz^2 added by v_mad_i32_i16 into the dot2 (thr in its accumulator)
def screen_int(j):
    x0, x1, zz = s(2 * j), s(2 * j + 1), s(j, 3)
    t = ["v%d" % (16 + (8 * j + k) % 32) for k in range(8)]
    out = ["v_pk_sub_i16 %s, s22, %s" % (t[0], zz), "v_pk_sub_i16 %s, s23, %s" % (t[1], zz)]
    for (xx, hi) in ((x0, 0), (x1, 1)):
        for (pr, tz, acc, tt) in (("s24", t[0], "v81", t[2 + 2 * hi]), ("s25", t[1], "v82", t[3 + 2 * hi])):
            out += ["v_pk_sub_i16 %s, %s, %s" % (tt, pr, xx),
                    "v_dot2_i32_i16 %s, %s, %s, s26" % (tt, tt, tt),
                    "v_mad_i32_i16 %s, %s, %s, %s%s" % (tt, tz, tz, tt, " op_sel:[1,1,0,0]" if hi else ""),
                    "v_alignbit_b32 %s, %s, %s, 31" % (acc, acc, tt)]
    return "\n\t".join(out)


rep("SCREEN int: pk_sub, dot2(acc=-thr), mad_i32_i16, alignbit (18 VALU / 2x2)", screen_int, n=4, per=18)


# the same with VOP2 forms where they exist: v_sub_u32 for the packed difference (borrow = 1 unit),
# v_dot2c (accumulator preloaded with dz^2 - thr by one v_mad per slot and probe)
def screen_vop2(j):
    x0, x1, zz = s(2 * j), s(2 * j + 1), s(j, 3)
    t = ["v%d" % (16 + (8 * j + k) % 32) for k in range(8)]
    out = ["v_pk_sub_i16 %s, s22, %s" % (t[0], zz), "v_pk_sub_i16 %s, s23, %s" % (t[1], zz)]
    for (xx, hi) in ((x0, 0), (x1, 1)):
        for (pr, tz, acc, tt, td) in (("s24", t[0], "v81", t[2 + 2 * hi], t[6]), ("s25", t[1], "v82", t[3 + 2 * hi], t[7])):
            out += ["v_mad_i32_i16 %s, %s, %s, s26%s" % (tt, tz, tz, " op_sel:[1,1,0,0]" if hi else ""),
                    "v_sub_u32_e32 %s, %s, %s" % (td, pr, xx),
                    "v_dot2c_i32_i16_e32 %s, %s, %s" % (tt, td, td),
                    "v_alignbit_b32 %s, %s, %s, 31" % (acc, acc, tt)]
    return "\n\t".join(out)


rep("SCREEN vop2: mad_i32_i16, sub_u32, dot2c, alignbit (18 VALU / 2x2)", screen_vop2, n=4, per=18)


# flags through the carry: v_cmp into VCC + v_addc (both 32-bit encodings) instead of sub-thr + alignbit
def screen_cmp(j):
    x0, x1, zz = s(2 * j), s(2 * j + 1), s(j, 3)
    t = ["v%d" % (16 + (8 * j + k) % 32) for k in range(8)]
    out = ["v_pk_sub_i16 %s, s22, %s" % (t[0], zz), "v_pk_sub_i16 %s, s23, %s" % (t[1], zz)]
    for (xx, hi) in ((x0, 0), (x1, 1)):
        for (pr, tz, acc, tt, td) in (("s24", t[0], "v81", t[2 + 2 * hi], t[6]), ("s25", t[1], "v82", t[3 + 2 * hi], t[7])):
            out += ["v_mul_i32_i24_e32 %s, %s, %s" % (tt, tz, tz),
                    "v_sub_u32_e32 %s, %s, %s" % (td, pr, xx),
                    "v_dot2c_i32_i16_e32 %s, %s, %s" % (tt, td, td),
                    "v_cmp_gt_u32_e32 vcc, s26, %s" % tt,
                    "v_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (acc, acc, acc)]
    return "\n\t".join(out)


rep("SCREEN cmp: mul24, sub_u32, dot2c, cmp, addc (22 VALU / 2x2, all 32-bit encodings)", screen_cmp, n=4, per=22)

# ------------------------------------------------------------------------------------------------
print("// generated by gen_issue.py -- do not edit")
print("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdlib>\n#include <vector>\n#include <algorithm>\n#include <cstring>")
clob = ", ".join('"v%d"' % i for i in range(16, 84)) + ", " + ", ".join('"s%d"' % i for i in list(range(20, 32)) + list(range(36, 52))) + ', "vcc", "scc", "memory"'
print("""
#define INIT \\
    "v_cvt_f32_u32_e32 v16, v0\\n\\t" "v_mul_f32_e32 v16, 0x3a83126f, v16\\n\\t" \\
    "s_mov_b32 s22, 0x3f800123\\n\\t" "s_mov_b32 s23, 0x3f810000\\n\\t" "s_mov_b32 s24, 0x12345678\\n\\t" \\
    "s_mov_b32 s25, 0x23456789\\n\\t" "s_mov_b32 s26, 0xffff0000\\n\\t" "s_mov_b32 s30, 3\\n\\t" "s_mov_b32 s31, 0\\n\\t" \\
    "s_mov_b64 vcc, 0\\n\\t" "s_mov_b64 s[28:29], 0x0f0f0f0f\\n\\t" "s_mov_b32 s40, -1\\n\\t" "s_mov_b32 s41, 0\\n\\t" \\
    "s_mov_b32 s42, 0\\n\\t" "s_mov_b32 s43, -1\\n\\t"
""")
# registers v17..v83 initialised from v16 in the asm prologue
init = "".join('"v_add_f32_e32 v%d, 0x3f8ccccd, v%d\\n\\t" ' % (i, i - 1) for i in range(17, 84))
for idx, (name, lines, per) in enumerate(V):
    body = "".join('"%s\\n\\t"\n        ' % ln.replace("\n\t", '\\n\\t" "') for ln in lines)
    print("__global__ void __launch_bounds__(1024) k%d(unsigned long long *out, int iters)\n{" % idx)
    print("    unsigned long long t0, t1, r0, r1;")
    print("    asm volatile(INIT %s" % init)
    print('        "s_mov_b32 s20, %4\\n\\t" "s_memtime %0\\n\\t" "s_memrealtime %2\\n\\t" "s_waitcnt lgkmcnt(0)\\n\\t"')
    print('        "1:\\n\\t"')
    print("        " + body)
    print('        "s_sub_u32 s20, s20, 1\\n\\t" "s_cmp_lg_u32 s20, 0\\n\\t" "s_cbranch_scc1 1b\\n\\t"')
    print('        "s_memtime %1\\n\\t" "s_memrealtime %3\\n\\t" "s_waitcnt lgkmcnt(0)\\n\\t"')
    print('        : "=&s"(t0), "=&s"(t1), "=&s"(r0), "=&s"(r1) : "s"(iters) : %s);' % clob)
    print("    if ((threadIdx.x & 63) == 0) { unsigned long long *o = out + 2 * ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));")
    print("        o[0] = t1 - t0; o[1] = r1 - r0; }\n}")
print("struct Var { const char *name; void (*fn)(unsigned long long *, int); int n; };")
print("static Var vars[] = {")
for idx, (name, lines, per) in enumerate(V):
    print('    {"%s", k%d, %d},' % (name, idx, len(lines) * per))
print("};")
print(r"""
int main(int argc, char **argv)
{
    const int iters = 1500;
    const char *only = argc > 1 ? argv[1] : nullptr;
    unsigned long long *out;
    hipMalloc(&out, sizeof(unsigned long long) * 2 * 512 * 16);
    std::vector<unsigned long long> h(2 * 512 * 16);
    printf("# SIMD cycles per wave-instruction = median wave cycles / (waves per SIMD x instructions); clock = cycles / s_memrealtime (100 MHz)\n");
    printf("%-78s %8s %8s %8s %8s   %s\n", "instruction form", "1 w/SIMD", "2", "3", "4", "GHz at 4");
    for (auto &v : vars) {
        if (only && !strstr(v.name, only)) continue;
        double res[4], ghz = 0;
        int wi = 0;
        for (int w : {1, 2, 3, 4}) {
            const int threads = 256 * w, blocks = 256;
            const int waves = blocks * threads / 64;
            v.fn<<<blocks, threads>>>(out, 50);
            v.fn<<<blocks, threads>>>(out, iters);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), out, sizeof(unsigned long long) * 2 * waves, hipMemcpyDeviceToHost);
            std::vector<double> c(waves), r(waves);
            for (int i = 0; i < waves; i++) { c[i] = (double)h[2 * i]; r[i] = (double)h[2 * i + 1]; }
            std::nth_element(c.begin(), c.begin() + waves / 2, c.end());
            std::nth_element(r.begin(), r.begin() + waves / 2, r.end());
            res[wi++] = c[waves / 2] / ((double)w * iters * v.n);
            if (w == 4) ghz = c[waves / 2] / (r[waves / 2] * 10.0);
        }
        printf("%-78s %8.2f %8.2f %8.2f %8.2f   %.2f\n", v.name, res[0], res[1], res[2], res[3], ghz);
        fflush(stdout);
    }
    return 0;
}""")
