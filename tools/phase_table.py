#!/usr/bin/env python3
"""phase_table.py BODY.inc -- static instruction count per phase of the STEADY copy of the move of a merged-pass kernel
(sweep_kernel_mc64: csrc/build/smcx_sweep_mc_body64.inc), by instruction kind.  VERDICT r4 next #4: where the non-vector
instructions of a move sit.  Static = the main line of the steady copy as written (cold pieces -- near-wall pass, further
rounds, second hand-over, unsafe probes -- are not on it); the two screens are unrolled over all 16 groups and entered by a
computed jump, so their group blocks are listed per block and weighted with the executed groups per pass in the last column."""
import re
import sys

path = sys.argv[1]
groups_per_pass = float(sys.argv[2]) if len(sys.argv) > 2 else 3.63      # executed, bench line `executed` (lattice start)
accept = float(sys.argv[3]) if len(sys.argv) > 3 else 0.46
lines = [re.sub(r'\\n\\t"$', "", l.strip()[1:]) for l in open(path) if l.startswith('"')]
pre = re.match(r"(L\d+z8\w*?)_S_move:", next(l for l in lines if "_S_move:" in l)).group(1)
lab = lambda n: "%s_S_%s:" % (pre, n)
idx = {l: i for i, l in enumerate(lines)}
at = lambda n: idx[lab(n)]
end = next(i for i in range(at("move"), len(lines)) if lines[i].startswith("s_branch %s_G_move" % pre) or lines[i] == "%s_move:" % pre)
rej_branch = next(i for i in range(at("mgR0"), at("reject")) if lines[i].startswith("s_cbranch_scc0 %s_S_reject" % pre))
rowtest = next(i for i in range(at("nsr"), end) if lines[i].startswith("s_cmp_eq_u32") and "63" in lines[i])


def kind(l):
    if l.endswith(":"):
        return None
    op = l.split()[0]
    if op in ("s_waitcnt", "s_nop"):
        return "wait/nop"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc")):
        return "branch"
    if op.startswith("s_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu_f64" if "_f64" in op else "valu"
    return "other"


KINDS = ["valu", "valu_f64", "salu", "branch", "lds", "vmem", "smem", "wait/nop"]
# issue cost per wave-instruction at four wavefronts per SIMD, measured INSIDE this kernel (profiles/r04_instruction_costs_in_kernel.txt:
# scalar 2.0, fast 32-bit VALU 1.8, slow 32-bit VALU 3.2, fp64 3.4, s_nop 0.44) and, for v_rcp_f64, in isolation (r02_issue_costs.txt: 8).
# Fast 32-bit forms (r02_issue_costs.txt): add/sub/mul/fma(c)_f32, mov, and/or/xor, add/sub_u32, lshr/ashr -- with VGPR or inline
# sources only; any SGPR or literal source, and every other opcode (dot4, alignbit, mad, bfe, perm, cvt, cmp, cndmask with an
# SGPR mask, DPP, lane reads, ffbl, mbcnt, med3, min/max ...) is a slow form.
FAST = ("v_add_f32", "v_sub_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32")
COST = {"fast": 1.8, "slow": 3.2, "valu_f64": 3.4, "rcp_f64": 8.0, "salu": 2.0, "branch": 2.0, "lds": 2.0, "vmem": 2.0, "smem": 2.0, "nop": 0.44}


def cost(l):
    k = kind(l)
    if k is None:
        return 0.0, None
    op = l.split()[0]
    if k == "valu":
        ops = l[len(op):]
        sgpr = re.search(r"(?<![\w\[])s\d+|s\[\d+:\d+\]|vcc|exec|0x[0-9a-f]{3,}", ops) is not None
        dpp = "row_" in l or "quad_perm" in l
        c = "fast" if (op in FAST and not sgpr and not dpp) else "slow"
        return COST[c], c
    if k == "valu_f64":
        return (COST["rcp_f64"], "rcp_f64") if op == "v_rcp_f64" else (COST["valu_f64"], "valu_f64")
    if k == "wait/nop":
        return COST["nop"], "nop"
    return COST[k], k


def count(a, b):
    c = dict.fromkeys(KINDS, 0)
    for l in lines[a:b]:
        k = kind(l)
        if k in c:
            c[k] += 1
    return c


grp = (at("sg14_A") - at("sg15_A"))
# round 5: probe B's screen of the NEXT move runs inside the wait for the candidates' fetch (generator switch PS); a move then skips its
# own screen B unless the pre-screen was not possible (row crossing 1 in 64, near-wall pass, further rounds, unsafe probe)
HAS_PS = lab("psx") in idx
ps_frac = (float(sys.argv[4]) if len(sys.argv) > 4 else 0.98) if HAS_PS else 0.0
phases = [
    ("move head, probe B's compact copy", at("move"), at("nob1"), 1.0),
    ("screen A: range test, computed jump", at("nob1"), at("sg15_A"), 1.0),
    ("screen A: ONE group of 4 slots (x %.2f executed)" % groups_per_pass, at("sg15_A"), at("sg14_A"), groups_per_pass),
    ("screen A: exit (flag words to their slots)", at("sfinL_A"), at("sdone_A"), 0.6),
    ("fix A: unsafe probe test", at("sdone_A"), at("nofa"), 1.0),
    ] + ([("screen B: is it pre-screened?", at("nofa"), at("nofa") + 3, 1.0)] if HAS_PS else []) + [
    ("screen B%s: range test, computed jump" % (" (x %.2f: not pre-screened)" % (1 - ps_frac) if HAS_PS else ""),
     at("nofa") + (3 if HAS_PS else 0), at("sg15_B"), 1.0 - ps_frac),
    ("screen B: ONE group of 4 slots (x %.2f executed)" % groups_per_pass, at("sg15_B"), at("sg14_B"), groups_per_pass * (1.0 - ps_frac)),
    ("screen B: exit", at("sfinL_B"), at("sdone_B"), 0.6 * (1.0 - ps_frac)),
    ("fix B: unsafe probe test", at("sdone_B"), at("nofb"), 1.0),
    ("pass set-up: log-uniform asked for, near-wall test, two hand-overs through the list, probes, side sources, "
     "displacement asked for, exclusions by compare, wall dz", at("nofb"), at("wdz_mf"), 1.0),
    ] + ([("pre-screen of the next move's probe B: tests, range test, jump", at("wdz_mf"), at("sg15_P"), 1.0),
          ("pre-screen: ONE group of 4 slots (x %.2f x %.2f)" % (groups_per_pass, ps_frac), at("sg15_P"), at("sg14_P"), groups_per_pass * ps_frac),
          ("pre-screen: exit, unsafe cells, flag", at("sfinL_P"), at("psx"), 0.6 * ps_frac)] if HAS_PS else []) + [
    ("fp64 body (both probes, one pass)", at("psx") if HAS_PS else at("wdz_mf"), at("nolj_mf"), 1.0),
    ("side pair captured", at("nolj_mf"), at("mgR0"), 1.0),
    ("reduction (8 sums) + Metropolis step", at("mgR0"), rej_branch + 1, 1.0),
    ("accept path (x %.2f accepted)" % accept, rej_branch + 1, at("reject"), accept),
    ("Fm of particle n+1 (side result), its proposal, fixed-point copies", at("reject"), rowtest, 1.0),
    ("row test; row fill + issue priority (1 move in 64)", rowtest, at("nocross"), 1.0 / 64),
    ("loop bookkeeping", at("nocross"), end + 1, 1.0),
]
print("%s: steady copy of the move, static instruction counts by phase (lines %d..%d of the body)" % (path.split("/")[-1], at("move"), end))
print("%-64s" % "phase" + "".join("%9s" % k for k in KINDS) + "%9s%10s" % ("all", "weighted"))
tot = dict.fromkeys(KINDS, 0.0)
wsum = 0.0
for name, a, b, w in phases:
    c = count(a, b)
    n = sum(c.values()) - c["wait/nop"]
    print("%-64s" % name[:64] + "".join("%9d" % c[k] for k in KINDS) + "%9d%10.1f" % (n, n * w))
    for k in KINDS:
        tot[k] += c[k] * w
    wsum += n * w
print("%-64s" % "per move, weighted with the executed fractions" + "".join("%9.1f" % tot[k] for k in KINDS) + "%9s%10.1f" % ("", wsum))
# the same main line priced at the measured issue costs: SIMD cycles per move, by cost class
cyc = {}
cnt = {}
for name, a, b, w in phases:
    for l in lines[a:b]:
        c, cl = cost(l)
        if cl:
            cyc[cl] = cyc.get(cl, 0.0) + c * w
            cnt[cl] = cnt.get(cl, 0.0) + w
print()
print("priced at the issue costs measured in this kernel (SIMD cycles per wave-instruction; four wavefronts per SIMD issue ONE instruction at a time):")
for cl in ("fast", "slow", "valu_f64", "rcp_f64", "salu", "branch", "lds", "vmem", "smem", "nop"):
    if cl in cnt:
        print("   %-10s %7.1f instructions x %4.2f = %7.1f cycles" % (cl, cnt[cl], COST[cl], cyc[cl]))
print("   sum %.0f SIMD cycles per move (static main line, executed fractions as above; cold pieces -- second hand-over 15 %% of the probes, "
      "near-wall passes, further rounds -- not included)" % sum(cyc.values()))
print("   32-bit VALU: %.1f fast + %.1f slow forms (the PMC classes cannot split them: this is the split behind the line's bracket)" % (cnt.get("fast", 0), cnt.get("slow", 0)))
