#!/usr/bin/env python3
"""gen_sweep_ma.py OUT.inc [NS [MODE [WPR]]] -- writes the body of one hand-scheduled gfx950 sweep kernel (one
`asm volatile` statement of smcx_sweep_ma.hip): one wavefront per replica with NS = 64, 32 or 16 particles
per lane (32 NS < N <= 64 NS), or several (mode z8w).  Same algorithm and the same arithmetic per pair as sweep_kernel_mi
(smcx_sweep_mi.hip: integer screen, fp64 decision and evaluation of every candidate, SMC.c:278-351); every
instruction, register and wait is chosen here instead of by hipcc -- on this chip a wavefront's time is its
instruction count.

MODE  kernel                what differs
 ""   sweep_kernel_ma<NS>   particle l in lane l % 64, slot l / 64 (rotated so that the moving particle is in slot 0);
                            packed int16 x,y in registers, int16 z in LDS; every slot screened, 5.5 instr per slot+probe
 zb   sweep_kernel_mb64     cells in z order (zsort_kernel: Rs, loc); a probe screens only the 4-slot groups whose z
                            range can reach it (computed jump into the unrolled pass); specials on free lanes;
                            issue priority from the progress of the SIMD's other wavefronts
 zbc  (diagnostic of zb)    every ranged pass followed by the full pass; counts the bits the ranged one lacks
 z8   sweep_kernel_mc<NS>   zb with ONE word per cell (int16 z | int8 x | int8 y, units of L/256), screened by
                            v_sub_u32 + v_dot4_i32_i8 + v_alignbit_b32; no z words in LDS -- the benchmark's kernel
 z8c  (diagnostic of z8)    the fp64 cutoff test of every cell beside every pass; counts unflagged pairs, and the
                            executed work: passes, 4-slot groups screened, candidate bits
 z8w  sweep_kernel_mc64x4   z8 for 4 (NS = 64) or 8 (NS = 32) wavefronts per replica, 8192 < N <= 16384: every wave owns a
      sweep_kernel_mc32x8   slab of the z order and runs the whole move loop; reductions completed across waves via LDS
      sweep_kernel_mc32x4   (NS = 32 with 4 wavefronts: 4096 < N <= 8192)
 z8wc (diagnostic of z8w)   as z8c, every wave testing its own cells
 z8t  sweep_kernel_mt16x2   TWO TEAMS of wavefronts per replica (round 3): WPR = 2 K wavefronts, team A = waves 0..K-1, team B
      sweep_kernel_mt64x8   = waves K..2K-1; BOTH teams hold all cells (wave w owns slab w mod K of the z order).  Team A
                            screens, fetches and evaluates probe A (the proposal) while team B does the same for probe B (the
                            next particle) -- without the pair (n, n+1), which team B's slab-0 wave evaluates for BOTH outcomes
                            on two lanes (old position of n from the row cache, proposal from Q) and keeps out of the reduction.
                            ONE exchange per move: every wave writes its row-layout partial sums, team B's slab-0 wave the two
                            side results; s_barrier; every wave sums team A's partials (-> Fn), team B's (-> Fm of the next
                            particle without the side pair), takes the Metropolis decision, adds the side result that applies
                            and forms the next proposal.  A move's dependent chain is one probe long instead of two and a
                            wavefront issues ~40 % fewer instructions: for the latency-bound configurations (N = 1024 x 1024
                            replicas: one wavefront per SIMD; N = 16384 x 256: eight per replica, 61 % of their cycles waiting)
 z8tc (diagnostic of z8t)   as z8wc

A move (iteration i of a run; particle n = first + i; ma: in register slot 0 of lane tl; zb/z8: in cell locA):
  B-COPY  probe B = current position of particle n+1: compact copy by v_readlane (ma: from its owner lane; zb: from
          the row registers rxy/rzl, filled with the row cache every 64 moves)
  SCREEN  ma: 16 groups of 4 slots x 2 probes, 44 instructions each, z words read two groups ahead;
          zb/z8: the groups in z reach only, 22 / 12 instructions per group and probe
  FIX     unsafe / exclusion bits; the first candidate of either probe picked, its fp64 position asked for
  PROBE A wall sites + plane (ma: lanes 0..M2; zb: the first lanes without a candidate; position and coefficients from
          a 32-byte table row per lane), candidates; ONE fp64 body; reduce4 leaves e, fx, fy, fz in the four 16-lane
          rows of one register
  MET     Metropolis step in "row layout": row 0 carries the energies, rows 1-3 the x, y, z components, so
          g = Fn - Fm, h = Fn + Fm and dX = Fm A/T + displ are one instruction each and
          arg = sum over rows of  h (dX/2 + A/(4T) g)   [= dX.(Fn+Fm)/2 + deltaW: g.g + 2 g.Fm = g.h]
                plus 4 (eA - eB) from row 0            [= Un - Um, exactly]
  PROBE B the same body + the pair with particle n where the move left it (LDS cache p0[tl]) on the side lane;
          its result vector FmV stays in a register for the next move
  NEXT    proposal of particle n+1 in row layout: q = p0[.] + (FmV A/T + displ) per row, wrap of rows 1-2,
          fixed-point copies by one v_mul / v_rndne / v_cvt for all three coordinates
Registers: v64..v(63+NS) = the cells of this lane (packed x,y / one word); everything else v0..v63; s0..s95 named
below; the inline-asm operands (lane id, kernarg pointer, block id) live in v0 and s96+.
Hazards are padded by the rules hipcc applies on gfx950 (read off its output): 2 wait states between a
VALU write of an SGPR/VCC and a VALU read of it, 1 before v_readlane / v_readfirstlane of a fresh VGPR,
2 before DPP or v_permlane*_swap of a fresh VGPR, 1 after v_rcp_f64, 3 after v_dot2 / v_dot4 (met by the interleave).
Generator switches for experiments: SMCX_GEN_PRIO_MODE=static (priority = wave slot), SMCX_GEN_PRIO_SHIFT.
"""
import os
import re
import sys

# The experiment switches below change the generated kernel (SMCX_GEN_FAKEFETCH even its results).  They belong to VARIANT builds
# (csrc/Makefile passes SMCX_GEN_ALLOW=1 there); the product and the diagnostic build refuse a stale one in the environment.
_stale = sorted(k for k in os.environ if k.startswith("SMCX_GEN_") and k != "SMCX_GEN_ALLOW")
if _stale and os.environ.get("SMCX_GEN_ALLOW") != "1":
    sys.exit("gen_sweep_ma.py: generator switches %s are set but this is not a VARIANT build (make VARIANT=name GENENV=...); "
             "unset them" % ", ".join(_stale))

out = []


MUTE = [False]     # while set, nothing is emitted (the sections a variant replaces are skipped without re-indenting them)


REDIR = [None]     # while set to a list, E() appends there (whole sections emitted among the cold pieces)


FORCE_TAG = [""]   # while set, lines without a tag of their own get this one (whole helpers emitted for one copy of the move only)


def E(txt="", tag=""):
    if MUTE[0]:
        return
    tag = tag or FORCE_TAG[0]
    for ln in txt.strip("\n").split("\n"):
        ln = ln.strip()
        if ln and not ln.startswith("//"):
            (out if REDIR[0] is None else REDIR[0]).append(ln + tag)


def GW(txt):
    """as G where hasAw is hasA (one wavefront per replica); with several, hasAw is 0 on the waves without specials"""
    (E if W4 else G)(txt)


def G(txt):
    """lines that only the GENERIC copy of the move keeps (tests of hasA / hasB: the steady copy runs the moves that have
    both a proposal to decide and a next particle, i.e. all but the first and the last of a run)"""
    E(txt, " @G" if PEEL else "")


def SO(txt):
    """lines that only the STEADY copy of the move keeps"""
    if PEEL:
        E(txt, " @S")


# ---------------------------------------------------------------------------------------------- registers
ZS = 4
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 64      # particles per lane: 64, 32 or 16
assert NS in (16, 32, 64)
# "zb": z-binned storage.  The register/LDS cells hold the particles in the order of a z sort done before the
# launch (Rs = positions in cell order, loc = cell of each particle); a probe screens only the 4-slot groups
# whose z range (kept per group, widened by accepted moves) can reach it.
MODE = sys.argv[3] if len(sys.argv) > 3 else ""
ZBC = MODE == "zbc"                                       # diagnostic: every pass also runs the full screen and counts
                                                          # the bits the ranged pass lacks (must be none)
# "z8": zb with ONE 32-bit word per cell -- z as int16 in bits 0..15, x and y as int8 in bits 16..23 / 24..31, all
# in units of L/256 -- screened by v_sub_u32 + v_dot4_i32_i8 + v_alignbit_b32 (3 instructions per slot and
# probe, no z words in LDS).  The byte-wise squares alias |dz| >= 128 units; such cells are far outside the
# cutoff, cost at most a wasted evaluation, and the group ranges keep them out of the passes anyway.
# "z8c": its diagnostic build: the fp64 cutoff test of every cell beside every pass, counting unflagged pairs.
TT = MODE in ("z8t", "z8tc")                              # two teams of wavefronts per replica (see above)
PRIO = MODE not in ("z8w", "z8wc", "z8t", "z8tc")         # issue priority from the SIMD neighbours' progress (one-wave kernels)
PRIO_MODE = os.environ.get("SMCX_GEN_PRIO_MODE", "rotate")
NT = " nt" if os.environ.get("SMCX_GEN_NT") == "1" else ""   # experiment: streaming hint on once-per-sweep data
STAMPS = os.environ.get("SMCX_GEN_TT_STAMPS") == "1"        # z8t diagnostic: where a move's time goes (before / at the barrier)
# one-wave kernels, measurement only: SMCX_GEN_IVAL=list puts a label L_ivp<k>_<n> at every mark(k) (their order in the generated
# move is read off the body); SMCX_GEN_IVAL=a:b sums, per replica, the cycles from mark a to mark b of every move in m0 (which these kernels
# use only as the index register of s_set_gpr_idx_on on the accept path: saved in s101 around it in these builds) and returns the sum in the end stamp's s_memrealtime slot (smcx_debug_clk_rows column 3).  Either stamp drains the
# LDS / scalar-memory counter (s_memtime's result is awaited there): a few cycles of perturbation per move.  Needs s100, s101
# among the kernel's clobbers (tools/probes/ival_phases.sh builds the variants).
IVAL = os.environ.get("SMCX_GEN_IVAL", "")
IVA, IVB = (int(x) for x in IVAL.split(":")) if (":" in IVAL and not IVAL.startswith("count")) else (None, None)
IVC = int(IVAL.split(":")[1]) if IVAL.startswith("count:") else None      # count:k -- m0 counts the executions of mark k
IVN = [0]
FAKE = os.environ.get("SMCX_GEN_FAKEFETCH") == "1"          # TIMING experiment only (wrong results): every candidate fetch reads cell 0
PRIO_SHIFT = int(os.environ.get("SMCX_GEN_PRIO_SHIFT", "14"))   # ... every 2^14 ticks of the 100 MHz clock (164 us)
Z8C = MODE in ("z8c", "z8wc", "z8tc", "z8lc")
# "z8l" / "z8lc": z8 with 16 cells per lane and the fp64 positions of ALL cells (24 KB) in LDS -- the few-replica form of
# N <= 1024 (sweep_kernel_ml16): with one wavefront per SIMD nothing hides the round trip of a candidate fetch to L2
ZL = MODE in ("z8l", "z8lc")
# "z8w": z8 for FOUR wavefronts per replica (8192 < N <= 16384): wave w owns the cells 4096 w .. 4096 w + 4095 of the
# z order (its own 16 groups and ranges) and runs the whole move loop itself -- same scalar state, own copy of the row
# cache -- except that the wall sites, plane and side pair live on wave 0, a cell is written by its owner only, and
# the two reductions of a move are completed across the waves through LDS (fixed order, so every wave takes the
# same Metropolis decision); operand %3 = the wave's index
W4 = MODE in ("z8w", "z8wc") or TT
WPR = (int(sys.argv[4]) if len(sys.argv) > 4 else 256 // NS) if W4 else 1   # wavefronts per replica: 4 (NS = 64) or 8 (NS = 32)
assert WPR in (1, 2, 4, 8, 16)                            # (NS = 32 with 4: 4096 < N <= 8192; two teams: 16 x 2, 64 x 8, 32 x 16)
KS = WPR // 2 if TT else WPR                              # slabs of the z order = wavefronts that share out the cells
WSH = (NS * 64).bit_length() - 1                          # cell >> WSH = the wave that owns it
SLOTF = ((NS.bit_length() - 1) << 16) | 6                 # s_bfe field of the slot inside a cell index
Z8 = Z8C or W4 or MODE in ("z8", "z8l")
ZB = ZBC or Z8 or MODE == "zb"
# slots per z group: 4 (256 cells).  Groups of 2 slots (128 cells; the code below is written for either) were built and
# measured in round 4 (SMCX_GEN_GS=2 with zsort_kernel sorting into groups of 128): at the benchmark start a group of 128 is ONE
# lattice layer, and the ranges widened by a sweep of moves put 6.4 of 32 such groups in reach where 3.4 of 16 are: 6 % fewer
# cells screened at 13 % more per cell (an exit test per 2 slots) -- 8.90 against 8.97 ms per sweep, while the dense N = 16384
# start lost (24.2 -> 27.0 ms: more lanes with two candidates).  Not adopted.
GS = int(os.environ.get("SMCX_GEN_GS", "4")) if Z8 else 4
GSH = GS.bit_length() - 1
NG = NS // GS                                             # z groups of this lane's cells
# The move loop is emitted twice for the z-ordered kernels: a GENERIC copy that tests hasA (a proposal to decide: not in the
# first pass of a run, which only evaluates the first particle) and hasB (a next particle: not in the last pass), and a
# STEADY copy for every other pass without those eight test-and-branch pairs (16 scalar instructions of ~180 per move).
PEEL = ZB and os.environ.get("SMCX_GEN_NOPEEL") != "1"
# "MG": both probes of a move in ONE pass of the fp64 body (z8 with one wavefront per replica).  After the two screens every
# lane with a candidate hands its lowest one over to a WORKING lane through a list in LDS: lanes 0..31 work for probe A (the
# proposal of particle n), lanes 32..63 for probe B (the current position of particle n+1) -- first the plane / wall sites
# (and, in B's half, the pair (n, n+1) for BOTH outcomes of the move on two lanes, kept out of the sums as in the two-team
# kernels), then the candidates in the order of their owner lanes.  One body, ONE reduction (4 accumulators x 2 halves -> 8
# sums in 8-lane groups: e | fy in row 0, fx | fz in row 1 of each half), Metropolis and proposal in that "group layout".
# Per move ~130 instructions fewer than two probes one after the other (of ~540).
MG = Z8 and not TT and os.environ.get("SMCX_GEN_NOMERGE") != "1"      # (switch for the A/B: SMCX_GEN_NOMERGE=1)
# ... also with several wavefronts per replica (z8w): every wave merges the two probes over ITS cells; the 8 sums of the waves
# are added in ONE exchange per move (two before); the wall sites, planes and the side pair work on the slab-0 wave
# z8t with 16 cells per lane ("LP"): the fp64 positions of all cells (24 KB, what candidates are fetched from) and the
# wall table live in LDS at offset 0, shared by the two wavefronts of the replica; every other LDS area moves up
# z8t, round 4 ("TL"): the candidates of a team's probe are handed over to working lanes through a list in LDS as in the
# merged pass (all 64 lanes work for the ONE probe of the wave; wall lanes and side lanes on fixed lanes in front), so a lane
# with two candidates costs a second hand-over instead of a second pass of the fp64 body (round 3's PF2)
TL = TT and os.environ.get("SMCX_GEN_TTLIST", "1") != "0"
# z8t: which slab's wave of team B carries probe B's wall lanes / the side pair (team A's wall lanes: its slab-0 wave).  Round 4's
# stamps (profiles/r04_two_team_phases.txt) have team B's slab-0 wave, which carried both, the last at the barrier by 280 cycles over
# its team mates; on separate waves each fits into that slack.
TT_WALLB = int(os.environ.get("SMCX_GEN_TT_WALLB_SLAB", "0"))
TT_SIDE = int(os.environ.get("SMCX_GEN_TT_SIDE_SLAB", "0"))
TTCAP = int(os.environ.get("SMCX_GEN_TTCAP", "63"))
# round 5, the one-wavefront merged kernels (mc64/32/16, ml16): five small trims of the steady move, built, correct (192 GPU tests) and
# NOT adopted (switch SMCX_GEN_TRIM5=1 builds them) -- the accept path's energy update by one fma + two lane reads (was 7
# instructions), its position registers by three 64-bit moves (6), the unsafe-z bit update behind a "some particle is unsafe" flag
# in s100 (8 -> 2), the group range with RZ in place, the log-uniform's load with register + immediate offset: 424 -> 417
# instructions per move on the main line, predicted -1.8 %; measured (profiles/r05_trim5_ab.txt, A/B/A/B/A/B in one session)
# config 3 8.611 -> 8.577 ms per sweep (-0.4 %) and config 2 1.203 -> 1.221 (+1.4 %: its lone wavefront pays for the longer
# dependent chain of the one-instruction energy update).  The accept path is not where the issue port is short of slots.
# (A build with the switch on must also list "s100" among the clobbers of sweep_kernel_mc64/32/16/ml16 in smcx_sweep_ma.hip.)
TRIM5 = os.environ.get("SMCX_GEN_TRIM5", "0") == "1"
T5 = TRIM5 and MG and not W4
# TIMING experiments only (WRONG results; VARIANT builds): what a phase of the move costs, by leaving it out -- SMCX_GEN_ABL = a
# comma-separated list of noaccept (every move rejected), nobody (the fp64 body of round 0 jumped over), noscreenA / noscreenB (a probe's
# screen jumped over: no candidates), nosides (no side pair).  tools/sessions/r05_session17.sh, profiles/r05_ablation_mc64.txt
ABL = set(x for x in os.environ.get("SMCX_GEN_ABL", "").split(",") if x)
# "PS" (one-wave z-ordered kernels, steady copy of the move): probe B's screen of the NEXT move -- particle n+2 against the cells,
# which does not depend on this move's outcome except through the moving particle's own cell -- runs between the issue of the
# candidates' fetch and the wait for it (profiles/r05_ival_phases_mc64.txt: that wait is 770 cycles of a wavefront's 4850 per move
# with 12 instructions in it; a screen is 620), into the flag words wb0/wb1 that this move's hand-over has emptied.  psv (the high
# word of the hand-over's scratch pair hB, dead between the hand-over and the next one) says the next move finds them set and skips
# its own screen B.  Not done -- psv = 0, the next move screens as before -- when further rounds still need the flag words, in the
# near-wall form of the pass, when the order crosses into the next row of 64 (tl = 63), when particle n+2 is beyond the safe z range,
# and in the generic copy.  On acceptance the moving particle's cell gets its bit set in the pre-screened words (its bytes, its
# group's range and its unsafe bit may have changed under them: a candidate too many costs one evaluation, one too few a pair).
# "EVM" (steady copy of the merged pass): the wait for the next move's displacement -- asked for a whole pass earlier, long there --
# stands BEFORE the Metropolis decision instead of behind the accept path, where s_waitcnt vmcnt(0) also waited for the accepted
# move's four stores to Rs and R to be acknowledged (nothing later in the move reads them back except through the same
# wavefront's own loads, which the memory pipeline keeps in order behind the stores).
# "ELDS" (same copy): the two LDS reads behind the decision -- particle n+1's cached position for its proposal and the side pair's
# result for the outcome -- had their latency in the open (ds_read, s_waitcnt, use).  They are asked for before the Metropolis
# chain instead (both outcomes of the side result: v[40:41] = "rejected", v[38:39] = "accepted", copied over it on the accept path;
# the position in v[36:37]): the body's temporaries v36..v41 are free from the reduction to the next move's pre-screen.
ELDS_ON = os.environ.get("SMCX_GEN_ELDS", "1") == "1"      # (A/B: make VARIANT=noelds GENENV="SMCX_GEN_ELDS=0")
EVM = os.environ.get("SMCX_GEN_EVM", "1") == "1"          # (A/B: make VARIANT=noevm GENENV="SMCX_GEN_EVM=0")
PS_ON = os.environ.get("SMCX_GEN_PS", "1") == "1"          # (A/B: make VARIANT=nops GENENV="SMCX_GEN_PS=0")
ANYU = "s100"      # T5: 1 while some cell of this replica carries the unsafe-z bit (set where the cells are built and on acceptance)       # working lanes of a two-team wave's list (64 = round 4's value: drops items, see tt_assign)
# "XC" (merged pass, steady copy of the move): the cells that are no neighbours of a probe -- the moving particle n
# for both, particle n+1 itself for probe B -- keep their candidate bits and travel through the hand-over list like any other; the
# working lane that reads one of them as its item is taken out of the pass by a compare of the item with the cell (4 instructions
# where clearing the three bits in their owner lanes took 18).  The generic copy (moves without a proposal or without a next
# particle, where locA / locB may be stale) keeps the bit clearing.
XC = MG and PEEL and os.environ.get("SMCX_GEN_XCMP", "1") != "0"
# ... and in the two-team kernel with the list hand-over ("XCT"): a wave's list items are cells of ITS slab (slot << 6 | lane), so
# the compare is with cell - (slab << WSH), which no item equals when the cell belongs to another wave (3 instructions per
# exclusion on every wave, where the bit clearing took 4 on the waves that do not own the cell and 10 on the one that does)
XCT = TL and PEEL and os.environ.get("SMCX_GEN_XCMP", "1") != "0"
PF2 = TT and NS >= 32 and not TL     # z8t with many cells: the first TWO candidates of a lane are fetched together (dense states: the
                          # second round's memory round trip was the longest stretch of the slowest wavefront's move)
LP = (TT and NS == 16) or ZL
ELDS = ELDS_ON and MG and PEEL and not W4
PS = PS_ON and MG and PEEL and not W4 and (not LP or os.environ.get("SMCX_GEN_PSLP", "0") == "1")      # (sweep_kernel_mc64 / mc32 / mc16
                                                          # and their diagnostic builds; ml16 -- the fetch is an LDS read there -- with SMCX_GEN_PSLP=1)
assert not ZL or NS == 16
LDS_RS, LDS_WT = 0, NS * 64 * 24
LDS_BASE = LDS_WT + 1024 if LP else 0
LDS_P0 = LDS_BASE if Z8 else (NS // 2) * 256              # after the int16 z words
LDS_GB = LDS_P0                                           # zb: (min, max) z of each group while the copies are built (p0 is filled afterwards)
LDS_CNT = LDS_P0 + 65 * 24 + 8                            # zbc: per lane (candidates, bits missing from the ranged pass)
LDS_WAVE = 2048                                           # z8w: each wave's copy of the row cache (v1 = wave * LDS_WAVE)
LDS_X = LDS_BASE + WPR * LDS_WAVE                         # z8w: exchange area [2 buffers][WPR waves][64 lanes] doubles
LDS_SIDE = LDS_X + 2 * WPR * 512                          # z8t: [2 buffers][old, new][e, fx, fy, fz] of the side pair
if W4:
    LDS_CNT = LDS_X + 2 * WPR * 512 + 128                 # z8wc: 8 counter words per wave behind the exchange and side areas
LDS_TM = LDS_SIDE + 128                                   # z8t stamps variant: per wave {t0, cycles before the barrier, cycles at it}
LDS_LIST, LDS_SIDEM = LDS_BASE + 2048, LDS_BASE + 2304    # mg: hand-over list [64] words; side results [old, new][e, fx, fy, fz]
if W4:              # z8w, z8t: the list inside the wave's own block (behind its row cache; addressed through v1)
    LDS_LIST = 1600
if W4 and not TT:   # z8w: the side results -- [2 buffers by the parity of the move] -- behind the exchange area
    LDS_SIDEM = LDS_X + 2 * WPR * 512
LANE, KARG, REP, WAVE = "%0", "%1", "%2", "%3"
# z8t: the slab of a wave is wave mod K (team B = waves K .. 2K-1), formed with s_and_b32 in a scratch register where
# it is needed (the inline-asm statement has no SGPR operand to spare); otherwise slab = wave

V = dict(zaddr=1, uns0=2, uns1=3, wa0=4, wa1=5, wb0=6, wb1=7, axy=8, bxy=9,
         zA=10, zB=12,            # z words of the screen: two pairs
         t=14,                    # v14..v25: screen temporaries; afterwards XA (v14-19) and XB (v20-25)
         XA=14, XB=20, C=26, D=30, M=36, dr2=40, ir2=42, T=44, S6=46, F=48,
         acc=50,                  # e, fx, fy, fz: v50..v57
         wdz=58, FmV=60, DdV=62)
XY0 = 64


if ZB:
    # the probes' packed x,y live in SGPRs (copied to a screen temporary per pass); v8/v9 carry the current
    # 64-particle row instead: rxy = packed x,y, rzl = z16 | unsafe << 16 | cell << 17 of particle row*64 + lane.
    # The walls' dz goes to the F registers (free until the body's force step); v58/v59 = the groups' z ranges
    # (lane g: lowest z - RZ, highest z + RZ, in z units).
    del V['axy'], V['bxy']
    V.update(rxy=8, rzl=9, pxy=20, wdz=V['F'], gloR=58, ghiR=59)


def v(name, i=0): return "v%d" % (V[name] + i)
def vp(name, i=0): return "v[%d:%d]" % (V[name] + 2 * i, V[name] + 2 * i + 1)
def xy(k): return "v%d" % (XY0 + k)


S = dict(Rg=0, displ=2, uni=4, dK=6, uK=8, wtab=10, rec=12, clk=14,
         L=16, invL=18, cut2=20, invT=22, AoT=24, Ao4T=26, toFix=28, zFix=30, zsafe=32, halfLz=34, Lz=36, invLz=38,
         neg24=40,
         N=42, negC=43, M2=44, nsw=45, sw=46, run=47, n0=48, first=49, len=50, i=51, tl=52, rot=53, jacc=54,
         hasA=55, hasB=56, cross=57, lb=58, azz=59, az16=60, ua=61, bzz=62, ub=63, cbase=64, vbase=65,
         Q=66, lu=72, nlu=74, E=76, haveA=78, wallM=80, haveB=82, planeM=84, sideM=86, t=88)   # s88..s95 scratch


if ZB:
    # displ/uni/rec/clk are re-derived from the kernel arguments where they are used; their registers hold
    # Rs, the cells of particles n and n+1, the probes' packed x,y and the compact copy of particle row*64 + 64
    # ... n0, vbase and cbase live in temporaries where a run starts; the lanes of the wall sites, the plane and
    # the side pair are chosen per probe among the lanes without a candidate (wallM/planeM = probe A's, wallB/
    # planeB = probe B's, sideL = the side pair's lane)
    for k in ('displ', 'uni', 'rec', 'clk', 'n0', 'vbase', 'cbase', 'sideM'):
        del S[k]
    S.update(Rs=2, locA=4, locB=5, axys=12, bxys=13, nxy=14, nzl=15, first=65, planeB=48, sideL=64, wallB=86)


if W4:
    # bzz: real cells of this wave ; az16: M2 on wave 0, -1 elsewhere ; azz: hasA on wave 0, 0 elsewhere
    S.update(Nw=S['bzz'], M2w=S['az16'], hasAw=S['azz'])


if Z8 and not W4:
    S.update(RZ=S['azz'])     # the reach of a group's z range in z units: azz is not used by the one-word screen
if MG:
    # wlp = (plane lanes of half A, of half B) = (hasA & walls, hasB & walls): bit 0 of either word -- the planes work on
    # lanes 0 and 32; stB = first candidate lane of half B (behind its plane and the two side lanes); sidesHi = the side
    # lanes' bits in the high word; nearA / nearB: the probe is within the cutoff of a wall (its sites join in: cold path);
    # a0s, b0s: the plane's coefficients; hA, hB: scratch pairs; accf: this move was accepted
    S.update(hA=78, a0s=80, b0s=82, hB=84, znear=86, wlp=48, stB=57, sidesHi=58, nearA=60, nearB=62, accf=64)
    if W4:   # the real cells of this wave move to azz (the merged pass needs no hasAw; M2w and hasAw are never written: the
        S.update(Nw=S['azz'])   # state of the pass -- wlp, stB, sidesHi -- is per wave, empty off the slab-0 wave)
if TT:
    S.update(wbase=S['cross'])    # XCT: slab << WSH (cross is not used by the z-ordered kernels)
    # team B's waves never use probe A's masks: haveA's low word holds the lane of the side pair evaluated with the
    # PROPOSAL (sideL: with n's current position); its high word carries the accept flag from the Metropolis step to
    # the point where the side result that applies is added (every wave; team A's haveA is dead by then)
    S.update(sideN=S['haveA'], accf=S['haveA'] + 1)


def s(name, i=0): return "s%d" % (S[name] + i)
def s_Nw(): return s('Nw') if W4 else s('N')
def s_M2w(): return s('M2w') if W4 else s('M2')
def s_hasAw(): return s('hasAw') if W4 else s('hasA')
def sp(name, i=0): return "s[%d:%d]" % (S[name] + 2 * i, S[name] + 2 * i + 1)
def st(i): return "s%d" % (S["t"] + i)
def stp(i): return "s[%d:%d]" % (S["t"] + i, S["t"] + i + 1)
def st4(i): return "s[%d:%d]" % (S["t"] + i, S["t"] + i + 3)


ONE_HI = "0x3ff00000"

# kernarg layout (struct MaArgs in smcx_sweep_ma.hip)
K_R, K_DISPL, K_UNI, K_OFFS, K_OBS, K_REC, K_WTAB, K_CLK = 0x0, 0x8, 0x10, 0x18, 0x20, 0x28, 0x30, 0x38
K_CONST16, K_CONST8, K_INTS, K_M2 = 0x40, 0x80, 0xa0, 0xb0
K_RZ, K_RS, K_LOC, K_SW0, K_DBG, K_PRIO = 0xb4, 0xb8, 0xc0, 0xc8, 0xd0, 0xd8   # zb
K_ZNEAR = 0xe0                                                                 # mg: |z| from which a wall's sites are in reach

# ---------------------------------------------------------------------------------------------- prologue
E(f"""
s_load_dwordx2 {sp('Rg')}, {KARG}, {K_R}
s_load_dwordx4 {st4(0)}, {KARG}, {K_OBS}
s_load_dwordx4 {st4(4)}, {KARG}, {K_WTAB}
s_load_dwordx16 s[16:31], {KARG}, {K_CONST16}
s_load_dwordx8 s[32:39], {KARG}, {K_CONST8}
s_waitcnt lgkmcnt(0)
// stp(0) obs, stp(2) rec, stp(4) wtab, stp(6) clk
{"" if ZB else f"s_mov_b64 {sp('rec')}, {stp(2)}"}
s_mov_b64 {sp('wtab')}, {stp(4)}
{"" if ZB else f"s_mov_b64 {sp('clk')}, {stp(6)}"}
s_lshl_b32 {st(2)}, {REP}, 6
s_add_u32 {st(0)}, {st(0)}, {st(2)}
s_addc_u32 {st(1)}, {st(1)}, 0
s_load_dwordx2 {sp('E')}, {stp(0)}, 0x20
s_load_dwordx4 {st4(4)}, {KARG}, {K_INTS}
s_load_dword {s('M2')}, {KARG}, {K_M2}
s_waitcnt lgkmcnt(0)
// stp(4..7) = N, chunk, nsweeps, negC
s_mov_b32 {s('N')}, {st(4)}
s_mov_b32 {s('nsw')}, {st(6)}
s_mov_b32 {s('negC')}, {st(7)}
{'' if ZB else f"s_mul_i32 {s('cbase')}, {REP}, {st(5)}"}
s_mov_b32 {s('neg24')}, 0
s_mov_b32 {s('neg24',1)}, 0xc0380000
// Rg = R + rep * N * 24 ; rec += rep * chunk * 16 ; clk += rep * 32
s_mul_i32 {st(0)}, {s('N')}, 24
s_mul_hi_u32 {st(1)}, {st(0)}, {REP}
s_mul_i32 {st(0)}, {st(0)}, {REP}
s_add_u32 {s('Rg')}, {s('Rg')}, {st(0)}
s_addc_u32 {s('Rg',1)}, {s('Rg',1)}, {st(1)}
""")
if not ZB:
    E(f"""
    s_lshl_b32 {st(0)}, {s('cbase')}, 4
    s_add_u32 {s('rec')}, {s('rec')}, {st(0)}
    s_addc_u32 {s('rec',1)}, {s('rec',1)}, 0
    s_lshl_b32 {st(0)}, {REP}, 5
    s_add_u32 {s('clk')}, {s('clk')}, {st(0)}
    s_addc_u32 {s('clk',1)}, {s('clk',1)}, 0
    """)
else:
    # Rs = Rs + rep * NS*64*24 ; the sweep counter runs from sw0 (the host sorts between launches)
    E(f"""
    s_load_dwordx2 {sp('Rs')}, {KARG}, {K_RS}
    s_load_dword {st(3)}, {KARG}, {K_SW0}
    s_mov_b32 {st(0)}, {NS * 64 * 24}
    {f"s_mul_i32 {st(2)}, {REP}, {KS}" if W4 else ""}
    {f"s_and_b32 {st(1)}, {WAVE}, {KS - 1}" if TT else ""}
    {f"s_add_u32 {st(2)}, {st(2)}, {st(1) if TT else WAVE}" if W4 else ""}
    s_mul_hi_u32 {st(1)}, {st(0)}, {st(2) if W4 else REP}
    s_mul_i32 {st(0)}, {st(0)}, {st(2) if W4 else REP}
    s_waitcnt lgkmcnt(0)
    s_add_u32 {s('Rs')}, {s('Rs')}, {st(0)}
    s_addc_u32 {s('Rs',1)}, {s('Rs',1)}, {st(1)}
    """)
    if W4:
        # Rs points at this wave's 4096 cells; Nw = how many of them hold a particle; walls and side pair: wave 0
        E(f"""
        // the groups of 256 cells are dealt to the KS wavefronts round-robin (zsort_kernel: group g = local group g / KS of
        // wave g % KS): this wave's real cells = its full groups + the partial last group if that one is its own
        s_and_b32 {st(0)}, {WAVE}, {KS - 1}
        s_lshr_b32 {st(4)}, {s('N')}, 8
        s_and_b32 {st(5)}, {s('N')}, 255
        s_add_u32 {st(6)}, {st(4)}, {KS - 1}
        s_sub_u32 {st(6)}, {st(6)}, {st(0)}
        s_lshr_b32 {st(6)}, {st(6)}, {KS.bit_length() - 1}
        s_lshl_b32 {st(6)}, {st(6)}, 8
        s_and_b32 {st(4)}, {st(4)}, {KS - 1}
        s_cmp_eq_u32 {st(4)}, {st(0)}
        s_cselect_b32 {st(5)}, {st(5)}, 0
        s_add_u32 {s('Nw')}, {st(6)}, {st(5)}
        s_min_i32 {s('Nw')}, {s('Nw')}, {NS * 64}
        {f"s_and_b32 {st(0)}, {WAVE}, {KS - 1}" if TT else ""}
        {f"s_cmp_ge_u32 {WAVE}, {KS}" if TT else ""}
        {f"s_cselect_b32 {st(4)}, {TT_WALLB}, 0" if TT else ""}
        {"" if MG else f"s_cmp_eq_u32 {st(0) if TT else WAVE}, {st(4) if TT else 0}"}
        {"" if MG else f"s_cselect_b32 {s('M2w')}, {s('M2')}, -1"}
        s_mul_i32 {st(0)}, {WAVE}, {LDS_WAVE}
        v_mov_b32 v1, {st(0)}
        """)
    E(f"""
    s_mov_b32 {s('sw')}, {st(3)}
    s_add_u32 {s('nsw')}, {s('nsw')}, {st(3)}
    // a launch that continues a chunk (sw0 > 0) takes the running energy from the previous sweep's record
    s_cmp_eq_u32 {st(3)}, 0
    s_cbranch_scc1 L_e0
    s_load_dwordx2 {stp(4)}, {KARG}, {K_REC}
    s_load_dword {st(0)}, {KARG}, {K_INTS + 4}
    s_waitcnt lgkmcnt(0)
    s_mul_i32 {st(0)}, {st(0)}, {REP}
    s_add_u32 {st(0)}, {st(0)}, {st(3)}
    s_sub_u32 {st(0)}, {st(0)}, 1
    s_lshl_b32 {st(0)}, {st(0)}, 4
    s_waitcnt lgkmcnt(0)
    s_load_dwordx2 {sp('E')}, {stp(4)}, {st(0)}
    s_waitcnt lgkmcnt(0)
    L_e0:
    """)


def cbase_to(dst):
    """zb: dst (s) <- rep * chunk, the replica's first row in the per-sweep arrays (the other kernels keep it in a register)"""
    E(f"""
    s_load_dword {dst}, {KARG}, {K_INTS + 4}
    s_waitcnt lgkmcnt(0)
    s_mul_i32 {dst}, {dst}, {REP}
    """)


def clk_ptr(dst):
    """dst (s pair) <- this replica's row of the clock stamps"""
    if not ZB:
        return sp('clk')
    lo, hi = (int(x) for x in dst[2:-1].split(":"))
    E(f"""
    s_load_dwordx2 {dst}, {KARG}, {K_CLK}
    s_lshl_b32 {st(0)}, {REP}, {11 if (TT and STAMPS) else 5}
    s_waitcnt lgkmcnt(0)
    s_add_u32 s{lo}, s{lo}, {st(0)}
    s_addc_u32 s{hi}, s{hi}, 0
    """)
    return dst


if TT and STAMPS:   # the clk row collects the phase sums of this variant: no start stamp; zero this wave's 16 LDS words
    E(f"""
    s_lshl_b32 {st(4)}, {WAVE}, 6
    v_lshl_add_u32 v14, {LANE}, 2, {st(4)}
    v_mov_b32 v16, 0
    s_mov_b64 exec, 0xffff
    ds_write_b32 v14, v16 offset:{LDS_TM}
    s_mov_b64 exec, -1
    """)
else:
    CLK0 = clk_ptr(stp(2))
    E(f"""
    // start stamp: clk[rep][0..1] = s_memtime, s_memrealtime
    s_memtime {stp(4)}
    s_memrealtime {stp(6)}
    s_waitcnt lgkmcnt(0)
    v_mov_b32 v14, 0
    v_mov_b32 v16, {st(4)}
    v_mov_b32 v17, {st(5)}
    v_mov_b32 v18, {st(6)}
    v_mov_b32 v19, {st(7)}
    s_mov_b64 exec, 1
    global_store_dwordx4 v14, v[16:19], {CLK0}
    s_mov_b64 exec, -1
    {"s_mov_b32 m0, 0" if (IVA is not None or IVC is not None) else ""}
    """)
if not ZB:
    E(f"""
    // masks of the special lanes: wall sites + plane = lanes 0..M2 (none if M2 < 0), plane = lane M2, side pair = lane 30
    s_mov_b64 {sp('wallM')}, 0
    s_mov_b64 {sp('planeM')}, 0
    s_cmp_lt_i32 {s('M2')}, 0
    s_cbranch_scc1 L_nowalls
    s_add_u32 {st(2)}, {s('M2')}, 1
    s_lshl_b64 {sp('wallM')}, 1, {st(2)}
    s_sub_u32 {s('wallM')}, {s('wallM')}, 1
    s_lshl_b64 {sp('planeM')}, 1, {s('M2')}
    L_nowalls:
    s_mov_b32 {s('sideM')}, 0x40000000
    s_mov_b32 {s('sideM',1)}, 0
    """)
# per-lane constant of the row layout: 8 (row - 1) for rows 1..3 (components x, y, z), 0 for row 0 -- where a register is
# free for it (z8 with one wavefront: v1, which holds the z words' address in the other forms; two teams: KPROW)
KROW = "v1" if (Z8 and not W4 and not MG) else None
KC, KL4, KSD = ("v24" if W4 else "v1"), "v10", "v11"    # mg: per-lane constants of the group layout (below; z8w: v1 = the wave's LDS block)
if MG:
    # group layout: in either half, row 0 = [e | fy], row 1 = [fx | fz] in 8-lane groups.  KC = byte offset of the lane's
    # component in a position / displacement triple (x 0, y 8, z 16; the e group idles along with x); KSD = 8 x index of its
    # sum in a side result (e, fx, fy, fz); KL4 = 4 lane.  The plane's coefficients and the near-wall bound in SGPRs.
    E(f"""
    s_load_dwordx2 {sp('znear')}, {KARG}, {K_ZNEAR}
    s_mov_b64 {sp('a0s')}, 0
    s_mov_b64 {sp('b0s')}, 0
    s_cmp_lt_i32 {s('M2')}, 0
    s_cbranch_scc1 L_nopl
    s_lshl_b32 {st(0)}, {s('M2')}, 5
    s_add_u32 {st(0)}, {st(0)}, 16
    s_load_dwordx4 s[{S['a0s']}:{S['a0s'] + 3}], {sp('wtab')}, {st(0)}
    L_nopl:
    v_bfe_u32 v14, {LANE}, 3, 1
    v_bfe_u32 v15, {LANE}, 4, 1
    v_lshlrev_b32 v16, 3, v14
    v_lshlrev_b32 {KC}, v15, v16
    v_lshl_add_u32 v16, v14, 1, v15
    v_lshlrev_b32 {KSD}, 3, v16
    {f"v_lshl_add_u32 {KL4}, {LANE}, 2, v1" if W4 else f"v_lshlrev_b32 {KL4}, 2, {LANE}"}
    s_waitcnt lgkmcnt(0)
    """)
if KROW:
    E(f"""
    v_lshrrev_b32 {KROW}, 4, {LANE}
    v_add_u32 {KROW}, -1, {KROW}
    v_max_i32 {KROW}, 0, {KROW}
    v_lshlrev_b32 {KROW}, 3, {KROW}
    """)
E(f"""
{"" if (W4 or KROW or MG) else f"v_lshlrev_b32 {v('zaddr')}, 2, {LANE}"}
v_mov_b32 {v('uns0')}, 0
v_mov_b32 {v('uns1')}, 0
s_mov_b32 {s('rot')}, 0
""")
SRC = sp('Rs') if ZB else sp('Rg')    # what the compact copies are built from / candidates are fetched from
if W4 and not TT and os.environ.get("SMCX_GEN_W0_PRIO", "1") == "1":
    # z8w: the wavefront that carries the wall sites and the side pair (slab 0) is the last at both exchanges of a move: it
    # gets the higher issue priority (mc32x4 at N = 8192 x 1024: 29.7 -> 28.9 ms per sweep; mc64x4 at 16384 x 512: 42.4 -> 41.6)
    E(f"""
    s_cmp_eq_u32 {WAVE}, 0
    s_cbranch_scc0 L_prioW
    s_setprio 3
    L_prioW:
    """)
if TT:
    # z8t: team B's wavefronts carry the side pair and two exclusions on top of what team A's do and arrive last at the
    # exchange, and where a SIMD holds wavefronts of both teams its arbiter serves the older -- team A's -- first.  Team B
    # gets the higher issue priority for the whole kernel: config 2 1.56 -> 1.41 ms per sweep, config 5 27.0 -> 26.1 (any
    # split of the levels with B above A measured the same; switch for the A/B: SMCX_GEN_TT_PRIO=0)
    # (measured, config 2 / config 5 ms per sweep: none 1.562 / 27.11; all 1.403 / 25.91; team B until its reduction is
    # done 1.451 / 25.65; until its probe is done 1.467 / 25.67)
    # team B always + team A from its probe to the end of its reduction ("allA"): 1.43 / 25.40
    TTP = os.environ.get("SMCX_GEN_TT_PRIO", "all" if NS == 16 else "allA")     # all | allA | allA2 | allA1 | probe | red | 0
    TTPA = {"allA": 3, "allA2": 2, "allA1": 1}.get(TTP)       # team A's level from its probe to the end of its reduction
    if TTPA:
        TTP = "allA"
    if TTP in ("all", "allA"):
        E(f"""
        s_cmp_ge_u32 {WAVE}, {KS}
        s_cbranch_scc0 L_prioA
        s_setprio 3
        L_prioA:
        """)
if TT:
    # z8t: per-lane constants of the part that follows the exchange, in the registers behind the cells: row r = lane >> 4
    # carries component r - 1 (row 0: the energies); inside a row, lanes 0-7 work for the outcome "rejected", lanes 8-15
    # for "accepted" (the side pair's two results), so the next proposal is formed for BOTH before the decision is known
    KR = XY0 + NS
    KINV, KFIX, KPROW, KSIDE = f"v[{KR}:{KR+1}]", f"v[{KR+2}:{KR+3}]", f"v{KR+4}", f"v{KR+5}"
    KRZ = f"v{KR+6}"          # the reach of a group's z range (an accepted move widens its group's range by it)
    KL4T = f"v{KR+7}"         # TL: 4 lane + this wave's LDS block (the hand-over list is read from there)
    # TL: the addresses that alternate with the parity of the move counter (exchange buffer: where this wave writes its partial
    # sums, where every wave reads them; the side result's row of this move's buffer) are three registers toggled by one v_xor
    # each where the counter is incremented, instead of being formed from the counter in every move (v{KR+8}..v{KR+17}: the
    # wall lanes' rows, below)
    VXW, VXR, VSR = f"v{KR+18}", f"v{KR+19}", f"v{KR+20}"
    if TL:
        # per-wave constants of the special lanes: wall sites and plane on lanes 0 .. M2w (the plane last; none off the slab-0
        # waves, where M2w = -1), the side pair on the two lanes behind them (team B's slab-0 wave)
        E(f"""
        v_lshl_add_u32 {KL4T}, {LANE}, 2, v1
        s_and_b32 {st(0)}, {WAVE}, {KS - 1}
        s_lshl_b32 {s('wbase')}, {st(0)}, {WSH}
        s_add_u32 {st(0)}, {s('M2w')}, 1
        s_bfm_b64 {sp('wallM')}, {st(0)}, 0
        s_mov_b64 {sp('wallB')}, {sp('wallM')}
        s_mov_b64 {sp('planeM')}, 0
        s_cmp_lt_i32 {s('M2w')}, 0
        s_cbranch_scc1 L_nopw
        s_lshl_b64 {sp('planeM')}, 1, {s('M2w')}
        L_nopw:
        s_mov_b64 {sp('planeB')}, {sp('planeM')}
        s_add_u32 {s('sideL')}, {s('M2w')}, 1
        s_add_u32 {s('sideN')}, {s('M2w')}, 2
        // the wall lanes' rows of the wall table, once per launch (tt_wall_rows restores them behind a further round)
        v_mov_b64 v[{KR+14}:{KR+15}], 1.0
        v_mov_b64 v[{KR+16}:{KR+17}], 1.0
        v_lshlrev_b32 v49, 5, {LANE}
        s_mov_b64 exec, {sp('wallM')}
        global_load_dwordx4 v[{KR+8}:{KR+11}], v49, {sp('wtab')}
        global_load_dwordx4 v[{KR+14}:{KR+17}], v49, {sp('wtab')} offset:16
        s_mov_b64 exec, -1
        s_waitcnt vmcnt(0)
        """)
    E(f"""
    s_load_dword {st(0)}, {KARG}, {K_RZ}
    s_waitcnt lgkmcnt(0)
    v_mov_b32 {KRZ}, {st(0)}
    v_lshrrev_b32 v14, 4, {LANE}
    v_add_u32 v15, -1, v14
    v_max_i32 v15, 0, v15
    v_lshlrev_b32 {KPROW}, 3, v15
    v_bfe_u32 v15, {LANE}, 3, 1
    v_lshlrev_b32 v15, 5, v15
    v_lshl_add_u32 {KSIDE}, v14, 3, v15
    v_mov_b32 v{KR}, {s('invL')}
    v_mov_b32 v{KR+1}, {s('invL',1)}
    v_mov_b32 v{KR+2}, {s('toFix')}
    v_mov_b32 v{KR+3}, {s('toFix',1)}
    s_mov_b32 exec_lo, 0x0000ffff
    s_mov_b32 exec_hi, 0xffff0000
    v_mov_b32 v{KR}, 0
    v_mov_b32 v{KR+1}, 0
    v_mov_b32 v{KR+2}, {s('zFix')}
    v_mov_b32 v{KR+3}, {s('zFix',1)}
    s_mov_b64 exec, -1
    """)


def mark(k, iv=None):
    """(one-wave kernels with SMCX_GEN_IVAL: point `iv`, or k, of the interval measurement described at the switch.)
    z8t stamps variant: the cycles since this wave's previous mark are added to its LDS word k (1..15); word 0 = the
    time of the previous mark.  Uses st(4..6), v46, v47 and drains the LDS/scalar counter: only where those are dead and
    exec is full.  A mark costs about 100 cycles itself, charged to the interval that follows it."""
    if IVAL and not TT:
        k = k if iv is None else iv
        IVN[0] += 1
        if IVAL == "list":
            E(f"L_ivp{k}_{IVN[0]}:")
        if IVC is not None and k == IVC:
            E("s_add_u32 m0, m0, 1")
        for point, op in ((IVA, "s_sub_u32"), (IVB, "s_add_u32")):
            if k == point:
                E(f"""
                s_memtime s[100:101]
                s_waitcnt lgkmcnt(0)
                {op} m0, m0, s100
                """)
        return
    if k is None or not (TT and STAMPS):
        return
    E(f"""
    s_memtime {stp(4)}
    s_lshl_b32 {st(6)}, {WAVE}, 6
    v_mov_b32 v47, {st(6)}
    s_waitcnt lgkmcnt(0)
    s_mov_b64 exec, 1
    ds_read_b32 v46, v47 offset:{LDS_TM}
    s_waitcnt lgkmcnt(0)
    v_sub_u32 v46, {st(4)}, v46
    {f"ds_add_u32 v47, v46 offset:{LDS_TM + 4 * k}" if k else ""}
    v_mov_b32 v46, {st(4)}
    ds_write_b32 v47, v46 offset:{LDS_TM}
    s_mov_b64 exec, -1
    """)


def cnt_addr(vreg):
    """z8c / z8wc: vreg <- LDS address (before the offset LDS_CNT) of this wave's diagnostic counters:
    +0 pairs inside the cutoff, +4 candidate bits, +8 pairs inside whose bit is missing, +12 groups screened, +16 passes"""
    if W4:
        E(f"v_mov_b32 {vreg}, {WAVE}")
        E(f"v_lshlrev_b32 {vreg}, 5, {vreg}")
    else:
        E(f"v_mov_b32 {vreg}, 0")


if ZB:
    # LDS gb[g] = (max int, min int); lane g will read its group's pair when the copies are built
    E(f"""
    {f"v_lshl_add_u32 v14, {LANE}, 3, v1" if W4 else f"v_lshlrev_b32 v14, 3, {LANE}"}
    v_mov_b32 v16, 0x7fffffff
    v_mov_b32 v17, 0x80000000
    ds_write_b64 v14, v[16:17] offset:{LDS_GB}
    {f"v_mov_b32 v16, 0" if ZBC else ""}
    {f"v_mov_b32 v17, 0" if ZBC else ""}
    {f"ds_write_b64 v14, v[16:17] offset:{LDS_CNT}" if ZBC else ""}
    s_waitcnt lgkmcnt(0)
    """)
    if Z8C:   # lanes 0..7 zero this wave's eight counter words
        cnt_addr("v14")
        E(f"""
        v_lshl_add_u32 v14, {LANE}, 2, v14
        v_mov_b32 v16, 0
        s_mov_b64 exec, 0xff
        ds_write_b32 v14, v16 offset:{LDS_CNT}
        s_mov_b64 exec, -1
        s_waitcnt lgkmcnt(0)
        """)

zb_range_update = "" if not ZB else f"""
s_lshr_b32 {st(6)}, {st(0)}, {GSH}
s_lshl_b32 {st(6)}, {st(6)}, 3
{f"v_add_u32 v26, {st(6)}, v1" if W4 else f"v_mov_b32 v26, {st(6)}"}
s_mov_b64 exec, vcc
ds_min_i32 v26, v24 offset:{LDS_GB}
ds_max_i32 v26, v24 offset:{LDS_GB + 4}
s_mov_b64 exec, -1
"""

if LP:   # the wall table rows (site x, y, two coefficients; the plane last) into LDS: lanes 0 .. M2
    E(f"""
    s_cmp_lt_i32 {s('M2')}, 0
    s_cbranch_scc1 L_nowt
    s_add_u32 {st(0)}, {s('M2')}, 1
    s_lshl_b64 {stp(2)}, 1, {st(0)}
    s_sub_u32 {st(2)}, {st(2)}, 1
    s_subb_u32 {st(3)}, {st(3)}, 0
    s_mov_b64 exec, {stp(2)}
    v_lshlrev_b32 v14, 5, {LANE}
    global_load_dwordx4 v[16:19], v14, {sp('wtab')}
    global_load_dwordx4 v[20:23], v14, {sp('wtab')} offset:16
    s_waitcnt vmcnt(0)
    ds_write_b64 v14, v[16:17] offset:{LDS_WT}
    ds_write_b64 v14, v[18:19] offset:{LDS_WT + 8}
    ds_write_b64 v14, v[20:21] offset:{LDS_WT + 16}
    ds_write_b64 v14, v[22:23] offset:{LDS_WT + 24}
    s_mov_b64 exec, -1
    L_nowt:
    """)
# ---- compact copies: 64 passes, each loads logical slot k of this lane, packs it, and shifts it in at the
# top of the register file (xy[j] <- xy[j+1], xy[63] <- new): after 64 passes slot k sits in xy[k]
E(f"""
s_mov_b32 {st(0)}, 0
L_init:
v_lshl_or_b32 v14, {st(0)}, 6, {LANE}
v_cmp_gt_u32 vcc, {s_Nw()}, v14
v_mul_u32_u24 v15, 24, v14
v_mov_b32 v16, 0
v_mov_b32 v17, 0
v_mov_b32 v18, 0
v_mov_b32 v19, 0
v_mov_b32 v20, 0
v_mov_b32 v21, 0
s_and_saveexec_b64 {stp(2)}, vcc
global_load_dwordx4 v[16:19], v15, {SRC}
global_load_dwordx2 v[20:21], v15, {SRC} offset:16
s_waitcnt vmcnt(0)
{f"ds_write_b64 v15, v[16:17] offset:{LDS_RS}" if LP else ""}
{f"ds_write_b64 v15, v[18:19] offset:{LDS_RS + 8}" if LP else ""}
{f"ds_write_b64 v15, v[20:21] offset:{LDS_RS + 16}" if LP else ""}
s_mov_b64 exec, -1
v_mul_f64 v[22:23], v[16:17], {sp('toFix')}
v_mul_f64 v[24:25], v[18:19], {sp('toFix')}
v_rndne_f64 v[22:23], v[22:23]
v_rndne_f64 v[24:25], v[24:25]
v_cvt_i32_f64 v22, v[22:23]
v_cvt_i32_f64 v24, v[24:25]
{"v_and_b32 v22, 0xff, v22" if Z8 else "v_and_b32 v22, 0xffff, v22"}
{"v_and_b32 v24, 0xff, v24" if Z8 else ""}
{"v_lshl_or_b32 v22, v24, 8, v22" if Z8 else "v_lshl_or_b32 v22, v24, 16, v22"}
// z -> int16 in units uz, clamped to +-32767; padding slots hold 0x7fff ; unsafe = real && !(|z| < zsafe)
v_mul_f64 v[24:25], v[20:21], {sp('zFix')}
v_rndne_f64 v[24:25], v[24:25]
v_cvt_i32_f64 v24, v[24:25]
v_mov_b32 v25, 0x7fff
v_mov_b32 v27, 0xffff8001
v_med3_i32 v24, v24, v25, v27
v_cmp_nlt_f64 {stp(4)}, |v[20:21]|, {sp('zsafe')}
s_and_b64 {stp(4)}, {stp(4)}, vcc
v_cndmask_b32 v22, 0, v22, vcc
v_cndmask_b32 v24, v25, v24, vcc
""")
if Z8:
    # word = z16 | x8 << 16 | y8 << 24
    E(f"""
    v_and_b32 v26, 0xffff, v24
    v_lshl_or_b32 v22, v22, 16, v26
    """)
else:
    E(f"""
    s_lshr_b32 {st(6)}, {st(0)}, 1
    s_lshl_b32 {st(6)}, {st(6)}, 8
    s_and_b32 {st(7)}, {st(0)}, 1
    s_lshl_b32 {st(7)}, {st(7)}, 1
    s_add_u32 {st(6)}, {st(6)}, {st(7)}
    v_add_u32 v26, {st(6)}, {v('zaddr')}
    ds_write_b16 v26, v24
    """)
E(f"""
{zb_range_update}
v_lshrrev_b64 v[{V['uns0']}:{V['uns1']}], 1, v[{V['uns0']}:{V['uns1']}]
v_cndmask_b32 v27, 0, 1, {stp(4)}
v_lshl_or_b32 {v('uns1') if NS == 64 else v('uns0')}, v27, {(NS - 1) % 32}, {v('uns1') if NS == 64 else v('uns0')}
""")
for k in range(NS - 1):
    E(f"v_mov_b32 {xy(k)}, {xy(k+1)}")
E(f"""
v_mov_b32 {xy(NS - 1)}, v22
s_add_u32 {st(0)}, {st(0)}, 1
s_cmp_lt_u32 {st(0)}, {NS}
s_cbranch_scc1 L_init
s_waitcnt lgkmcnt(0)
""")
if T5:   # some cell of this replica beyond the safe z range?
    E(f"""
    v_or_b32 v14, {v('uns0')}, {v('uns1')}
    v_cmp_ne_u32 vcc, 0, v14
    s_nop 1
    s_cmp_lg_u64 vcc, 0
    s_cselect_b32 {ANYU}, 1, 0
    """)
if ZB:
    # lane g < NG: (lowest z - RZ, highest z + RZ) of group g; an empty group and the other lanes: never reached
    E(f"""
    s_load_dword {st(0)}, {KARG}, {K_RZ}
    {f"v_lshl_add_u32 v14, {LANE}, 3, v1" if W4 else f"v_lshlrev_b32 v14, 3, {LANE}"}
    ds_read_b64 v[16:17], v14 offset:{LDS_GB}
    s_waitcnt lgkmcnt(0)
    {f"s_mov_b32 {s('RZ')}, {st(0)}" if (Z8 and not W4) else ""}
    v_subrev_u32 {v('gloR')}, {st(0)}, v16
    v_add_u32 {v('ghiR')}, {st(0)}, v17
    v_cmp_gt_i32 vcc, v16, v17
    v_cmp_le_u32 {stp(2)}, {NG}, {LANE}
    s_or_b64 vcc, vcc, {stp(2)}
    v_mov_b32 v16, 0x7fffffff
    v_mov_b32 v17, 0x80000000
    v_cndmask_b32 {v('gloR')}, {v('gloR')}, v16, vcc
    v_cndmask_b32 {v('ghiR')}, {v('ghiR')}, v17, vcc
    """)


def compact_row(xyd, zld, locv):
    """zb: compact copy of the fp64 position in v[16:21] -> xyd = packed x,y ; zld = z16 | unsafe << 16 | cell << 17
    (cell = locv, loaded from loc[]).  Same conversions as the register copies above."""
    E(f"""
    v_mul_f64 v[26:27], v[16:17], {sp('toFix')}
    v_mul_f64 v[28:29], v[18:19], {sp('toFix')}
    v_rndne_f64 v[26:27], v[26:27]
    v_rndne_f64 v[28:29], v[28:29]
    v_cvt_i32_f64 v26, v[26:27]
    v_cvt_i32_f64 v28, v[28:29]
    {"v_and_b32 v26, 0xff, v26" if Z8 else "v_and_b32 v26, 0xffff, v26"}
    {"v_and_b32 v28, 0xff, v28" if Z8 else ""}
    {f"v_lshl_or_b32 {xyd}, v28, 8, v26" if Z8 else f"v_lshl_or_b32 {xyd}, v28, 16, v26"}
    v_mul_f64 v[28:29], v[20:21], {sp('zFix')}
    v_rndne_f64 v[28:29], v[28:29]
    v_cvt_i32_f64 v28, v[28:29]
    v_mov_b32 v29, 0x7fff
    v_mov_b32 v27, 0xffff8001
    v_med3_i32 v28, v28, v29, v27
    v_and_b32 v28, 0xffff, v28
    v_cmp_nlt_f64 vcc, |v[20:21]|, {sp('zsafe')}
    v_cndmask_b32 v29, 0, 1, vcc
    {f"v_lshl_or_b32 {xyd}, {xyd}, 16, v28" if Z8 else ""}
    v_lshl_or_b32 v28, v29, 16, v28
    v_lshl_or_b32 {zld}, {locv}, 17, v28
    """)
    if MG:   # bit 31: within the cutoff of a wall (or beyond it)
        E(f"""
        v_cmp_nlt_f64 vcc, |v[20:21]|, {sp('znear')}
        v_cndmask_b32 v29, 0, 1, vcc
        v_lshl_or_b32 {zld}, v29, 31, {zld}
        """)


def fill_p0(tag):
    """p0[lane] = fp64 position of particle rot*64 + lane (this lane's slot-0 particle), p0[64] = particle
    rot*64 + 64 (lane 0's slot-1 particle); zb: also the compact copies and cells of the NEXT particles: lane l of rxy, rzl <-
    particle rot*64 + l + 1 (probe B of the move of particle rot*64 + l is one v_readlane at lane tl, also when the order
    crosses into the next row), and where a run starts (tag r1) nxy, nzl <- those of the run's first particle"""
    E(f"""
    s_lshl_b32 {st(0)}, {s('rot')}, 6
    v_or_b32 v14, {st(0)}, {LANE}
    v_cmp_gt_u32 vcc, {s('N')}, v14
    v_mul_u32_u24 v15, 24, v14
    {f"v_mad_u32_u24 v22, {LANE}, 24, v1" if W4 else f"v_mul_u32_u24 v22, 24, {LANE}"}
    """)
    if ZB:
        E(f"""
        s_load_dwordx2 {stp(4)}, {KARG}, {K_LOC}
        s_mul_i32 {st(6)}, {REP}, {s('N')}
        s_lshl_b32 {st(6)}, {st(6)}, 1
        v_lshlrev_b32 v23, 1, v14
        s_waitcnt lgkmcnt(0)
        s_add_u32 {st(4)}, {st(4)}, {st(6)}
        s_addc_u32 {st(5)}, {st(5)}, 0
        """)
    E(f"""
    s_and_saveexec_b64 {stp(2)}, vcc
    global_load_dwordx4 v[16:19], v15, {sp('Rg')}{NT}
    global_load_dwordx2 v[20:21], v15, {sp('Rg')} offset:16{NT}
    s_waitcnt vmcnt(0)
    ds_write_b64 v22, v[16:17] offset:{LDS_P0}
    ds_write_b64 v22, v[18:19] offset:{LDS_P0 + 8}
    ds_write_b64 v22, v[20:21] offset:{LDS_P0 + 16}
    """)
    E(f"""
    s_mov_b64 exec, 1
    s_add_u32 {st(0)}, {st(0)}, 64
    s_cmp_lt_u32 {st(0)}, {s('N')}
    s_cbranch_scc0 L_p0done_{tag}
    s_mul_i32 {st(1)}, {st(0)}, 24
    v_mov_b32 v15, {st(1)}
    s_nop 1
    global_load_dwordx4 v[16:19], v15, {sp('Rg')}{NT}
    global_load_dwordx2 v[20:21], v15, {sp('Rg')} offset:16{NT}
    {f"v_add_u32 v25, {LDS_P0 + 64 * 24}, v1" if W4 else f"v_mov_b32 v25, {LDS_P0 + 64 * 24}"}
    s_waitcnt vmcnt(0)
    ds_write_b64 v25, v[16:17]
    ds_write_b64 v25, v[18:19] offset:8
    ds_write_b64 v25, v[20:21] offset:16
    L_p0done_{tag}:
    s_mov_b64 exec, -1
    s_waitcnt lgkmcnt(0)
    """)
    if ZB:
        # the row registers: lane l <- particle rot*64 + l + 1 (fp64 position = p0[l + 1], just written; cell from loc[])
        E(f"""
        v_add_u32 v14, 1, v14
        v_cmp_gt_u32 vcc, {s('N')}, v14
        s_and_saveexec_b64 {stp(2)}, vcc
        global_load_ushort v24, v23, {stp(4)} offset:2
        ds_read_b64 v[16:17], v22 offset:{LDS_P0 + 24}
        ds_read_b64 v[18:19], v22 offset:{LDS_P0 + 32}
        ds_read_b64 v[20:21], v22 offset:{LDS_P0 + 40}
        s_waitcnt vmcnt(0) lgkmcnt(0)
        """)
        compact_row(v('rxy'), v('rzl'), "v24")
        if tag == "r1":   # a run starts: its first particle is probe B of the pass that has no proposal to decide
            E(f"""
            s_mov_b64 exec, 1
            s_and_b32 {st(0)}, {s('first')}, 63
            s_mul_i32 {st(0)}, {st(0)}, 24
            {f"v_add_u32 v25, {st(0)}, v1" if W4 else f"v_mov_b32 v25, {st(0)}"}
            s_lshl_b32 {st(1)}, {s('first')}, 1
            v_mov_b32 v23, {st(1)}
            s_nop 0
            global_load_ushort v24, v23, {stp(4)}
            ds_read_b64 v[16:17], v25 offset:{LDS_P0}
            ds_read_b64 v[18:19], v25 offset:{LDS_P0 + 8}
            ds_read_b64 v[20:21], v25 offset:{LDS_P0 + 16}
            s_waitcnt vmcnt(0) lgkmcnt(0)
            """)
            compact_row("v30", "v31", "v24")
            E(f"""
            s_nop 0
            v_readfirstlane_b32 {s('nxy')}, v30
            v_readfirstlane_b32 {s('nzl')}, v31
            """)
        E("s_mov_b64 exec, -1")
    if PRIO:
        # The SIMD's arbiter favours the oldest of its four wavefronts: left alone, one replica runs at nearly the
        # lone-wave rate and finishes after 7 ms while the youngest needs 13 and runs the last third of its sweep
        # alone on the SIMD.  So at every row (64 moves) a wavefront publishes its progress in a table row shared
        # by the wavefronts of its SIMD (index from HW_ID / XCC_ID) and takes as its issue priority the number of
        # its neighbours that are AHEAD of it (four levels; two levels -- the two behind get 1 -- measured 2.5 % slower).
        # Only speed depends on this table, never a result.
        # Row index (16-byte rows, 4096 rows = the 64 KB table of smcx_create): HW_ID[14:8] (CU_ID 4 bits, SH_ID 1, SE_ID 2)
        # << 4 | HW_ID[5:4] (SIMD_ID) << 2 ... in words: bits 4-10, 2-3, and XCC_ID (3 bits, MI355X has 8 XCDs) in bits
        # 11-13; the slot inside the row is HW_ID[1:0] (WAVE_ID & 3).  Every field is read with its exact width, so the
        # store below stays inside the table whatever the registers hold; wavefronts whose WAVE_IDs alias modulo 4 share
        # a slot and only blunt the heuristic.
        E(f"""
        s_load_dwordx2 {stp(4)}, {KARG}, {K_PRIO}
        s_getreg_b32 {st(0)}, hwreg(HW_REG_HW_ID)
        s_getreg_b32 {st(1)}, hwreg(HW_REG_XCC_ID, 0, 3)
        s_bfe_u32 {st(2)}, {st(0)}, 0x70008
        s_lshl_b32 {st(2)}, {st(2)}, 4
        s_bfe_u32 {st(3)}, {st(0)}, 0x20004
        s_lshl_b32 {st(3)}, {st(3)}, 2
        s_or_b32 {st(2)}, {st(2)}, {st(3)}
        s_lshl_b32 {st(1)}, {st(1)}, 11
        s_or_b32 {st(2)}, {st(2)}, {st(1)}
        s_lshl_b32 {st(2)}, {st(2)}, 2
        s_and_b32 {st(3)}, {st(0)}, 3
        // key = (moves done: sweep index of the chunk x N + moves of this sweep) << 2 | slot
        s_lshl_b32 {st(6)}, {s('rot')}, 6
        s_sub_u32 {st(7)}, {s('N')}, {s('len')}
        s_sub_u32 {st(0)}, 0, {s('first')}
        s_cmp_eq_u32 {s('run')}, 0
        s_cselect_b32 {st(7)}, {st(0)}, {st(7)}
        s_add_u32 {st(6)}, {st(6)}, {st(7)}
        s_mul_i32 {st(7)}, {s('sw')}, {s('N')}
        s_add_u32 {st(6)}, {st(6)}, {st(7)}
        s_lshl_b32 {st(6)}, {st(6)}, 2
        s_or_b32 {st(6)}, {st(6)}, {st(3)}
        s_lshl_b32 {st(3)}, {st(3)}, 2
        s_add_u32 {st(3)}, {st(3)}, {st(2)}
        v_mov_b32 v14, {st(2)}
        v_mov_b32 v15, {st(6)}
        v_mov_b32 v16, {st(3)}
        s_mov_b64 exec, 1
        s_waitcnt lgkmcnt(0)
        global_store_dword v16, v15, {stp(4)} sc0 sc1
        global_load_dwordx4 v[20:23], v14, {stp(4)} sc0 sc1
        s_waitcnt vmcnt(0)
        v_readfirstlane_b32 {st(0)}, v20
        v_readfirstlane_b32 {st(1)}, v21
        v_readfirstlane_b32 {st(2)}, v22
        v_readfirstlane_b32 {st(3)}, v23
        s_mov_b64 exec, -1
        s_mov_b32 {st(7)}, 0
        s_cmp_gt_u32 {st(0)}, {st(6)}
        s_addc_u32 {st(7)}, {st(7)}, 0
        s_cmp_gt_u32 {st(1)}, {st(6)}
        s_addc_u32 {st(7)}, {st(7)}, 0
        s_cmp_gt_u32 {st(2)}, {st(6)}
        s_addc_u32 {st(7)}, {st(7)}, 0
        s_cmp_gt_u32 {st(3)}, {st(6)}
        s_addc_u32 {st(7)}, {st(7)}, 0
        s_min_u32 {st(7)}, {st(7)}, 3
        s_cmp_eq_u32 {st(7)}, 0
        s_cbranch_scc1 L_pr0_{tag}
        s_cmp_eq_u32 {st(7)}, 1
        s_cbranch_scc1 L_pr1_{tag}
        s_cmp_eq_u32 {st(7)}, 2
        s_cbranch_scc1 L_pr2_{tag}
        s_setprio 3
        s_branch L_pre_{tag}
        L_pr0_{tag}:
        s_setprio 0
        s_branch L_pre_{tag}
        L_pr1_{tag}:
        s_setprio 1
        s_branch L_pre_{tag}
        L_pr2_{tag}:
        s_setprio 2
        L_pre_{tag}:
        """)
    fill_p0_end()


def mg_coeff_init(emit):
    """the merged pass's coefficient registers: 1 for candidates and side lanes, (a0, b0) on the plane lanes (the words of wlp).
    The steady copy of the move keeps them from move to move: they are set where wlp changes (end of a generic pass), after
    a row fill (which uses the registers) and behind the cold pieces that write them (near pass, further rounds)"""
    emit(f"""
    v_mov_b64 v[{V["C"]}:{V["C"]+1}], 1.0
    v_mov_b64 v[{V["C"]+2}:{V["C"]+3}], 1.0
    s_mov_b64 exec, {sp('wlp')}
    v_mov_b64 v[{V["C"]}:{V["C"]+1}], {sp('a0s')}
    v_mov_b64 v[{V["C"]+2}:{V["C"]+3}], {sp('b0s')}
    s_mov_b64 exec, -1
    """)


def fill_p0_end():
    if MG:   # the row fill used the registers of the merged pass's vector constants
        E(f"""
        v_mov_b32 v22, {s('wlp')}
        v_mov_b32 v23, {s('stB')}
        """)
        mg_coeff_init(E)
        if W4:
            E(f"""
            v_bfe_u32 v14, {LANE}, 3, 1
            v_bfe_u32 v15, {LANE}, 4, 1
            v_lshlrev_b32 v16, 3, v14
            v_lshlrev_b32 {KC}, v15, v16
            """)


def rotate(tag):
    """slot j <- slot j+1 for the packed x,y, the int16 z in LDS and the unsafe bits; then the p0 cache
    (zb: the cells never move; only the row of current particles advances)"""
    if ZB:
        E(f"s_add_u32 {s('rot')}, {s('rot')}, 1")
        fill_p0(tag)
        return
    E(f"v_mov_b32 v14, {xy(0)}")
    for k in range(NS - 1):
        E(f"v_mov_b32 {xy(k)}, {xy(k+1)}")
    E(f"v_mov_b32 {xy(NS - 1)}, v14")
    # z: 32 words per lane; new word j = old[j].hi | old[j+1].lo << 16 ; the last one wraps to old[0]
    E(f"ds_read2st64_b32 v[14:15], {v('zaddr')} offset0:0 offset1:1")
    E("s_waitcnt lgkmcnt(0)")
    E("v_mov_b32 v24, v14")
    NWRD = NS // 2
    for j in range(0, NWRD, 2):
        if j + 2 < NWRD:
            E(f"ds_read2st64_b32 v[16:17], {v('zaddr')} offset0:{j+2} offset1:{j+3}")
            E("s_waitcnt lgkmcnt(0)")
            nxt = "v16"
        else:
            nxt = "v24"
        E("v_alignbit_b32 v18, v15, v14, 16")
        E(f"v_alignbit_b32 v19, {nxt}, v15, 16")
        E(f"ds_write2st64_b32 {v('zaddr')}, v18, v19 offset0:{j} offset1:{j+1}")
        if j + 2 < NWRD:
            E("v_mov_b32 v14, v16")
            E("v_mov_b32 v15, v17")
    E(f"""
    s_waitcnt lgkmcnt(0)
    v_and_b32 v14, 1, {v('uns0')}
    v_lshrrev_b64 v[{V['uns0']}:{V['uns1']}], 1, v[{V['uns0']}:{V['uns1']}]
    v_lshl_or_b32 {v('uns1') if NS == 64 else v('uns0')}, v14, {(NS - 1) % 32}, {v('uns1') if NS == 64 else v('uns0')}
    s_add_u32 {s('rot')}, {s('rot')}, 1
    s_and_b32 {s('rot')}, {s('rot')}, {NS - 1}
    """)
    fill_p0(tag)


fill_p0("init")

if Z8:
    # the nlu pair holds the address the screens' computed jumps are relative to: where block 31 of probe A's pass in the steady
    # copy would start if there were 32 (s_flbit counts from bit 31), so that pass adds nothing to flbit x block size
    # (L_jb is a label further down, its offset positive; the step back is taken apart from it: the sum can be negative, and
    # an unsigned add of a negative literal would carry into the high word)
    E(f"""
    s_getpc_b64 {sp('nlu')}
    L_jbase:
    s_add_u32 {s('nlu')}, {s('nlu')}, L_jb-L_jbase
    s_addc_u32 {s('nlu',1)}, {s('nlu',1)}, 0
    s_sub_u32 {s('nlu')}, {s('nlu')}, {32 - NG}*(L_sg{NG-2}_A-L_sg{NG-1}_A)
    s_subb_u32 {s('nlu',1)}, {s('nlu',1)}, 0
    """)

# ---------------------------------------------------------------------------------------------- sweeps, runs
if not ZB:
    E(f"""
    s_mov_b32 {s('sw')}, 0
    L_sweep:
    // displ + ((cbase + sw) * 3N) * 8 ; uni + ((cbase + sw) * N) * 8 ; n0 = offs[cbase + sw]
    s_load_dwordx4 {st4(4)}, {KARG}, {K_DISPL}
    s_load_dwordx2 {stp(2)}, {KARG}, {K_OFFS}
    s_add_u32 {st(0)}, {s('cbase')}, {s('sw')}
    s_lshl_b32 {st(1)}, {st(0)}, 2
    s_waitcnt lgkmcnt(0)
    s_load_dword {s('n0')}, {stp(2)}, {st(1)}
    s_mul_i32 {st(2)}, {s('N')}, 8
    s_mul_hi_u32 {st(3)}, {st(2)}, {st(0)}
    s_mul_i32 {st(2)}, {st(2)}, {st(0)}
    s_add_u32 {s('uni')}, {st(6)}, {st(2)}
    s_addc_u32 {s('uni',1)}, {st(7)}, {st(3)}
    s_mul_i32 {st(6)}, {s('N')}, 24
    s_mul_hi_u32 {st(3)}, {st(6)}, {st(0)}
    s_mul_i32 {st(2)}, {st(6)}, {st(0)}
    s_add_u32 {s('displ')}, {st(4)}, {st(2)}
    s_addc_u32 {s('displ',1)}, {st(5)}, {st(3)}
    s_mov_b32 {s('jacc')}, 0
    s_mov_b32 {s('run')}, 0
    s_waitcnt lgkmcnt(0)
    """)
else:
    E(f"""
    L_sweep:
    s_mov_b32 {s('jacc')}, 0
    s_mov_b32 {s('run')}, 0
    """)
if not ZB:
    E(f"""
    L_run:
    s_sub_u32 {st(0)}, {s('N')}, {s('n0')}
    s_cmp_eq_u32 {s('run')}, 0
    s_cselect_b32 {s('first')}, {s('n0')}, 0
    s_cselect_b32 {s('len')}, {st(0)}, {s('n0')}
    s_cselect_b32 {s('vbase')}, 0, {st(0)}
    s_cmp_eq_u32 {s('len')}, 0
    s_cbranch_scc1 L_run_next
    L_rot_to:
    s_lshr_b32 {st(7)}, {s('first')}, 6
    s_cmp_eq_u32 {s('rot')}, {st(7)}
    s_cbranch_scc1 L_rot_ok
    """)
    rotate("r1")
    E(f"""
    s_branch L_rot_to
    L_rot_ok:
    s_and_b32 {s('tl')}, {s('first')}, 63
    s_sub_u32 {s('tl')}, {s('tl')}, 1
    // dK = displ + 24 first ; uK = uni + 8 vbase
    s_mul_i32 {st(0)}, {s('first')}, 24
    s_add_u32 {s('dK')}, {s('displ')}, {st(0)}
    s_addc_u32 {s('dK',1)}, {s('displ',1)}, 0
    s_lshl_b32 {st(0)}, {s('vbase')}, 3
    s_add_u32 {s('uK')}, {s('uni')}, {st(0)}
    s_addc_u32 {s('uK',1)}, {s('uni',1)}, 0
    """)
else:
    # a run starts: st(0) = cbase + sw, n0 = offs[cbase + sw], first / len of this run, then
    # dK = displ + ((cbase + sw) * N + first) * 24 ; uK = uni + ((cbase + sw) * N + vbase) * 8 ; then the row
    E("L_run:")
    cbase_to(st(0))
    E(f"""
    s_load_dwordx2 {stp(2)}, {KARG}, {K_OFFS}
    s_load_dwordx4 {st4(4)}, {KARG}, {K_DISPL}
    s_add_u32 {st(0)}, {st(0)}, {s('sw')}
    s_lshl_b32 {st(1)}, {st(0)}, 2
    s_waitcnt lgkmcnt(0)
    s_load_dword {st(1)}, {stp(2)}, {st(1)}
    s_waitcnt lgkmcnt(0)
    // st(1) = n0 ; st(2) = N - n0
    s_sub_u32 {st(2)}, {s('N')}, {st(1)}
    s_cmp_eq_u32 {s('run')}, 0
    s_cselect_b32 {s('first')}, {st(1)}, 0
    s_cselect_b32 {s('len')}, {st(2)}, {st(1)}
    s_cselect_b32 {st(1)}, 0, {st(2)}
    s_cmp_eq_u32 {s('len')}, 0
    s_cbranch_scc1 L_run_next
    // st(1) = vbase
    s_mul_i32 {st(2)}, {s('N')}, 8
    s_mul_hi_u32 {st(3)}, {st(2)}, {st(0)}
    s_mul_i32 {st(2)}, {st(2)}, {st(0)}
    s_add_u32 {s('uK')}, {st(6)}, {st(2)}
    s_addc_u32 {s('uK',1)}, {st(7)}, {st(3)}
    s_lshl_b32 {st(1)}, {st(1)}, 3
    s_add_u32 {s('uK')}, {s('uK')}, {st(1)}
    s_addc_u32 {s('uK',1)}, {s('uK',1)}, 0
    s_mul_i32 {st(6)}, {s('N')}, 24
    s_mul_hi_u32 {st(3)}, {st(6)}, {st(0)}
    s_mul_i32 {st(2)}, {st(6)}, {st(0)}
    s_add_u32 {s('dK')}, {st(4)}, {st(2)}
    s_addc_u32 {s('dK',1)}, {st(5)}, {st(3)}
    s_mul_i32 {st(1)}, {s('first')}, 24
    s_add_u32 {s('dK')}, {s('dK')}, {st(1)}
    s_addc_u32 {s('dK',1)}, {s('dK',1)}, 0
    s_lshr_b32 {s('rot')}, {s('first')}, 6
    """)
    fill_p0("r1")
    E(f"""
    s_and_b32 {s('tl')}, {s('first')}, 63
    s_sub_u32 {s('tl')}, {s('tl')}, 1
    """)
if TL:   # the addresses that alternate with the parity of i, for i = -1 (odd)
    E(f"""
    s_movk_i32 {st(0)}, {WPR * 512}
    v_lshl_add_u32 {VXR}, {LANE}, 3, {st(0)}
    s_lshl_b32 {st(0)}, {WAVE}, 9
    v_add_u32 {VXW}, {st(0)}, {VXR}
    v_add_u32 {VSR}, 64, {KSIDE}
    """)
E(f"""
s_mov_b32 {s('i')}, -1
s_mov_b32 {s('hasA')}, 0
{f"s_mov_b32 {s('hasAw')}, 0" if (W4 and not MG) else ""}
s_mov_b32 {s('hasB')}, 1
{"" if Z8 else f"s_mov_b32 {s('azz')}, 0"}
{"" if Z8 else f"s_mov_b32 {s('az16')}, 0"}
s_mov_b32 {s('ua')}, 0
{f"s_mov_b32 {s('axys')}, 0" if ZB else f"v_mov_b32 {v('axy')}, 0"}
v_mov_b32 {v('FmV')}, 0
v_mov_b32 {v('FmV',1)}, 0
v_mov_b32 {v('DdV')}, 0
v_mov_b32 {v('DdV',1)}, 0
""")
if MG:   # the first pass of a run has no proposal: no plane for half A, no side pair
    E(f"""
    s_mov_b32 {s('nearA')}, 0
    s_mov_b32 {s('wlp')}, 0
    s_cmp_ge_i32 {s('M2')}, 0
    s_cselect_b32 {s('wlp',1)}, 1, 0
    {f"s_cmp_eq_u32 {WAVE}, 0" if W4 else ""}
    {f"s_cselect_b32 {s('wlp',1)}, {s('wlp',1)}, 0" if W4 else ""}
    s_mov_b32 {s('stB')}, {s('wlp',1)}
    s_mov_b32 {s('sidesHi')}, 0
    v_mov_b32 v22, 0
    v_mov_b32 v23, {s('stB')}
    """)
# rarely executed pieces of the move (zb) are gathered here, jumped over when a run starts
COLD_AT = None
if ZB:
    E("s_branch L_move")
    COLD_AT = len(out)
MOVE_AT = len(out)
E("L_move:")
G("L_G_move:")
mark(0)
# measurement switch SMCX_GEN_PAD=<kind><count>: <count> extra instructions per move (S = SALU, V = fast VALU form, W = slow VALU
# form, N = s_nop 0, D = fp64 VALU), all on dead registers: what an instruction of each kind costs the running kernel
PAD = os.environ.get("SMCX_GEN_PAD", "")
if PAD:
    for _ in range(int(PAD[1:])):
        E({"S": f"s_add_u32 {st(7)}, {st(7)}, 1", "V": "v_mov_b32 v14, v15", "W": "v_alignbit_b32 v14, v15, v16, 31",
           "N": "s_nop 0", "D": "v_add_f64 v[14:15], v[16:17], v[18:19]"}[PAD[0]])
cold = []


def COLD(txt):
    if MUTE[0]:
        return
    for ln in txt.strip("\n").split("\n"):
        ln = ln.strip()
        if ln and not ln.startswith("//"):
            cold.append(ln)



# ---------------------------------------------------------------------------------------------- B's compact copy
if not ZB:
  E(f"""
s_mov_b32 {s('bzz')}, 0
s_mov_b32 {s('ub')}, 0
s_mov_b32 {s('cross')}, 0
s_mov_b32 {s('lb')}, 0
v_mov_b32 {v('bxy')}, 0
s_cmp_eq_u32 {s('hasB')}, 0
s_cbranch_scc1 L_nob1
ds_read_b32 v14, {v('zaddr')}
s_add_u32 {s('lb')}, {s('tl')}, 1
s_cmp_eq_u32 {s('tl')}, 63
s_cselect_b32 {s('lb')}, 0, {s('lb')}
s_cselect_b32 {s('cross')}, 1, 0
s_cbranch_scc1 L_bcross
v_readlane_b32 {st(0)}, {xy(0)}, {s('lb')}
v_readlane_b32 {st(2)}, {v('uns0')}, {s('lb')}
s_waitcnt lgkmcnt(0)
v_readlane_b32 {st(1)}, v14, {s('lb')}
s_and_b32 {s('ub')}, {st(2)}, 1
s_and_b32 {st(1)}, {st(1)}, 0xffff
s_branch L_bjoin
L_bcross:
v_readlane_b32 {st(0)}, {xy(1)}, 0
v_readlane_b32 {st(2)}, {v('uns0')}, 0
s_waitcnt lgkmcnt(0)
v_readlane_b32 {st(1)}, v14, 0
s_bfe_u32 {s('ub')}, {st(2)}, 0x10001
s_lshr_b32 {st(1)}, {st(1)}, 16
L_bjoin:
s_mul_i32 {s('bzz')}, {st(1)}, 0x10001
s_nop 0
v_mov_b32 {v('bxy')}, {st(0)}
L_nob1:
""")
else:
  # zb: from the row registers (lane tl + 1), or the scalar copy of particle row*64 + 64 when the order crosses rows
  G(f"""
  s_cmp_eq_u32 {s('hasB')}, 0
  s_cbranch_scc1 L_nob0
  """)
  # the first pass of a run (no proposal yet) takes the run's first particle from nxy, nzl
  G(f"""
  s_cmp_eq_u32 {s('hasA')}, 0
  s_cbranch_scc1 L_bfirst
  """)
  RW = s('ub') if Z8 else st(1)   # z8: the row word itself stays in ub (its bit 16 = particle n+1 is beyond the safe z range)
  COLD(f"""
  L_bfirst:
  s_mov_b32 {s('bxys')}, {s('nxy')}
  s_mov_b32 {RW}, {s('nzl')}
  s_branch L_bjoin
  """)
  E(f"""
  v_readlane_b32 {s('bxys')}, {v('rxy')}, {s('tl')}
  v_readlane_b32 {RW}, {v('rzl')}, {s('tl')}
  L_bjoin:
  {"" if Z8 else f"s_bfe_u32 {s('ub')}, {st(1)}, 0x10010"}
  {f"s_lshr_b32 {s('nearB')}, {RW}, 31" if MG else ""}
  {f"s_bfe_u32 {s('locB')}, {RW}, 0xe0011" if MG else f"s_lshr_b32 {s('locB')}, {RW}, 17"}
  {"" if Z8 else f"s_and_b32 {st(1)}, {st(1)}, 0xffff"}
  {"" if Z8 else f"s_mul_i32 {s('bzz')}, {st(1)}, 0x10001"}
  L_nob1:
  """)
  COLD(f"""
  L_nob0:
  {"" if Z8 else f"s_mov_b32 {s('bzz')}, 0"}
  s_mov_b32 {s('ub')}, 0
  {f"s_mov_b32 {s('nearB')}, 0" if MG else ""}
  s_mov_b32 {s('bxys')}, 0
  s_mov_b32 {s('locB')}, 0
  s_branch L_nob1
  """)

# ---------------------------------------------------------------------------------------------- screen


def screen_group(k0, zlo, zhi, pxy, pzz, w):
    """one probe against slots k0..k0+3 (z words zlo = slots (k0, k0+1), zhi = (k0+2, k0+3)), descending slot
    order: 22 instructions, four chains; every result is consumed at least four instructions after its issue
    (six for the v_dot2 results)"""
    a = ["v%d" % (V['t'] + j) for j in range(4)]       # slots k0+3, k0+2, k0+1, k0
    z1, z0 = "v%d" % (V['t'] + 4), "v%d" % (V['t'] + 5)
    X = [xy(k0 + 3), xy(k0 + 2), xy(k0 + 1), xy(k0)]
    for j in range(4):
        E(f"v_sub_u32 {a[j]}, {pxy}, {X[j]}")
    for j in range(4):
        E(f"v_dot2_i32_i16 {a[j]}, {a[j]}, {a[j]}, {s('negC')}")
    E(f"v_pk_sub_i16 {z1}, {pzz}, {zhi} clamp")
    E(f"v_pk_sub_i16 {z0}, {pzz}, {zlo} clamp")
    for j in range(4):
        E(f"v_ashrrev_i32 {a[j]}, {2 * ZS}, {a[j]}")
    zz = [z1, z1, z0, z0]
    hi = [True, False, True, False]
    for j in range(4):
        sel = " op_sel:[1,1,0,0]" if hi[j] else ""
        E(f"v_mad_i32_i16 {a[j]}, {zz[j]}, {zz[j]}, {a[j]}{sel}")
    for j in range(4):
        E(f"v_alignbit_b32 {w}, {w}, {a[j]}, 31")


def screen_pass(pxy, pzz, w0, w1):
    """candidate bits of one probe: 16 groups, the z words read two groups ahead (counted LDS waits: nothing
    else may be in flight on the LDS / scalar-memory counter while a pass runs)"""
    E(f"""
    v_mov_b32 {w0}, 0
    v_mov_b32 {w1}, 0
    ds_read2st64_b32 v[{V['zA']}:{V['zA']+1}], {v('zaddr')} offset0:{NS // 2 - 2} offset1:{NS // 2 - 1}
    ds_read2st64_b32 v[{V['zB']}:{V['zB']+1}], {v('zaddr')} offset0:{NS // 2 - 4} offset1:{NS // 2 - 3}
    """)
    groups = list(range(NS - 4, -1, -4))
    for gi, k0 in enumerate(groups):
        buf = 'zA' if gi % 2 == 0 else 'zB'
        E("s_waitcnt lgkmcnt(1)" if gi < len(groups) - 1 else "s_waitcnt lgkmcnt(0)")
        screen_group(k0, "v%d" % V[buf], "v%d" % (V[buf] + 1), pxy, pzz, w1 if k0 >= 32 else w0)
        if gi + 2 < len(groups):
            nk = groups[gi + 2]
            E(f"ds_read2st64_b32 v[{V[buf]}:{V[buf]+1}], {v('zaddr')} offset0:{nk // 2} offset1:{nk // 2 + 1}")


def screen_ranged(tag, pxys, pzz, w0, w1):
    """zb: candidate bits of one probe from the groups whose z range can hold a slot within RZ of the probe's z.
    Lane g of gloR/ghiR holds group g's range; the groups between the lowest and the highest hit run in
    descending order (entered by a computed jump), each fetching the z words of the next one; the bits land
    at the bottom of w0 / w1 and are shifted to their slots at the end."""
    w = lambda g: w1 if 4 * g >= 32 else w0
    buf = lambda g: 'zA' if g % 2 == 0 else 'zB'
    E(f"""
    v_mov_b32 {w0}, 0
    v_mov_b32 {w1}, 0
    s_sext_i32_i16 {st(0)}, {pzz}
    v_mov_b32 {v('pxy')}, {pxys}
    v_cmp_ge_i32 vcc, {st(0)}, {v('gloR')}
    v_cmp_le_i32 {stp(2)}, {st(0)}, {v('ghiR')}
    s_and_b32 {st(1)}, vcc_lo, {st(2)}
    s_cmp_eq_u32 {st(1)}, 0
    s_cbranch_scc1 L_sdone_{tag}
    s_ff1_i32_b32 {st(4)}, {st(1)}
    s_flbit_i32_b32 {st(5)}, {st(1)}
    s_sub_u32 {st(5)}, 31, {st(5)}
    s_getpc_b64 {stp(2)}
    L_spc_{tag}:
    s_mul_i32 {st(5)}, {st(5)}, L_se1_{tag}-L_se0_{tag}
    s_add_u32 {st(2)}, {st(2)}, {st(5)}
    s_addc_u32 {st(3)}, {st(3)}, 0
    s_add_u32 {st(2)}, {st(2)}, L_se0_{tag}-L_spc_{tag}
    s_addc_u32 {st(3)}, {st(3)}, 0
    s_setpc_b64 {stp(2)}
    """)
    for g in range(NG):
        E(f"L_se{g}_{tag}:")
        E(f"ds_read2st64_b32 v[{V[buf(g)]}:{V[buf(g)]+1}], {v('zaddr')} offset0:{2 * g} offset1:{2 * g + 1}")
        E(f"s_branch L_sg{g}_{tag}")
    if NG == 1:
        E(f"L_se1_{tag}:")
    for g in range(NG - 1, -1, -1):
        E(f"L_sg{g}_{tag}:")
        if g > 0:
            E(f"ds_read2st64_b32 v[{V[buf(g-1)]}:{V[buf(g-1)]+1}], {v('zaddr')} offset0:{2 * g - 2} offset1:{2 * g - 1}")
            E("s_waitcnt lgkmcnt(1)")
        else:
            E("s_waitcnt lgkmcnt(0)")
        screen_group(4 * g, "v%d" % V[buf(g)], "v%d" % (V[buf(g)] + 1), v('pxy'), pzz, w(g))
        if g > 0:
            E(f"s_cmp_eq_u32 {st(4)}, {g}")
            E(f"s_cbranch_scc1 L_sfin_{tag}")
    E(f"""
    L_sfin_{tag}:
    s_lshl_b32 {st(0)}, {st(4)}, 2
    s_waitcnt lgkmcnt(0)
    v_lshlrev_b32 {w0}, {st(0)}, {w0}
    """)
    if NS == 64:
        E(f"""
        s_sub_u32 {st(1)}, {st(0)}, 32
        s_max_i32 {st(1)}, {st(1)}, 0
        v_lshlrev_b32 {w1}, {st(1)}, {w1}
        """)
    E(f"L_sdone_{tag}:")


def zbc_check(pzz, w0, w1):
    """diagnostic: the full pass of the same probe; count its bits and those the ranged pass did not set"""
    screen_pass(v('pxy'), pzz, "v21", "v22")
    E(f"""
    v_not_b32 v23, {w0}
    v_not_b32 v24, {w1}
    v_and_b32 v23, v21, v23
    v_and_b32 v24, v22, v24
    v_bcnt_u32_b32 v23, v23, 0
    v_bcnt_u32_b32 v23, v24, v23
    v_bcnt_u32_b32 v24, v21, 0
    v_bcnt_u32_b32 v24, v22, v24
    v_lshlrev_b32 v25, 3, {LANE}
    ds_add_u32 v25, v24 offset:{LDS_CNT}
    ds_add_u32 v25, v23 offset:{LDS_CNT + 4}
    s_waitcnt lgkmcnt(0)
    """)


def screen_group8(k0, p, w, fill="", tb=None):
    """z8: one probe against slots k0..k0+GS-1, highest first: 3 instructions per slot; a v_dot4 result is read at least
    three instructions later (`fill`: what the caller puts between the dot products and the flag shifts of a 2-slot group)"""
    a = ["v%d" % ((V['t'] if tb is None else tb) + j) for j in range(GS)]
    X = [xy(k0 + GS - 1 - j) for j in range(GS)]
    for j in range(GS):
        E(f"v_sub_u32 {a[j]}, {p}, {X[j]}")
    for j in range(GS):
        E(f"v_dot4_i32_i8 {a[j]}, {a[j]}, {a[j]}, {s('negC')}")
    if GS < 4:
        E(fill if fill else "s_nop 1")
    for j in range(GS):
        E(f"v_alignbit_b32 {w}, {w}, {a[j]}, 31")


def screen_ranged8(tag, pws, w0, w1, R=None):
    """z8: as screen_ranged, without z words to fetch: the computed jump lands on the highest group in reach.
    R (the pre-screen of the next move's probe B, PS): its own registers -- dict(t0, t1, t4, t5: s; p2: (pair, lo, hi); pxy: v; tb:
    first of the GS temporaries) in place of st(0), st(1), st(4), st(5), stp(2), v('pxy'), V['t']"""
    w = lambda g: w1 if GS * g >= 32 else w0
    xt, xtp = st, stp
    if R is not None:
        xt = lambda i: {0: R['t0'], 1: R['t1'], 2: R['p2'][1], 3: R['p2'][2], 4: R['t4'], 5: R['t5']}[i]
        xtp = lambda i: {2: R['p2'][0]}[i]
    pxy = v('pxy') if R is None else R['pxy']
    tb = None if R is None else R['tb']
    ca, cv = ("v16", "v14") if R is None else ("v42", "v41")      # z8c: address and value of the counters' LDS adds
    E(f"""
    v_mov_b32 {w0}, 0
    v_mov_b32 {w1}, 0
    s_sext_i32_i16 {xt(0)}, {pws}
    v_mov_b32 {pxy}, {pws}
    v_cmp_ge_i32 vcc, {xt(0)}, {v('gloR')}
    v_cmp_le_i32 {xtp(2)}, {xt(0)}, {v('ghiR')}
    s_and_b32 {xt(1)}, vcc_lo, {xt(2)}
    {"s_branch" if ("noscreen" + tag) in ABL else "s_cbranch_scc0"} L_sdone_{tag}
    s_ff1_i32_b32 {xt(4)}, {xt(1)}
    s_flbit_i32_b32 {xt(5)}, {xt(1)}
    {f"s_add_u32 {xt(1)}, {xt(4)}, {xt(5)}" if Z8C else ""}
    {f"s_sub_u32 {xt(1)}, 32, {xt(1)}" if Z8C else ""}
    s_mul_i32 {xt(5)}, {xt(5)}, L_sg{NG-2}_{tag}-L_sg{NG-1}_{tag}
    """)
    # (probe A's pass of the steady copy starts at L_jb: nothing to add)
    (G if (tag == "A" and PEEL) else E)(f"s_add_u32 {xt(5)}, {xt(5)}, L_sg{NG-1}_{tag}-L_jb" if (tag != "A" or PEEL) else "")
    E(f"""
    s_add_u32 {xt(2)}, {s('nlu')}, {xt(5)}
    s_addc_u32 {xt(3)}, {s('nlu',1)}, 0
    s_setpc_b64 {xtp(2)}
    """)
    TWO = NS == 64 and GS == 4 and not Z8C    # two exits: a pass that ends in the upper word shifts that one only
    # blocks in descending group order; flbit = 31 - highest group, so block g sits (flbit - (32 - NG)) blocks in
    for g in range(NG - 1, -1, -1):
        if g == NG - 1 and tag == "A":
            E("L_jb:", " @S" if PEEL else "")   # what the computed jumps of BOTH copies of the move are relative to (the
                                                 # steady copy lies first: every offset is positive)
        E(f"L_sg{g}_{tag}:")
        # (2-slot groups: the exit test and one s_nop stand between the dot products and the shifts that read them)
        screen_group8(GS * g, pxy, w(g), f"s_cmp_eq_u32 {xt(4)}, {g}\ns_nop 0" if (g > 0 and GS < 4) else "", tb)
        if g > 0:
            if GS == 4:
                E(f"s_cmp_eq_u32 {xt(4)}, {g}")
            E(f"s_cbranch_scc1 L_sfin{('H' if GS * g >= 32 else 'L') if TWO else ''}_{tag}")
    if TWO:
        # the bits of the groups in reach lie at the bottom of their word: the word that holds the lowest group is shifted to
        # its slots (4 x lowest group, mod 32: v_lshlrev reads five bits); a pass that crossed into the lower word left the
        # upper one's bits where they belong
        E(f"""
        L_sfinL_{tag}:
        s_lshl_b32 {xt(0)}, {xt(4)}, {GSH}
        v_lshlrev_b32 {w0}, {xt(0)}, {w0}
        s_branch L_sdone_{tag}
        L_sfinH_{tag}:
        s_lshl_b32 {xt(0)}, {xt(4)}, {GSH}
        v_lshlrev_b32 {w1}, {xt(0)}, {w1}
        L_sdone_{tag}:
        """)
        return
    E(f"L_sfin_{tag}:")
    if Z8C:   # executed work: groups of this pass = highest - lowest + 1 = 32 - flbit - ff1 (in st(1) since the jump)
        cnt_addr(ca)
        E(f"""
        v_mov_b32 {cv}, {xt(1)}
        s_mov_b64 exec, 1
        ds_add_u32 {ca}, {cv} offset:{LDS_CNT + 12}
        s_mov_b64 exec, -1
        """)
    E(f"""
    s_lshl_b32 {xt(0)}, {xt(4)}, {GSH}
    v_lshlrev_b32 {w0}, {xt(0)}, {w0}
    """)
    if NS == 64:
        E(f"""
        s_sub_u32 {xt(1)}, {xt(0)}, 32
        s_max_i32 {xt(1)}, {xt(1)}, 0
        v_lshlrev_b32 {w1}, {xt(1)}, {w1}
        """)
    E(f"L_sdone_{tag}:")
    if Z8C:   # every pass, also one that found no group in reach
        cnt_addr(ca)
        E(f"""
        v_mov_b32 {cv}, 1
        s_mov_b64 exec, 1
        ds_add_u32 {ca}, {cv} offset:{LDS_CNT + 16}
        s_mov_b64 exec, -1
        """)


def z8c_check(tag, P_sgpr, w0, w1, locs, guard):
    """diagnostic (z8c): the fp64 cutoff test (minimum image in x, y; SMC.c:567-578) of the probe against EVERY cell,
    from the positions in memory.  Counts, in LDS words of lane 0: pairs inside the cutoff, candidate bits, and
    pairs inside the cutoff whose bit is not set (the cells in `locs` are excluded by construction).
    Runs after the exclusions and before the first candidates are taken out of w."""
    E(f"""
    s_cmp_eq_u32 {guard}, 0
    s_cbranch_scc1 L_ck_end_{tag}
    """)
    if P_sgpr:
        for j in range(6):
            E(f"v_mov_b32 v{20 + j}, {s('Q', j)}")
    else:
        E(f"""
        s_add_u32 {st(0)}, {s('tl')}, 1
        s_mul_i32 {st(0)}, {st(0)}, 24
        {f"v_add_u32 v36, {st(0)}, v1" if W4 else f"v_mov_b32 v36, {st(0)}"}
        ds_read_b64 v[20:21], v36 offset:{LDS_P0}
        ds_read_b64 v[22:23], v36 offset:{LDS_P0 + 8}
        ds_read_b64 v[24:25], v36 offset:{LDS_P0 + 16}
        s_waitcnt lgkmcnt(0)
        """)
    E(f"""
    v_bcnt_u32_b32 v46, {w0}, 0
    v_bcnt_u32_b32 v46, {w1}, v46
    v_mov_b32 v47, 0
    s_mov_b32 {st(7)}, 0
    L_ck_{tag}:
    v_lshl_or_b32 v36, {st(7)}, 6, {LANE}
    v_cmp_gt_u32 vcc, {s_Nw()}, v36
    v_mul_u32_u24 v37, 24, v36
    s_mov_b64 {stp(0)}, vcc
    s_mov_b64 exec, vcc
    global_load_dwordx4 v[14:17], v37, {sp('Rs')}
    global_load_dwordx2 v[18:19], v37, {sp('Rs')} offset:16
    s_waitcnt vmcnt(0)
    s_mov_b64 exec, -1
    v_add_f64 v[38:39], v[20:21], -v[14:15]
    v_add_f64 v[40:41], v[22:23], -v[16:17]
    v_add_f64 v[42:43], v[24:25], -v[18:19]
    v_mul_f64 v[14:15], v[38:39], {sp('invL')}
    v_mul_f64 v[16:17], v[40:41], {sp('invL')}
    v_rndne_f64 v[14:15], v[14:15]
    v_rndne_f64 v[16:17], v[16:17]
    v_fma_f64 v[38:39], -v[14:15], {sp('L')}, v[38:39]
    v_fma_f64 v[40:41], -v[16:17], {sp('L')}, v[40:41]
    v_mul_f64 v[14:15], v[38:39], v[38:39]
    v_fma_f64 v[14:15], v[40:41], v[40:41], v[14:15]
    v_fma_f64 v[14:15], v[42:43], v[42:43], v[14:15]
    v_cmp_gt_f64 {stp(2)}, {sp('cut2')}, v[14:15]
    s_and_b64 {stp(2)}, {stp(2)}, {stp(0)}
    """)
    for loc in locs:      # (register, guard) pairs: the cell is not a neighbour when guard != 0
        if W4:            # v36 counts this wave's cells: compare with the cell's index inside its owner wave, if that is us
            E(f"""
            {f"s_and_b32 {st(6)}, {WAVE}, {KS - 1}" if TT else ""}
            s_lshl_b32 {st(6)}, {st(6) if TT else WAVE}, {WSH}
            s_sub_u32 {st(6)}, {loc[0]}, {st(6)}
            """)
        E(f"""
        v_cmp_ne_u32 {stp(4)}, {st(6) if W4 else loc[0]}, v36
        s_cmp_eq_u32 {loc[1]}, 0
        s_cselect_b64 {stp(4)}, -1, {stp(4)}
        s_and_b64 {stp(2)}, {stp(2)}, {stp(4)}
        """)
    E(f"""
    v_lshrrev_b64 v[44:45], {st(7)}, v[{w0[1:]}:{w1[1:]}]
    v_and_b32 v44, 1, v44
    v_cmp_eq_u32 {stp(4)}, 1, v44
    s_andn2_b64 {stp(4)}, {stp(2)}, {stp(4)}
    s_bcnt1_i32_b64 {st(0)}, {stp(2)}
    s_bcnt1_i32_b64 {st(1)}, {stp(4)}
    s_lshl_b32 {st(1)}, {st(1)}, 16
    s_add_u32 {st(0)}, {st(0)}, {st(1)}
    v_add_u32 v47, {st(0)}, v47
    s_add_u32 {st(7)}, {st(7)}, 1
    s_cmp_lt_u32 {st(7)}, {NS}
    s_cbranch_scc1 L_ck_{tag}
    // lane 0: inside += v47 & 0xffff ; missed += v47 >> 16 ; every lane: candidate bits
    v_and_b32 v44, 0xffff, v47
    v_lshrrev_b32 v45, 16, v47
    """)
    cnt_addr("v36")
    E(f"""
    ds_add_u32 v36, v46 offset:{LDS_CNT + 4}
    s_mov_b64 exec, 1
    ds_add_u32 v36, v44 offset:{LDS_CNT}
    ds_add_u32 v36, v45 offset:{LDS_CNT + 8}
    s_mov_b64 exec, -1
    s_waitcnt lgkmcnt(0)
    L_ck_end_{tag}:
    """)
    fill_p0_end()     # (mg: the check used the registers of the vector constants)


NEXCL = [0]


def excl(w0, w1, loc):
    """zb: the particle in cell `loc` (s: slot << 6 | lane; z8w: wave << 12 | ...) is not a candidate"""
    NEXCL[0] += 1
    if W4:   # only the wave that owns the cell holds its bit
        E(f"""
        s_lshr_b32 {st(1)}, {loc}, {WSH}
        {f"s_and_b32 {st(2)}, {WAVE}, {KS - 1}" if TT else ""}
        s_cmp_lg_u32 {st(1)}, {st(2) if TT else WAVE}
        s_cbranch_scc1 L_excl{NEXCL[0]}
        """)
    # lane loc & 63 clears bit (slot) of its flag words: d = (mask & 0) | (~mask & w)
    E(f"""
    {f"s_bfe_u32 {st(1)}, {loc}, {SLOTF}" if W4 else f"s_lshr_b32 {st(1)}, {loc}, 6"}
    s_lshl_b64 exec, 1, {loc}
    s_bfm_b64 {stp(4)}, 1, {st(1)}
    v_bfi_b32 {w0}, {st(4)}, 0, {w0}
    v_bfi_b32 {w1}, {st(5)}, 0, {w1}
    s_mov_b64 exec, -1
    """)
    if W4:
        E(f"L_excl{NEXCL[0]}:")


# ---------------------------------------------------------------------------------------------- helpers
def pick_fetch(w0, w1, X, spec_mask, have):
    """lanes with a candidate (and not in spec_mask) take their lowest one out of w and load its fp64
    position into X[0:5]; `have` (s pair) <- the lanes whose load is in flight"""
    E(f"""
    v_cmp_ne_u64 vcc, 0, v[{w0}:{w1}]
    s_andn2_b64 {have}, vcc, {spec_mask}
    s_mov_b64 exec, {have}
    v_ffbl_b32 v44, v{w0}
    v_ffbl_b32 v45, v{w1}
    v_lshl_add_u64 v[46:47], v[{w0}:{w1}], 0, -1
    v_or_b32 v45, 32, v45
    v_min_u32 v44, v44, v45
    v_and_b32 v{w0}, v{w0}, v46
    v_and_b32 v{w1}, v{w1}, v47
    {"" if ZB else f"v_add_u32 v44, {s('rot')}, v44"}
    {"" if ZB else f"v_and_b32 v44, {NS - 1}, v44"}
    v_lshl_or_b32 v44, v44, 6, {LANE}
    {"" if ZB else f"v_cmp_gt_u32 vcc, {s('N')}, v44"}
    v_mul_u32_u24 v45, 24, v44
    {"v_mov_b32 v45, 0" if FAKE else ""}
    {"" if ZB else "s_and_b64 exec, exec, vcc"}
    {f"ds_read2_b64 v[{X}:{X+3}], v45 offset1:1" if LP else f"global_load_dwordx4 v[{X}:{X+3}], v45, {SRC}"}
    {f"ds_read_b64 v[{X+4}:{X+5}], v45 offset:16" if LP else f"global_load_dwordx2 v[{X+4}:{X+5}], v45, {SRC} offset:16"}
    s_mov_b64 {have}, exec
    s_mov_b64 exec, -1
    """)


def coeff_one(C):
    E(f"""
    v_mov_b64 v[{C}:{C+1}], 1.0
    v_mov_b64 v[{C+2}:{C+3}], 1.0
    """)


def wall_fetch(X, C):
    """lanes 0..M2: site position (sx, sy) into X[0:3] and the coefficients (ca, cb) into C[0:3], from the table;
    every other lane's coefficients are 1"""
    coeff_one(C)
    E(f"""
    s_mov_b64 exec, {sp('wallM')}
    v_lshlrev_b32 v46, 5, {LANE}
    global_load_dwordx4 v[{X}:{X+3}], v46, {sp('wtab')}
    global_load_dwordx4 v[{C}:{C+3}], v46, {sp('wtab')} offset:16
    s_mov_b64 exec, -1
    """)


def wall_dz(tag, pz_is_sgpr, pz):
    """wdz (uniform) = signed distance of pz to the nearer wall with the reference's clamp (SMC.c:736-739)"""
    if pz_is_sgpr:
        E(f"v_mov_b32 {v('wdz')}, {pz[0]}")
        E(f"v_mov_b32 {v('wdz',1)}, {pz[1]}")
        src = vp('wdz')
    else:
        src = pz
    if ZB:
        E(f"""
        v_add_f64 {vp('T')}, {src}, {sp('halfLz')}
        v_cmp_ge_f64 vcc, |{src}|, {sp('halfLz')}
        v_mul_f64 {vp('S6')}, {vp('T')}, {sp('invLz')}
        v_rndne_f64 {vp('S6')}, {vp('S6')}
        v_fma_f64 {vp('wdz')}, -{vp('S6')}, {sp('Lz')}, {vp('T')}
        s_cbranch_vccz L_wdz_{tag}
        // at or beyond a wall: +1e-4 below the lower one, -1e-4 above the upper one (T = pz + Lz/2 <= 0: the lower)
        v_cmp_ge_f64 {stp(0)}, 0, {vp('T')}
        v_mov_b32 {v('wdz')}, 0xeb1c432d
        v_mov_b32 {v('T')}, 0xbf1a36e2
        v_mov_b32 {v('T',1)}, 0x3f1a36e2
        s_nop 0
        v_cndmask_b32 {v('wdz',1)}, {v('T')}, {v('T',1)}, {stp(0)}
        L_wdz_{tag}:
        """)
        return
    E(f"""
    v_add_f64 {vp('T')}, {src}, {sp('halfLz')}
    v_cmp_le_f64 {stp(0)}, {src}, -{sp('halfLz')}
    v_cmp_ge_f64 {stp(2)}, {src}, {sp('halfLz')}
    v_mul_f64 {vp('S6')}, {vp('T')}, {sp('invLz')}
    v_rndne_f64 {vp('S6')}, {vp('S6')}
    v_fma_f64 {vp('wdz')}, -{vp('S6')}, {sp('Lz')}, {vp('T')}
    s_or_b64 {stp(4)}, {stp(0)}, {stp(2)}
    s_cmp_lg_u64 {stp(4)}, 0
    s_cbranch_scc0 L_wdz_{tag}
    // at or beyond a wall: +1e-4 below the lower one, -1e-4 above the upper one
    v_mov_b32 {v('wdz')}, 0xeb1c432d
    v_mov_b32 {v('T')}, 0xbf1a36e2
    v_mov_b32 {v('T',1)}, 0x3f1a36e2
    v_cndmask_b32 {v('wdz',1)}, {v('T')}, {v('T',1)}, {stp(0)}
    L_wdz_{tag}:
    """)


def body(tag, P, X, C, items, round0, wl=None, pl=None):
    """the fp64 body for the lanes in `items` (s pair): d = probe - X, the walls' dz, signed minimum image,
    the plane's rules, cutoff test, lj_acc's sequence (SMC.c:567-578, 601-614, 740-761, 787-809)"""
    wl = wl or sp('wallM')
    pl = pl or sp('planeM')
    Xp = X if isinstance(X, (list, tuple)) else [f"v[{X}:{X+1}]", f"v[{X+2}:{X+3}]", f"v[{X+4}:{X+5}]"]
    ca, cb = ("1.0", "1.0") if C is None else (f"v[{C}:{C+1}]", f"v[{C+2}:{C+3}]")
    E(f"""
    s_mov_b64 exec, {items}
    v_add_f64 {vp('D',0)}, {P[0]}, -{Xp[0]}
    v_add_f64 {vp('D',1)}, {P[1]}, -{Xp[1]}
    v_add_f64 {vp('D',2)}, {P[2]}, -{Xp[2]}
    """)
    if round0 and ZB:
        E(f"""
        v_cndmask_b32 {v('D',4)}, {v('D',4)}, {v('wdz')}, {wl}
        v_cndmask_b32 {v('D',5)}, {v('D',5)}, {v('wdz',1)}, {wl}
        """)
    elif round0:
        E(f"""
        s_and_b64 exec, {items}, {wl}
        v_mov_b32 {v('D',4)}, {v('wdz')}
        v_mov_b32 {v('D',5)}, {v('wdz',1)}
        s_mov_b64 exec, {items}
        """)
    E(f"""
    v_mul_f64 {vp('T')}, {vp('D',0)}, {sp('invL')}
    v_mul_f64 {vp('S6')}, {vp('D',1)}, {sp('invL')}
    v_rndne_f64 {vp('T')}, {vp('T')}
    v_rndne_f64 {vp('S6')}, {vp('S6')}
    v_fma_f64 {vp('M',0)}, -{vp('T')}, {sp('L')}, {vp('D',0)}
    v_fma_f64 {vp('M',1)}, -{vp('S6')}, {sp('L')}, {vp('D',1)}
    """)
    if round0 and ZB:
        E(f"""
        v_cndmask_b32 {v('M',0)}, {v('M',0)}, 0, {pl}
        v_cndmask_b32 {v('M',1)}, {v('M',1)}, 0, {pl}
        v_cndmask_b32 {v('M',2)}, {v('M',2)}, 0, {pl}
        v_cndmask_b32 {v('M',3)}, {v('M',3)}, 0, {pl}
        """)
    elif round0:
        E(f"""
        s_and_b64 exec, {items}, {pl}
        v_mov_b32 {v('M',0)}, 0
        v_mov_b32 {v('M',1)}, 0
        v_mov_b32 {v('M',2)}, 0
        v_mov_b32 {v('M',3)}, 0
        s_mov_b64 exec, {items}
        """)
    E(f"""
    v_mul_f64 {vp('dr2')}, {vp('M',0)}, {vp('M',0)}
    v_fma_f64 {vp('dr2')}, {vp('M',1)}, {vp('M',1)}, {vp('dr2')}
    v_fma_f64 {vp('dr2')}, {vp('D',2)}, {vp('D',2)}, {vp('dr2')}
    v_cmp_gt_f64 vcc, {sp('cut2')}, {vp('dr2')}
    """)
    if round0:
        E(f"s_or_b64 vcc, vcc, {pl}")
    # (round 0 of the merged pass: with walls a plane lane is always in, and without them an empty exec costs the same instructions)
    E(f"""
    s_and_b64 exec, exec, vcc
    {"" if (MG and round0) else f"s_cbranch_execz L_nolj_{tag}"}
    v_rcp_f64 {vp('ir2')}, {vp('dr2')}
    s_nop 0
    v_fma_f64 {vp('T')}, -{vp('dr2')}, {vp('ir2')}, 1.0
    v_fma_f64 {vp('ir2')}, {vp('T')}, {vp('ir2')}, {vp('ir2')}
    v_fma_f64 {vp('T')}, -{vp('dr2')}, {vp('ir2')}, 1.0
    v_fma_f64 {vp('ir2')}, {vp('T')}, {vp('ir2')}, {vp('ir2')}
    v_mul_f64 {vp('T')}, {vp('ir2')}, {vp('ir2')}
    v_mul_f64 {vp('S6')}, {vp('T')}, {vp('ir2')}
    v_mul_f64 {vp('T')}, {ca}, {vp('S6')}
    v_mul_f64 {vp('F')}, {cb}, {vp('S6')}
    v_mul_f64 {vp('T')}, {vp('T')}, {vp('S6')}
    v_add_f64 {vp('S6')}, {vp('T')}, -{vp('F')}
    v_mul_f64 {vp('F')}, {vp('F')}, {sp('neg24')}
    v_add_f64 {vp('acc',0)}, {vp('acc',0)}, {vp('S6')}
    v_fmac_f64 {vp('F')}, 0x40480000, {vp('T')}
    v_mul_f64 {vp('F')}, {vp('ir2')}, {vp('F')}
    v_fma_f64 {vp('acc',1)}, {vp('F')}, {vp('M',0)}, {vp('acc',1)}
    v_fma_f64 {vp('acc',2)}, {vp('F')}, {vp('M',1)}, {vp('acc',2)}
    v_fma_f64 {vp('acc',3)}, {vp('F')}, {vp('D',2)}, {vp('acc',3)}
    L_nolj_{tag}:
    s_mov_b64 exec, -1
    """)


def reduce4(dst):
    """dst pair <- totals of acc[0..3], one per 16-lane row"""
    a = [V['acc'] + 2 * j for j in range(4)]
    lo, hi = (int(x) for x in dst[2:-1].split(":"))
    E(f"""
    s_nop 1
    v_permlane32_swap_b32 v{a[0]}, v{a[2]}
    v_permlane32_swap_b32 v{a[0]+1}, v{a[2]+1}
    v_permlane32_swap_b32 v{a[1]}, v{a[3]}
    v_permlane32_swap_b32 v{a[1]+1}, v{a[3]+1}
    s_nop 0
    v_add_f64 v[{a[0]}:{a[0]+1}], v[{a[0]}:{a[0]+1}], v[{a[2]}:{a[2]+1}]
    v_add_f64 v[{a[1]}:{a[1]+1}], v[{a[1]}:{a[1]+1}], v[{a[3]}:{a[3]+1}]
    s_nop 1
    v_permlane16_swap_b32 v{a[0]}, v{a[1]}
    v_permlane16_swap_b32 v{a[0]+1}, v{a[1]+1}
    s_nop 0
    v_add_f64 {dst}, v[{a[0]}:{a[0]+1}], v[{a[1]}:{a[1]+1}]
    """)
    for ctrl in ("row_ror:8", "row_half_mirror", "quad_perm:[2,3,0,1]", "quad_perm:[1,0,3,2]"):
        E(f"""
        s_nop 1
        v_mov_b32_dpp {v('T')}, v{lo} {ctrl} row_mask:0xf bank_mask:0xf bound_ctrl:1
        v_mov_b32_dpp {v('T',1)}, v{hi} {ctrl} row_mask:0xf bank_mask:0xf bound_ctrl:1
        v_add_f64 {dst}, {dst}, {vp('T')}
        """)


def xchg(dst, buf):
    """z8w: dst (row layout, this wave's partial sums) <- the sum over the waves of the replica, added in wave order by
    every wave alike; two buffers alternate (probe A, probe B) so that a fast wave cannot overwrite what a slow one
    still reads"""
    X = LDS_X + buf * WPR * 512
    E(f"""
    s_lshl_b32 {st(0)}, {WAVE}, 9
    v_lshl_add_u32 v44, {LANE}, 3, {st(0)}
    v_lshlrev_b32 v45, 3, {LANE}
    ds_write_b64 v44, {dst} offset:{X}
    s_waitcnt lgkmcnt(0)
    s_barrier
    """)
    for k in range(0, WPR, 4):           # four partial sums at a time in v46..v53
        for j in range(4):
            E(f"ds_read_b64 v[{46 + 2 * j}:{47 + 2 * j}], v45 offset:{X + 512 * (k + j)}")
        E("s_waitcnt lgkmcnt(0)")
        if k == 0:
            E(f"v_add_f64 {dst}, v[46:47], v[48:49]")
        else:
            E(f"v_add_f64 {dst}, {dst}, v[46:47]")
            E(f"v_add_f64 {dst}, {dst}, v[48:49]")
        E(f"v_add_f64 {dst}, {dst}, v[50:51]")
        E(f"v_add_f64 {dst}, {dst}, v[52:53]")


def side_capture():
    """z8t, team B's slab-0 wave after round 0 of probe B: the accumulators of the two side lanes (each holds that one
    item) go to the side area of this move's buffer -- [old, new][e, fx, fy, fz] -- and are zeroed, so that the reduction
    carries probe B WITHOUT the pair (n, n+1); every wave adds the one that applies after the decision"""
    a = [vp('acc', j) for j in range(4)]
    E(f"""
    s_cmp_eq_u32 {s_hasAw()}, 0
    s_cbranch_scc1 L_nocap
    s_lshl_b64 {stp(0)}, 1, {s('sideL')}
    s_lshl_b64 {stp(2)}, 1, {s('sideN')}
    {"" if TL else f"s_and_b32 {st(4)}, {s('i')}, 1"}
    {"" if TL else f"s_lshl_b32 {st(4)}, {st(4)}, 6"}
    s_or_b64 exec, {stp(0)}, {stp(2)}
    v_cmp_eq_u32 vcc, {s('sideN')}, {LANE}
    {f"v_and_b32 v44, 64, {VSR}" if TL else f"v_mov_b32 v44, {st(4)}"}
    s_nop 1
    v_cndmask_b32_e64 v45, 0, 32, vcc
    v_add_u32 v44, v44, v45
    ds_write_b64 v44, {a[0]} offset:{LDS_SIDE}
    ds_write_b64 v44, {a[1]} offset:{LDS_SIDE + 8}
    ds_write_b64 v44, {a[2]} offset:{LDS_SIDE + 16}
    ds_write_b64 v44, {a[3]} offset:{LDS_SIDE + 24}
    v_mov_b64 {a[0]}, 0
    v_mov_b64 {a[1]}, 0
    v_mov_b64 {a[2]}, 0
    v_mov_b64 {a[3]}, 0
    s_mov_b64 exec, -1
    L_nocap:
    """)


def xchg2(part, fn, fb, extra=""):
    """z8t: the ONE exchange of a move.  `part` (row layout) = this wave's partial sums of ITS team's probe; afterwards
    fn = sum over team A's waves (probe A), fb = sum over team B's (probe B without the side pair), added in wave order
    by every wave alike.  Two buffers alternate with the parity of the move counter, so that a wave that is already in
    the next move cannot overwrite what a slower one still reads (the next barrier stops it before the move after)."""
    if TL:
        XW_, XR_ = VXW, VXR
    else:
        XW_, XR_ = "v44", "v45"
        E(f"""
        s_and_b32 {st(1)}, {s('i')}, 1
        s_mul_i32 {st(1)}, {st(1)}, {WPR * 512}
        s_lshl_b32 {st(0)}, {WAVE}, 9
        s_add_u32 {st(0)}, {st(0)}, {st(1)}
        v_lshl_add_u32 v44, {LANE}, 3, {st(0)}
        v_lshl_add_u32 v45, {LANE}, 3, {st(1)}
        """)
    E(f"""
    ds_write_b64 {XW_}, {part} offset:{LDS_X}
    s_waitcnt lgkmcnt(0)
    """)
    mark(5)
    E("s_barrier")
    mark(6)
    E(extra)      # further LDS reads of the caller: they travel with the exchange reads
    if KS == 4:   # all eight reads in flight at once (team B's into v24..v31, dead since the probes): one LDS round trip instead of two
        for t0, w0_ in ((46, 0), (24, KS)):
            for j in range(4):
                E(f"ds_read_b64 v[{t0 + 2 * j}:{t0 + 2 * j + 1}], {XR_} offset:{LDS_X + 512 * (w0_ + j)}")
        for dst, t0, cnt in ((fn, 46, 4), (fb, 24, 0)):
            E(f"""s_waitcnt lgkmcnt({cnt})
            v_add_f64 {dst}, v[{t0}:{t0 + 1}], v[{t0 + 2}:{t0 + 3}]
            v_add_f64 {dst}, {dst}, v[{t0 + 4}:{t0 + 5}]
            v_add_f64 {dst}, {dst}, v[{t0 + 6}:{t0 + 7}]""")
        return
    for dst, w0_ in ((fn, 0), (fb, KS)):
        if KS == 1:
            E(f"ds_read_b64 {dst}, {XR_} offset:{LDS_X + 512 * w0_}")
            continue
        for k in range(0, KS, 4):
            n = min(4, KS - k)
            for j in range(n):
                E(f"ds_read_b64 v[{46 + 2 * j}:{47 + 2 * j}], {XR_} offset:{LDS_X + 512 * (w0_ + k + j)}")
            E("s_waitcnt lgkmcnt(0)")
            for j in range(n):
                if k == 0 and j == 0:
                    continue
                if k == 0 and j == 1:
                    E(f"v_add_f64 {dst}, v[46:47], v[48:49]")
                else:
                    E(f"v_add_f64 {dst}, {dst}, v[{46 + 2 * j}:{47 + 2 * j}]")
    E("s_waitcnt lgkmcnt(0)")


def probe(tag, P, pz_sgpr, pz, w0, w1, X, C, have, side, wait, wl=None, pl=None):
    """a whole probe: round 0 (specials + the first candidates, already requested into X / `have`), then
    further rounds while any lane still has a candidate.  `wait`: the s_waitcnt that covers round 0's loads"""
    wl = wl or sp('wallM')
    pl = pl or sp('planeM')
    for j in range(4):
        E(f"v_mov_b64 {vp('acc', j)}, 0")
    E(f"s_cmp_lg_u64 {wl}, 0")
    E(f"s_cbranch_scc0 L_nw_{tag}")
    wall_dz(tag, pz_sgpr, pz)
    E(f"L_nw_{tag}:")
    E(f"s_or_b64 {stp(6)}, {have}, {wl}")
    if side:
        GW(f"s_cmp_eq_u32 {s_hasAw()}, 0")
        GW(f"s_cbranch_scc1 L_noside_{tag}")
        if ZB:
            E(f"s_lshl_b64 {stp(0)}, 1, {s('sideL')}")
            E(f"s_or_b64 {stp(6)}, {stp(6)}, {stp(0)}")
            if TT:
                E(f"s_lshl_b64 {stp(0)}, 1, {s('sideN')}")
                E(f"s_or_b64 {stp(6)}, {stp(6)}, {stp(0)}")
        else:
            E(f"s_or_b64 {stp(6)}, {stp(6)}, {sp('sideM')}")
        E(f"L_noside_{tag}:")
    E(wait)
    body(tag + "r0", P, X, C, stp(6), True, wl, pl)
    if side and TT:   # now, while the side lanes hold nothing but their side item (the fixed-lane fallback gives them
        side_capture()  # candidates of their own in later rounds)
    mark(3)
    if PF2:           # the second candidates, fetched together with the first
        X2, _, _, have2 = second_regs(tag)
        E(f"""
        s_cmp_eq_u64 {have2}, 0
        s_cbranch_scc1 L_more_{tag}
        """)
        body(tag + "r0b", P, X2, None, have2, False)
    E(f"""
    L_more_{tag}:
    v_cmp_ne_u64 vcc, 0, v[{w0}:{w1}]
    {"s_cbranch_vccz L_done_" + tag if ZB else "s_cmp_lg_u64 vcc, 0"}
    {"" if ZB else "s_cbranch_scc0 L_done_" + tag}
    {f"L_moreB_{tag}:" if TL else ""}
    """)
    if Z8C:   # executed work: rounds beyond the first of a probe (a lane held two candidates)
        cnt_addr("v46")
        E(f"""
        v_mov_b32 v44, 1
        s_mov_b64 exec, 1
        ds_add_u32 v46, v44 offset:{LDS_CNT + 20}
        s_mov_b64 exec, -1
        """)
    coeff_one(C)
    pick_fetch(w0, w1, X, "0", stp(6))
    if TT:
        tt_exclude(tag, stp(6))       # (v44 = the cell pick_fetch took from this lane's flag words)
    E("s_waitcnt lgkmcnt(0)" if LP else "s_waitcnt vmcnt(0)")
    body(tag + "rm", P, X, C, stp(6), False)
    if TL:   # a further round gave every lane coefficients 1, 1 and its own candidate: the wall lanes take their rows back
        E(f"""
        v_cmp_ne_u64 vcc, 0, v[{w0}:{w1}]
        s_cbranch_vccnz L_moreB_{tag}
        """)
        tt_wall_rows()
    else:
        E(f"s_branch L_more_{tag}")
    E(f"L_done_{tag}:")


def second_regs(tag):
    """z8t (PF2): where a lane's SECOND candidate goes -- registers the team does not use: (three pairs, dwordx4 base,
    dwordx2 base, mask pair of the lanes that have one)"""
    if tag == "A":   # team A never touches probe B's registers
        return ["v[20:21]", "v[22:23]", "v[24:25]"], 20, 24, sp('haveB')
    return ["v[26:27]", "v[28:29]", "v[4:5]"], 26, 4, sp('wallM')    # team B: probe A's coefficients and flag words


def mg_handover(w0, w1, h, start, off, n, cap=32):
    """every lane with a candidate in (w0, w1) hands its lowest one over: working lane = start + its rank among those lanes
    (of this probe's half; only below `cap`), through list[off / 4 + lane]; h (s pair) <- the lanes that did, n (s) <- how many"""
    hlo, hhi = (int(x) for x in h[2:-1].split(":"))
    E(f"""
    v_cmp_ne_u64 {h}, 0, v[{w0}:{w1}]
    v_ffbl_b32 v46, v{w0}
    v_ffbl_b32 v47, v{w1}
    {"v_or_b32 v47, 32, v47" if start.startswith("v") else f"v_mov_b32 v45, {start}"}
    v_mbcnt_lo_u32_b32 v45, s{hlo}, {start if start.startswith("v") else "v45"}
    v_mbcnt_hi_u32_b32 v45, s{hhi}, v45
    {"" if start.startswith("v") else "v_or_b32 v47, 32, v47"}
    v_cmp_gt_u32 vcc, {cap}, v45
    v_min_u32 v46, v46, v47
    v_lshl_add_u64 v[48:49], v[{w0}:{w1}], 0, -1
    s_and_b64 {h}, {h}, vcc
    v_lshl_or_b32 v46, v46, 6, {LANE}
    s_bcnt1_i32_b64 {n}, {h}
    {"v_lshl_add_u32 v45, v45, 2, v1" if W4 else "v_lshlrev_b32 v45, 2, v45"}
    s_mov_b64 exec, {h}
    v_and_b32 v{w0}, v{w0}, v48
    v_and_b32 v{w1}, v{w1}, v49
    ds_write_b32 v45, v46 offset:{LDS_LIST + off}
    s_mov_b64 exec, -1
    """)


def mg_handover2(tagc, w0, w1, h, start, off, n, cap=32, flag=True):
    """the lanes that STILL hold a candidate after the hand-over (two of one probe in one lane: ~15 % of the probes) hand
    that one over as well, to the working lanes behind the first batch, so that it is evaluated in the same pass instead
    of a round of its own; what remains after that (a third candidate, a full half) sets bit 1 of nearB = "more rounds".
    Cold piece; start (s), n (s): first working lane and count of the first batch, n is updated"""
    def emit(txt):
        (COLD if REDIR[0] is None else E)(txt)
    E(f"""
    v_cmp_ne_u64 vcc, 0, v[{w0}:{w1}]
    s_cbranch_vccnz L_ho2{tagc}
    L_ho2r{tagc}:
    """)
    save, REDIR[0] = REDIR[0], (cold if REDIR[0] is None else REDIR[0])
    if save is not None:      # already among the cold pieces: in line, jumped over
        E(f"s_branch L_ho2x{tagc}")
    E(f"""
    L_ho2{tagc}:
    s_add_u32 {st(7)}, {start}, {n}
    """)
    mg_handover(w0, w1, h, st(7), off, st(7), cap)
    E(f"s_add_u32 {n}, {n}, {st(7)}")
    if flag:   # (the two-team kernels look at the flag words again after round 0 instead)
        E(f"""
        v_cmp_ne_u64 vcc, 0, v[{w0}:{w1}]
        s_cbranch_vccz L_ho2r{tagc}
        s_bitset1_b32 {s('nearB')}, 1
        """)
    E(f"s_branch L_ho2r{tagc}")
    if save is not None:
        E(f"L_ho2x{tagc}:")
    REDIR[0] = save


def tt_wall_rows():
    """TL: the wall lanes' items (site x, y) and coefficients from the wall table (row = lane), 1, 1 on the other lanes"""
    E(f"""
    v_mov_b64 v[{KR+14}:{KR+15}], 1.0
    v_mov_b64 v[{KR+16}:{KR+17}], 1.0
    v_lshlrev_b32 v49, 5, {LANE}
    s_mov_b64 exec, {sp('wallM')}
    global_load_dwordx4 v[{KR+8}:{KR+11}], v49, {sp('wtab')}
    global_load_dwordx4 v[{KR+14}:{KR+17}], v49, {sp('wtab')} offset:16
    s_mov_b64 exec, -1
    s_waitcnt vmcnt(0)
    """)


def tt_exclude(tag, items):
    """XCT, steady copy: the lanes whose item (v44: a cell of this wave's slab) is a cell that is no neighbour of the probe --
    particle n for both teams, particle n+1 itself for team B -- leave `items` (s pair)"""
    if not XCT:
        return
    for loc in ([s('locA')] if tag.startswith("A") else [s('locB'), s('locA')]):
        SO(f"""
        s_sub_u32 {st(1)}, {loc}, {s('wbase')}
        v_cmp_ne_u32 vcc, {st(1)}, v44
        s_and_b64 {items}, {items}, vcc
        """)


def tt_assign(tag, w0, w1, X, C, wl, pl, with_side, have):
    """z8t (TL), round 0 of this wave's probe: its candidates are handed over to working lanes through the list -- behind the
    wall lanes (table row = lane; wl, pl, sideL, sideN are per-wave constants set in the prologue) and, on team B's slab-0
    wave, the two side lanes; `have` <- the lanes whose candidate load is in flight"""
    E(f"""
    s_add_u32 {st(3)}, {s('M2w')}, 1
    {f"s_lshl_b32 {st(6)}, {s('hasAw')}, 1" if with_side else ""}
    {f"s_add_u32 {st(6)}, {st(6)}, {st(3)}" if with_side else ""}
    """)
    start = st(6) if with_side else st(3)
    # at most TTCAP = 63 working lanes: s_bfm_b64 below takes its COUNT from 6 bits, so 64 items from lane 0 (a wave without
    # special lanes whose two hand-overs fill the wavefront: a condensed state) would give the mask 0 and drop all of them --
    # their bits are already out of the flag words.  Lane 63's would-be item stays in its owner's flag word: a further round.
    mg_handover(w0, w1, have, start, 0, st(5), cap=TTCAP)
    mg_handover2(tag + "t", w0, w1, have, start, 0, st(5), cap=TTCAP, flag=False)
    E(f"""
    ds_read_b32 v44, {KL4T} offset:{LDS_LIST}
    s_bfm_b64 {have}, {st(5)}, {start}
    """)
    if Z8C:   # diagnostic build: items handed over for which no working lane is enabled (must stay 0) -> counter word 7
        cnt_addr("v47")
        E(f"""
        s_bcnt1_i32_b64 {st(7)}, {have}
        s_sub_u32 {st(7)}, {st(5)}, {st(7)}
        v_mov_b32 v46, {st(7)}
        s_mov_b64 exec, 1
        ds_add_u32 v47, v46 offset:{LDS_CNT + 28}
        s_mov_b64 exec, -1
        """)
    E(f"""
    s_waitcnt lgkmcnt(0)
    v_mul_u32_u24 v45, 24, v44
    {"v_mov_b32 v45, 0" if FAKE else ""}
    s_mov_b64 exec, {have}
    global_load_dwordx4 v[{X}:{X+3}], v45, {SRC}
    global_load_dwordx2 v[{X+4}:{X+5}], v45, {SRC} offset:16
    s_mov_b64 exec, -1
    """)
    # (the wall lanes' rows of the table and every lane's coefficients are where the launch's prologue put them)
    tt_exclude(tag, have)


def assign_specials(tag, w0, w1, X, C, wl, pl, with_side, have):
    """zb, round 0 of a probe: every lane with a candidate takes its lowest one (load of its fp64 position asked
    for; `have` <- those lanes); the wall sites, the plane and -- for probe B after a move -- the side pair go to
    the first lanes WITHOUT a candidate (rank among them = row of the wall table), so no candidate waits for a
    second round behind them.  With too few free lanes (cold path): the fixed lanes 0..M2 (and 30), as in
    sweep_kernel_ma.  No test of the cell against N: an empty cell (z = 0x7fff) is never flagged, and the
    all-flagged words of an unsafe probe are cut to the real cells where they are made."""
    wlo, whi = (int(x) for x in wl[2:-1].split(":"))
    need = st(6) if with_side else st(3)
    E(f"v_cmp_ne_u64 {have}, 0, v[{w0}:{w1}]")
    if Z8C:   # statistics for a design question: would folding the lanes l and l + 32 onto one make a candidate wait?
        hlo, hhi = (int(x) for x in have[2:-1].split(":"))
        cnt_addr("v16")
        E(f"""
        s_and_b32 {st(0)}, s{hlo}, s{hhi}
        s_cmp_lg_u32 {st(0)}, 0
        s_cselect_b32 {st(0)}, 1, 0
        v_mov_b32 v14, {st(0)}
        s_mov_b64 exec, 1
        ds_add_u32 v16, v14 offset:{LDS_CNT + 24}
        s_mov_b64 exec, -1
        """)
    E(f"""
    s_add_u32 {st(3)}, {s_M2w()}, 1
    s_not_b64 {stp(0)}, {have}
    s_bcnt1_i32_b64 {st(2)}, {stp(0)}
    {f"s_add_u32 {st(6)}, {st(3)}, {s_hasAw()}" if with_side else ""}
    {f"s_add_u32 {st(6)}, {st(6)}, {s_hasAw()}" if with_side and TT else ""}
    s_cmp_lt_u32 {st(2)}, {need}
    s_cbranch_scc1 L_sps_{tag}
    v_mbcnt_lo_u32_b32 v48, {st(0)}, 0
    v_mbcnt_hi_u32_b32 v48, {st(1)}, v48
    v_ffbl_b32 v44, v{w0}
    v_ffbl_b32 v45, v{w1}
    v_cmp_gt_u32 {wl}, {st(3)}, v48
    v_cmp_eq_u32 {pl}, {s_M2w()}, v48
    """)
    if with_side:
        E(f"""
        v_cmp_eq_u32 {stp(4)}, {st(3)}, v48
        s_and_b64 {stp(4)}, {stp(4)}, {stp(0)}
        s_ff1_i32_b64 {s('sideL')}, {stp(4)}
        """)
        if TT:   # the free lane after it: the side pair with the proposal
            E(f"""
            s_add_u32 {st(6)}, {st(3)}, 1
            v_cmp_eq_u32 {stp(4)}, {st(6)}, v48
            s_and_b64 {stp(4)}, {stp(4)}, {stp(0)}
            s_ff1_i32_b64 {s('sideN')}, {stp(4)}
            """)
    E(f"""
    s_and_b64 {wl}, {wl}, {stp(0)}
    s_and_b64 {pl}, {pl}, {stp(0)}
    L_spj_{tag}:
    v_lshl_add_u64 v[46:47], v[{w0}:{w1}], 0, -1
    v_or_b32 v45, 32, v45
    v_min_u32 v44, v44, v45
    v_lshl_or_b32 v44, v44, 6, {LANE}
    v_mul_u32_u24 v45, 24, v44
    {"v_mov_b32 v45, 0" if FAKE else ""}
    v_lshlrev_b32 v49, 5, v48
    // only the lanes that evaluate their candidate now take it out of w (the fixed-lane fallback keeps some waiting)
    s_mov_b64 exec, {have}
    v_and_b32 v{w0}, v{w0}, v46
    v_and_b32 v{w1}, v{w1}, v47
    {f"ds_read2_b64 v[{X}:{X+3}], v45 offset1:1" if LP else f"global_load_dwordx4 v[{X}:{X+3}], v45, {SRC}"}
    {f"ds_read_b64 v[{X+4}:{X+5}], v45 offset:16" if LP else f"global_load_dwordx2 v[{X+4}:{X+5}], v45, {SRC} offset:16"}
    """)
    if PF2:   # the lanes that still hold a candidate take that one too, its load travelling with the first
        _, b4, b2, have2 = second_regs(tag)
        E(f"""
        v_cmp_ne_u64 {have2}, 0, v[{w0}:{w1}]
        s_mov_b64 exec, {have2}
        v_ffbl_b32 v44, v{w0}
        v_ffbl_b32 v45, v{w1}
        v_lshl_add_u64 v[46:47], v[{w0}:{w1}], 0, -1
        v_or_b32 v45, 32, v45
        v_min_u32 v44, v44, v45
        v_and_b32 v{w0}, v{w0}, v46
        v_and_b32 v{w1}, v{w1}, v47
        v_lshl_or_b32 v44, v44, 6, {LANE}
        v_mul_u32_u24 v45, 24, v44
        global_load_dwordx4 v[{b4}:{b4+3}], v45, {SRC}
        global_load_dwordx2 v[{b2}:{b2+1}], v45, {SRC} offset:16
        """)
    E(f"""
    s_mov_b64 exec, -1
    """)
    coeff_one(C)
    if LP:
        E(f"""
        s_mov_b64 exec, {wl}
        ds_read_b64 v[{X}:{X+1}], v49 offset:{LDS_WT}
        ds_read_b64 v[{X+2}:{X+3}], v49 offset:{LDS_WT + 8}
        ds_read_b64 v[{C}:{C+1}], v49 offset:{LDS_WT + 16}
        ds_read_b64 v[{C+2}:{C+3}], v49 offset:{LDS_WT + 24}
        s_mov_b64 exec, -1
        """)
    else:
        E(f"""
        s_mov_b64 exec, {wl}
        global_load_dwordx4 v[{X}:{X+3}], v49, {sp('wtab')}
        global_load_dwordx4 v[{C}:{C+3}], v49, {sp('wtab')} offset:16
        s_mov_b64 exec, -1
        """)
    COLD(f"""
    L_sps_{tag}:
    s_lshl_b64 {wl}, 1, {st(3)}
    s_sub_u32 s{wlo}, s{wlo}, 1
    s_subb_u32 s{whi}, s{whi}, 0
    s_lshl_b64 {pl}, 1, {s_M2w()}
    s_cmp_lt_i32 {s_M2w()}, 0
    s_cselect_b64 {pl}, 0, {pl}
    s_mov_b64 {stp(4)}, {wl}
    {f"s_mov_b32 {s('sideL')}, 30" if with_side else ""}
    {f"s_mov_b32 {s('sideN')}, 31" if with_side and TT else ""}
    {f"s_or_b32 {st(4)}, {st(4)}, {'0xc0000000' if TT else '0x40000000'}" if with_side else ""}
    s_andn2_b64 {have}, {have}, {stp(4)}
    v_mov_b32 v48, {LANE}
    v_ffbl_b32 v44, v{w0}
    v_ffbl_b32 v45, v{w1}
    s_branch L_spj_{tag}
    """)


def all_real_cells(w0, w1, back):
    """zb, unsafe probe (cold piece, returns to `back`): every REAL cell of this lane is a candidate (cell = slot * 64 + lane < N)"""
    COLD(f"""
    v_sub_u32 v14, {s_Nw()}, {LANE}
    v_add_u32 v14, 63, v14
    v_ashrrev_i32 v14, 6, v14
    v_max_i32 v14, 0, v14
    v_mov_b32 v16, 1
    v_mov_b32 v17, 0
    v_lshlrev_b64 v[16:17], v14, v[16:17]
    v_lshl_add_u64 v[16:17], v[16:17], 0, -1
    v_cmp_lt_u32 vcc, 63, v14
    v_cndmask_b32 {w0}, v16, -1, vcc
    v_cndmask_b32 {w1}, v17, -1, vcc
    s_branch {back}
    """)


XA_, CA_, XB_, CB_ = 30, 26, 20, 10   # round-0 data: A in the D registers (d = p - X in place), B in v20..25; coefficients
if TL:
    # z8t with the list hand-over: the wall lanes are the same lanes in every move, so what they work with -- the site's
    # (x, y) and its two coefficients, a row of the wall table -- is loaded ONCE per launch into registers of their own (and
    # 1, 1 on every other lane) that nothing else writes: items and coefficients of both teams' probes live there
    XA_ = XB_ = KR + 8
    CA_ = CB_ = KR + 14


# ---------------------------------------------------------------------------------------------- screen + fetch, probe A first
if TT:   # team A screens and fetches for probe A only, team B for probe B only
    E(f"""
    s_cmp_ge_u32 {WAVE}, {KS}
    s_cbranch_scc1 L_teamB
    """)
if Z8:
    screen_ranged8("A", s('axys'), v('wa0'), v('wa1'))
    mark(1, 31)
elif ZB:
    screen_ranged("A", s('axys'), s('azz'), v('wa0'), v('wa1'))
    if ZBC:
        zbc_check(s('azz'), v('wa0'), v('wa1'))
else:
    screen_pass(v('axy'), s('azz'), v('wa0'), v('wa1'))
E(f"""
v_or_b32 {v('wa0')}, {v('wa0')}, {v('uns0')}
v_or_b32 {v('wa1')}, {v('wa1')}, {v('uns1')}
{'' if ZB else f"s_mov_b64 {sp('haveA')}, 0"}
""")
G(f"""
s_cmp_eq_u32 {s('hasA')}, 0
s_cbranch_scc1 {"L_nofaC" if MG else "L_nofa"}
""")
if MG:
    COLD(f"""
    L_nofaC:
    v_mov_b32 {v('wa0')}, 0
    v_mov_b32 {v('wa1')}, 0
    s_branch L_nofa
    """)
E(f"""
s_cmp_{"lg" if ZB else "eq"}_u32 {s('ua')}, 0
s_cbranch_scc1 {"L_uaC" if ZB else "L_ua0"}
""")
if ZB:
    COLD("L_uaC:")
    all_real_cells(v('wa0'), v('wa1'), "L_ua0")
else:
    E(f"v_mov_b32 {v('wa0')}, {'-1' if NS >= 32 else '0xffff'}")
    E(f"v_mov_b32 {v('wa1')}, {'-1' if NS == 64 else '0'}")
E(f"""
L_ua0:
""")
if ZB:
    FORCE_TAG[0] = " @G" if (XC or XCT) else ""
    excl(v('wa0'), v('wa1'), s('locA'))    # the moving particle itself is not a neighbour of its proposal
    FORCE_TAG[0] = ""
else:
    E(f"""
    // the moving particle itself (slot 0 of lane tl) is not a neighbour of its proposal
    s_lshl_b64 {stp(0)}, 1, {s('tl')}
    s_mov_b64 exec, {stp(0)}
    v_and_b32 {v('wa0')}, -2, {v('wa0')}
    s_mov_b64 exec, -1
    """)
if Z8C:
    z8c_check("A", True, v('wa0'), v('wa1'), [(s('locA'), "1")], s('hasA'))
if MG:
    pass
elif ZB:
    (tt_assign if TL else assign_specials)("A", V['wa0'], V['wa1'], XA_, CA_, sp('wallM'), sp('planeM'), False, sp('haveA'))
else:
    pick_fetch(V['wa0'], V['wa1'], XA_, sp('wallM'), sp('haveA'))
    wall_fetch(XA_, CA_)
E("L_nofa:")
if TT:
    E(f"""
    s_branch L_nofb
    L_teamB:
    {"s_setprio 3" if TTP in ("probe", "red") else ""}
    """)
if Z8:
    if PS:
        SO(f"""
        s_cmp_lg_u32 {s('hB', 1)}, 0
        s_cbranch_scc1 L_ub0
        """)
    screen_ranged8("B", s('bxys'), v('wb0'), v('wb1'))
    mark(1, 32)
elif ZB:
    screen_ranged("B", s('bxys'), s('bzz'), v('wb0'), v('wb1'))
    if ZBC:
        zbc_check(s('bzz'), v('wb0'), v('wb1'))
else:
    screen_pass(v('bxy'), s('bzz'), v('wb0'), v('wb1'))
E(f"""
v_or_b32 {v('wb0')}, {v('wb0')}, {v('uns0')}
v_or_b32 {v('wb1')}, {v('wb1')}, {v('uns1')}
{'' if ZB else f"s_mov_b64 {sp('haveB')}, 0"}
""")
G(f"""
s_cmp_eq_u32 {s('hasB')}, 0
s_cbranch_scc1 {"L_nofbC" if MG else "L_nofb0"}
""")
if MG:
    COLD(f"""
    L_nofbC:
    v_mov_b32 {v('wb0')}, 0
    v_mov_b32 {v('wb1')}, 0
    s_branch L_nofb
    """)
if not Z8:
    E(f"""
    // log-uniform of move i+1 (scalar load: only now that no screen pass is running on the LDS counter)
    s_add_u32 {st(0)}, {s('i')}, 1
    s_lshl_b32 {st(0)}, {st(0)}, 3
    s_load_dwordx2 {sp('nlu')}, {sp('uK')}, {st(0)}
    """)
UBTEST = f"s_bitcmp1_b32 {s('ub')}, 16" if Z8 else ("s_cmp_%s_u32 %s, 0" % ("lg" if ZB else "eq", s('ub')))
E(f"""
{UBTEST}
s_cbranch_scc1 {"L_ubC" if ZB else "L_ub0"}
""")
if ZB:
    COLD("L_ubC:")
    all_real_cells(v('wb0'), v('wb1'), "L_ub0")
else:
    E(f"v_mov_b32 {v('wb0')}, {'-1' if NS >= 32 else '0xffff'}")
    E(f"v_mov_b32 {v('wb1')}, {'-1' if NS == 64 else '0'}")
E(f"""
L_ub0:
""")
if ZB:
    # not neighbours of B: the particle it stands for, and the moving particle n, which reaches B through the side pair
    FORCE_TAG[0] = " @G" if (XC or XCT) else ""
    excl(v('wb0'), v('wb1'), s('locB'))
    FORCE_TAG[0] = ""
    G(f"""
    s_cmp_eq_u32 {s('hasA')}, 0
    s_cbranch_scc1 L_fb1
    """)
    FORCE_TAG[0] = " @G" if (XC or XCT) else ""
    excl(v('wb0'), v('wb1'), s('locA'))
    FORCE_TAG[0] = ""
    E(f"""
    L_fb1:
    // B's four loads (candidate x2, wall row x2) are asked for without waiting for probe A's: probe A's body waits
    // with vmcnt(4) -- its own data, issued first, complete first.  (The diagnostic build's all-cells test drains
    // the counter itself.)
    {"s_waitcnt vmcnt(0)" if (ZBC or Z8C) else ""}
    """)
else:
    E(f"""
    // not neighbours of B: the particle it stands for (slot 0 of lane tl+1, or slot 1 of lane 0 when the order
    // crosses slots) and the moving particle n (slot 0 of lane tl), which reaches B through the side pair
    s_lshl_b64 {stp(0)}, 1, {s('lb')}
    s_mov_b64 exec, {stp(0)}
    s_cmp_eq_u32 {s('cross')}, 1
    s_cbranch_scc1 L_exBc
    v_and_b32 {v('wb0')}, -2, {v('wb0')}
    s_branch L_exBd
    L_exBc:
    v_and_b32 {v('wb0')}, -3, {v('wb0')}
    L_exBd:
    s_mov_b64 {stp(0)}, {sp('wallM')}
    s_cmp_eq_u32 {s('hasA')}, 0
    s_cbranch_scc1 L_fb1
    s_lshl_b64 {stp(2)}, 1, {s('tl')}
    s_mov_b64 exec, {stp(2)}
    v_and_b32 {v('wb0')}, -2, {v('wb0')}
    s_or_b64 {stp(0)}, {sp('wallM')}, {sp('sideM')}
    L_fb1:
    s_mov_b64 exec, -1
    // probe A's data has had the whole second pass to arrive; B's travels while probe A is evaluated
    s_waitcnt vmcnt(0)
    """)
if Z8C:
    z8c_check("B", False, v('wb0'), v('wb1'), [(s('locB'), "1"), (s('locA'), s('hasA'))], s('hasB'))
if MG:
    pass
elif ZB:
    (tt_assign if TL else assign_specials)("B", V['wb0'], V['wb1'], XB_, CB_, sp('wallB'), sp('planeB'), True, sp('haveB'))
else:
    pick_fetch(V['wb0'], V['wb1'], XB_, stp(0), sp('haveB'))
    wall_fetch(XB_, CB_)
MUTE[0] = MG
# probe B's fp64 position from the LDS cache: row = cross ? 64 : tl + 1 (= tl + 1 either way)
E(f"""
s_add_u32 {st(0)}, {s('tl')}, 1
s_mul_i32 {st(0)}, {st(0)}, 24
{f"v_add_u32 {v('T')}, {st(0)}, v1" if W4 else f"v_mov_b32 {v('T')}, {st(0)}"}
ds_read_b64 v[14:15], {v('T')} offset:{LDS_P0}
ds_read_b64 v[16:17], {v('T')} offset:{LDS_P0 + 8}
ds_read_b64 v[18:19], {v('T')} offset:{LDS_P0 + 16}
s_branch L_nofb
L_nofb0:
s_waitcnt vmcnt(0)
L_nofb:
""")
MUTE[0] = False
if MG:
    E("L_nofb:")
# ---------------------------------------------------------------------------------------------- mg: both probes in one pass
PV = ["v[14:15]", "v[16:17]", "v[18:19]"]      # mg: the probe of this lane's half (A: the proposal Q, B: particle n+1)
KSTA, KSTB = "v22", "v23"                      # mg: wlp's low word and stB as vector constants (start values of the ranks)
MGW = "v[20:21]"                               # mg: probe B's sums (without the side pair), in both halves
DdNm = "v[12:13]"                              # mg: displacement of move i+1 in group layout, asked for during the pass


def mg_exclude():
    """XC, steady copy: a working lane whose item (v44) is particle n's cell (either half) or particle n+1's (half B) leaves the
    candidates' mask stp(0).  Several wavefronts: an item is a cell of THIS wave's slab, so the compare is with cell - slab base
    (no item equals it when the cell is another wave's)"""
    if W4:
        SO(f"""
        s_lshl_b32 {st(6)}, {WAVE}, {WSH}
        s_sub_u32 {st(7)}, {s('locA')}, {st(6)}
        v_cmp_ne_u32 vcc, {st(7)}, v44
        s_and_b64 {stp(0)}, {stp(0)}, vcc
        s_sub_u32 {st(7)}, {s('locB')}, {st(6)}
        v_cmp_ne_u32 vcc, {st(7)}, v44
        s_and_b32 {st(1)}, {st(1)}, vcc_hi
        """)
    else:
        SO(f"""
        v_cmp_ne_u32 vcc, {s('locA')}, v44
        s_and_b64 {stp(0)}, {stp(0)}, vcc
        v_cmp_ne_u32 vcc, {s('locB')}, v44
        s_and_b32 {st(1)}, {st(1)}, vcc_hi
        """)


def mg_probes():
    """the probes of the two halves: lanes 0..31 <- Q (the proposal, s), lanes 32..63 <- p0[tl + 1] (particle n+1); the steady
    copy leaves v47 = the address of row tl of the row cache (mg_side_sources reads particle n's position through it)"""
    # (the steady copy's tl and i are never negative: one 24-bit multiply-add and the instruction's offset field; the first pass
    # of a run, with tl = -1 or i = -1, is the generic copy's and keeps the scalar arithmetic)
    old = f"""
    s_add_u32 {st(7)}, {s('tl')}, 1
    s_mul_i32 {st(7)}, {st(7)}, 24
    {f"v_add_u32 v44, {st(7)}, v1" if W4 else f"v_mov_b32 v44, {st(7)}"}
    v_mov_b64 {PV[0]}, {sp('Q',0)}
    v_mov_b64 {PV[1]}, {sp('Q',1)}
    v_mov_b64 {PV[2]}, {sp('Q',2)}
    s_mov_b32 exec_lo, 0
    ds_read_b64 {PV[0]}, v44 offset:{LDS_P0}
    ds_read_b64 {PV[1]}, v44 offset:{LDS_P0 + 8}
    ds_read_b64 {PV[2]}, v44 offset:{LDS_P0 + 16}
    s_mov_b32 exec_lo, -1
    """
    if not PEEL:
        E(old)
        return
    G(old)
    SO(f"""
    v_mad_u32_u24 v47, {s('tl')}, 24, {"v1" if W4 else "0"}
    v_mov_b64 {PV[0]}, {sp('Q',0)}
    v_mov_b64 {PV[1]}, {sp('Q',1)}
    v_mov_b64 {PV[2]}, {sp('Q',2)}
    s_mov_b32 exec_lo, 0
    ds_read_b64 {PV[0]}, v47 offset:{LDS_P0 + 24}
    ds_read_b64 {PV[1]}, v47 offset:{LDS_P0 + 32}
    ds_read_b64 {PV[2]}, v47 offset:{LDS_P0 + 40}
    s_mov_b32 exec_lo, -1
    """)


def mg_wall_dz(tag):
    """wdz = signed distance of the lane's probe to its nearer wall, the reference's clamp at / beyond a wall per lane
    (SMC.c:736-739)"""
    E(f"""
    v_add_f64 {vp('T')}, {PV[2]}, {sp('halfLz')}
    v_cmp_ge_f64 vcc, |{PV[2]}|, {sp('halfLz')}
    v_mul_f64 {vp('S6')}, {vp('T')}, {sp('invLz')}
    v_rndne_f64 {vp('S6')}, {vp('S6')}
    v_fma_f64 {vp('wdz')}, -{vp('S6')}, {sp('Lz')}, {vp('T')}
    """)
    clamp = f"""
    s_mov_b64 {stp(0)}, vcc
    v_cmp_ge_f64 vcc, 0, {vp('T')}
    v_mov_b32 {v('T')}, 0xbf1a36e2
    v_mov_b32 {v('T',1)}, 0x3f1a36e2
    s_mov_b64 exec, {stp(0)}
    v_mov_b32 {v('wdz')}, 0xeb1c432d
    v_cndmask_b32 {v('wdz',1)}, {v('T')}, {v('T',1)}, vcc
    s_mov_b64 exec, -1
    """
    if REDIR[0] is None:      # hot path: the clamp lies among the cold pieces
        E(f"s_cbranch_vccnz L_wdzC_{tag}")
        E(f"L_wdz_{tag}:")
        COLD(f"L_wdzC_{tag}:")
        COLD(clamp)
        COLD(f"s_branch L_wdz_{tag}")
    else:                     # already among the cold pieces: in line, jumped over
        E(f"s_cbranch_vccz L_wdz_{tag}")
        E(clamp)
        E(f"L_wdz_{tag}:")


def mg_side_sources(lane_old):
    """the side pair's sources: lane `lane_old` (s) of half B <- particle n's CURRENT position p0[tl], the next lane <- the
    proposal Q; both evaluate against probe B (particle n+1)"""
    (G if PEEL else E)(f"""
    s_mul_i32 {st(7)}, {s('tl')}, 24
    {f"v_add_u32 v47, {st(7)}, v1" if W4 else f"v_mov_b32 v47, {st(7)}"}
    """)
    E(f"""
    s_lshl_b64 exec, 1, {lane_old}
    ds_read_b64 v[{XA_}:{XA_+1}], v47 offset:{LDS_P0}
    ds_read_b64 v[{XA_+2}:{XA_+3}], v47 offset:{LDS_P0 + 8}
    ds_read_b64 v[{XA_+4}:{XA_+5}], v47 offset:{LDS_P0 + 16}
    s_add_u32 {st(7)}, {lane_old}, 1
    s_lshl_b64 exec, 1, {st(7)}
    v_mov_b64 v[{XA_}:{XA_+1}], {sp('Q',0)}
    v_mov_b64 v[{XA_+2}:{XA_+3}], {sp('Q',1)}
    v_mov_b64 v[{XA_+4}:{XA_+5}], {sp('Q',2)}
    s_mov_b64 exec, -1
    """)


def mg_side_capture(sides, lane_old):
    """after round 0: the accumulators of the two side lanes (each holds that one item) go to the side area --
    [old, new][e, fx, fy, fz] -- and are zeroed: the reduction carries probe B WITHOUT the pair (n, n+1)"""
    a = [vp('acc', j) for j in range(4)]
    E(f"""
    {f"s_and_b32 {st(7)}, {s('i')}, 1" if W4 else ""}
    {f"s_lshl_b32 {st(7)}, {st(7)}, 6" if W4 else ""}
    s_mov_b64 exec, {sides}
    v_cmp_ne_u32 vcc, {lane_old}, {LANE}
    s_nop 1
    v_cndmask_b32 v44, 0, 32, vcc
    {f"v_add_u32 v44, {st(7)}, v44" if W4 else ""}
    ds_write_b64 v44, {a[0]} offset:{LDS_SIDEM}
    ds_write_b64 v44, {a[1]} offset:{LDS_SIDEM + 8}
    ds_write_b64 v44, {a[2]} offset:{LDS_SIDEM + 16}
    ds_write_b64 v44, {a[3]} offset:{LDS_SIDEM + 24}
    v_mov_b64 {a[0]}, 0
    v_mov_b64 {a[1]}, 0
    v_mov_b64 {a[2]}, 0
    v_mov_b64 {a[3]}, 0
    s_mov_b64 exec, -1
    """)


def mg_round0(near):
    """round 0 of the merged pass: hand-over of the first candidates, the planes (near: with the wall sites of a probe that
    is within the cutoff of a wall) and the side pair, one fp64 body.  The far form is the hot path: planes on lanes 0 and
    32 (the words of wlp), side pair on the two lanes behind half B's plane; the near form lies among the cold pieces."""
    tag = "mn" if near else "mf"
    if near:
        # wall lanes of either half: the sites and the plane (M2 + 1, the plane last) or the plane alone
        E(f"""
        L_mgN:
        s_add_u32 {st(0)}, {s('M2')}, 1
        {f"s_cmp_eq_u32 {WAVE}, 0" if W4 else ""}
        {f"s_cselect_b32 {st(0)}, {st(0)}, 0" if W4 else ""}
        s_cmp_lg_u32 {s('nearA')}, 0
        s_cselect_b32 {st(2)}, {st(0)}, {s('wlp')}
        s_cmp_lg_u32 {s('nearB')}, 0
        s_cselect_b32 {st(3)}, {st(0)}, {s('wlp',1)}
        s_bcnt1_i32_b32 {st(4)}, {s('sidesHi')}
        s_add_u32 {st(4)}, {st(4)}, {st(3)}
        """)
        startA, startB = st(2), st(4)
        mg_handover(V['wa0'], V['wa1'], sp('hA'), startA, 0, st(5))
    else:
        startA, startB = s('wlp'), s('stB')
        mg_handover(V['wa0'], V['wa1'], sp('hA'), KSTA, 0, st(5))
    mg_handover2("A" + tag, V['wa0'], V['wa1'], sp('hA'), startA, 0, st(5))
    mg_handover(V['wb0'], V['wb1'], sp('hB'), startB if near else KSTB, 128, st(6))
    mg_handover2("B" + tag, V['wb0'], V['wb1'], sp('hB'), startB, 128, st(6))
    mg_probes()
    # this lane's item; the candidates' lanes (st(0): half A's word, st(1): half B's)
    E(f"""
    ds_read_b32 v44, {KL4} offset:{LDS_LIST}
    s_bfm_b64 {stp(0)}, {st(5)}, {startA}
    """)
    if near or not PEEL or W4:
        E(f"""
        s_bfm_b64 {stp(6)}, {st(6)}, {startB}
        s_mov_b32 {st(1)}, {st(6)}
        """)
    else:   # steady copy, one wavefront per replica: half B has its two side lanes in front, so fewer than 32 candidate lanes --
            # a 32-bit field does (with several, the waves without special lanes can have all 32)
        G(f"""
        s_bfm_b64 {stp(6)}, {st(6)}, {startB}
        s_mov_b32 {st(1)}, {st(6)}
        """)
        SO(f"s_bfm_b32 {st(1)}, {st(6)}, {startB}")
    if near:
        # wl = the wall lanes, pl = the planes (last wall lane of a half), the side lanes behind half B's wall lanes
        E(f"""
        s_bfm_b32 {s('hA')}, {st(2)}, 0
        s_bfm_b32 {s('hA',1)}, {st(3)}, 0
        s_sub_u32 {st(7)}, {st(2)}, 1
        s_max_i32 {st(7)}, {st(7)}, 0
        s_lshl_b32 {s('hB')}, {s('wlp')}, {st(7)}
        s_sub_u32 {st(7)}, {st(3)}, 1
        s_max_i32 {st(7)}, {st(7)}, 0
        s_lshl_b32 {s('hB',1)}, {s('wlp',1)}, {st(7)}
        s_cmp_lg_u32 {s('sidesHi')}, 0
        s_cselect_b32 {st(5)}, 3, 0
        s_lshl_b32 {st(5)}, {st(5)}, {st(3)}
        s_mov_b32 {st(4)}, 0
        s_add_u32 {st(3)}, {st(3)}, 32
        """)
        wl, pl, sides, lane_old = sp('hA'), sp('hB'), stp(4), st(3)
    else:
        E(f"""
        s_mov_b32 {st(4)}, 0
        s_mov_b32 {st(5)}, {s('sidesHi')}
        s_add_u32 {st(3)}, {s('wlp',1)}, 32
        """)
        wl, pl, sides, lane_old = sp('wlp'), sp('wlp'), stp(4), st(3)
    # the side pair (a pass that has a proposal AND a next particle)
    (E if near else G)(f"""
    s_cmp_eq_u32 {s('sidesHi')}, 0
    s_cbranch_scc1 L_nss_{tag}
    """)
    mg_side_sources(lane_old)
    E(f"L_nss_{tag}:")
    if not near:   # what does not depend on the LDS reads in flight (list item, probe B, side sources) goes in front of the wait
        mg_coeff_init(G if PEEL else E)      # (the steady copy finds them set)
        for j in range(4):
            E(f"v_mov_b64 {vp('acc', j)}, 0")
    E(f"""
    s_waitcnt lgkmcnt(0)
    v_mul_u32_u24 v45, 24, v44
    {"v_mov_b32 v45, 0" if FAKE else ""}
    s_mov_b64 exec, {stp(0)}
    {f"ds_read2_b64 v[{XA_}:{XA_+3}], v45 offset1:1" if LP else f"global_load_dwordx4 v[{XA_}:{XA_+3}], v45, {SRC}"}
    {f"ds_read_b64 v[{XA_+4}:{XA_+5}], v45 offset:16" if LP else f"global_load_dwordx2 v[{XA_+4}:{XA_+5}], v45, {SRC} offset:16"}
    s_mov_b64 exec, -1
    """)
    mark(None, 21)       # (IVAL: the hand-overs are done, the candidates' fetch is on its way)
    if near:   # table rows of the wall lanes of a near probe: row = lane within its half
        E(f"""
        s_bitcmp1_b32 {s('nearA')}, 0
        s_cselect_b32 {st(6)}, {s('hA')}, 0
        s_bitcmp1_b32 {s('nearB')}, 0
        s_cselect_b32 {st(7)}, {s('hA',1)}, 0
        v_and_b32 v46, 31, {LANE}
        v_lshlrev_b32 v46, 5, v46
        """)
    # displacement of move i+1 in group layout: it travels during the pass and the Metropolis step
    (E if near else G)(f"""
    s_cmp_eq_u32 {s('hasB')}, 0
    s_cbranch_scc1 L_ndd_{tag}
    """)
    (G if PEEL else E)(f"""
    s_add_u32 {st(2)}, {s('i')}, 1
    s_mul_i32 {st(2)}, {st(2)}, 24
    v_add_u32 v45, {st(2)}, {KC}
    global_load_dwordx2 {DdNm}, v45, {sp('dK')}{NT}
    """)
    SO(f"""
    v_mad_u32_u24 v45, {s('i')}, 24, {KC}
    global_load_dwordx2 {DdNm}, v45, {sp('dK')} offset:24{NT}
    """)
    E(f"L_ndd_{tag}:")
    # coefficients: 1 for the candidates, (a0, b0) for a plane, the table's for wall sites
    if near:
        coeff_one(CA_)
        E(f"""
        s_mov_b64 exec, {stp(6)}
        global_load_dwordx4 v[{XA_}:{XA_+3}], v46, {sp('wtab')}
        global_load_dwordx4 v[{CA_}:{CA_+3}], v46, {sp('wtab')} offset:16
        s_andn2_b64 exec, {pl}, {stp(6)}
        v_mov_b64 v[{CA_}:{CA_+1}], {sp('a0s')}
        v_mov_b64 v[{CA_+2}:{CA_+3}], {sp('b0s')}
        s_mov_b64 exec, -1
        """)
        for j in range(4):
            E(f"v_mov_b64 {vp('acc', j)}, 0")
    if XC:   # steady copy: a lane whose item is particle n's cell (either half) or particle n+1's (half B) has no item
        mg_exclude()
    # everything that has an item: candidates, wall lanes, side lanes
    E(f"""
    s_or_b64 {stp(6)}, {stp(0)}, {wl}
    s_or_b64 {stp(6)}, {stp(6)}, {sides}
    """)
    mg_wall_dz(tag)
    if PS and not near:
        E(f"s_mov_b32 {s('hB', 1)}, 0")                     # psv (both copies: the hand-over used the pair)
        pre_screen_b()
    if near:
        E("s_waitcnt vmcnt(0) lgkmcnt(0)")
    elif LP:
        E("s_waitcnt lgkmcnt(0)")
    else:   # the candidates' positions; the displacement asked for behind them may still travel
        G("s_waitcnt vmcnt(0)")
        SO("s_waitcnt vmcnt(1)")
    mark(None, 22)       # (IVAL: the candidates' positions have arrived)
    if "nobody" in ABL and not near:
        E(f"s_branch L_nolj_{tag}")
    body(tag, PV, XA_, CA_, stp(6), True, wl, pl)
    (E if near else G)(f"""
    s_cmp_eq_u32 {s('sidesHi')}, 0
    s_cbranch_scc1 L_nsc_{tag}
    """)
    mg_side_capture(sides, lane_old)
    E(f"L_nsc_{tag}:")
    if near:
        mg_coeff_init(SO)     # (the wall lanes held the table's coefficients)
        if PS:
            E(f"s_mov_b32 {s('hB', 1)}, 0")                 # psv: no pre-screen in this form (hB was the planes' mask)
        E("s_branch L_mgR0")


def pre_screen_b():
    """PS (steady copy, far form): probe B's screen of the next move while the candidates' positions travel.  Free here: st(0..2) (the
    candidates' mask went into stp(6)), the hand-over's scratch pairs hA / hB, vcc, the body's temporaries v36..v40; live: st(3..7),
    v14..v35, v44..v49.  wb0 / wb1 are empty unless further rounds are pending (bit 1 of nearB)."""
    save, FORCE_TAG[0] = FORCE_TAG[0], " @S"
    psv, nxt, R = s('hB', 1), s('hB'), dict(t0=st(0), t1=st(1), t4=st(2), t5=s('hB'), p2=(sp('hA'), s('hA'), s('hA', 1)),
                                             pxy="v40", tb=36)
    E(f"""
    s_bitcmp1_b32 {s('nearB')}, 1
    s_cbranch_scc1 L_psx
    s_cmp_eq_u32 {s('tl')}, 63
    s_cbranch_scc1 L_psx
    s_add_u32 {st(0)}, {s('tl')}, 1
    s_nop 3
    v_readlane_b32 {st(1)}, {v('rzl')}, {st(0)}
    v_readlane_b32 {nxt}, {v('rxy')}, {st(0)}
    s_bitcmp1_b32 {st(1)}, 16
    s_cbranch_scc1 L_psx
    """)
    screen_ranged8("P", nxt, v('wb0'), v('wb1'), R)
    E(f"""
    v_or_b32 {v('wb0')}, {v('wb0')}, {v('uns0')}
    v_or_b32 {v('wb1')}, {v('wb1')}, {v('uns1')}
    s_mov_b32 {psv}, 1
    L_psx:
    """)
    FORCE_TAG[0] = save


def mg_more():
    """further rounds (cold): the candidates still in the flag words, 32 per half and round"""
    E(f"""
    L_mgMore:
    """)
    if Z8C:   # executed work: rounds beyond the first of a pass
        cnt_addr("v46")
        E(f"""
        v_mov_b32 v44, 1
        s_mov_b64 exec, 1
        ds_add_u32 v46, v44 offset:{LDS_CNT + 20}
        s_mov_b64 exec, -1
        """)
    mg_handover(V['wa0'], V['wa1'], sp('hA'), "0", 0, st(5))
    mg_handover(V['wb0'], V['wb1'], sp('hB'), "0", 128, st(6))
    E(f"""
    s_bitset0_b32 {s('nearB')}, 1
    v_or3_b32 v44, {v('wa0')}, {v('wa1')}, {v('wb0')}
    v_or_b32 v44, v44, {v('wb1')}
    v_cmp_ne_u32 vcc, 0, v44
    s_cbranch_vccz L_mgM1
    s_bitset1_b32 {s('nearB')}, 1
    L_mgM1:
    ds_read_b32 v44, {KL4} offset:{LDS_LIST}
    s_bfm_b64 {stp(0)}, {st(5)}, 0
    s_bfm_b64 {stp(6)}, {st(6)}, 0
    s_mov_b32 {st(1)}, {st(6)}
    s_waitcnt lgkmcnt(0)
    v_mul_u32_u24 v45, 24, v44
    s_mov_b64 exec, {stp(0)}
    {f"ds_read2_b64 v[{XA_}:{XA_+3}], v45 offset1:1" if LP else f"global_load_dwordx4 v[{XA_}:{XA_+3}], v45, {SRC}"}
    {f"ds_read_b64 v[{XA_+4}:{XA_+5}], v45 offset:16" if LP else f"global_load_dwordx2 v[{XA_+4}:{XA_+5}], v45, {SRC} offset:16"}
    s_mov_b64 exec, -1
    """)
    coeff_one(CA_)
    if XC:
        mg_exclude()
    E("s_waitcnt vmcnt(0) lgkmcnt(0)")
    body("mm", PV, XA_, CA_, stp(0), False)
    mg_coeff_init(SO)         # (the plane lanes worked with 1, 1)
    if PS:
        E(f"s_mov_b32 {s('hB', 1)}, 0")                     # psv: the rounds' hand-overs used the pair (and no pre-screen ran)
    E("s_branch L_mgR0")


def mg_reduce():
    """the four accumulators of the two halves -> 8 sums in group layout (v[50:51]): row 0 of a half = [e | fy], row 1 =
    [fx | fz]; then v[50:51] = probe A's sums in both halves (Fn), MGW = probe B's in both halves"""
    a = [V['acc'] + 2 * j for j in range(4)]
    E(f"""
    s_nop 1
    v_permlane16_swap_b32 v{a[0]}, v{a[1]}
    v_permlane16_swap_b32 v{a[0]+1}, v{a[1]+1}
    v_permlane16_swap_b32 v{a[2]}, v{a[3]}
    v_permlane16_swap_b32 v{a[2]+1}, v{a[3]+1}
    s_nop 0
    v_add_f64 v[{a[0]}:{a[0]+1}], v[{a[0]}:{a[0]+1}], v[{a[1]}:{a[1]+1}]
    v_add_f64 v[{a[2]}:{a[2]+1}], v[{a[2]}:{a[2]+1}], v[{a[3]}:{a[3]+1}]
    s_nop 1
    v_mov_b32_dpp v44, v{a[0]} row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1
    v_mov_b32_dpp v45, v{a[0]+1} row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1
    v_mov_b32_dpp v46, v{a[2]} row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1
    v_mov_b32_dpp v47, v{a[2]+1} row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1
    v_add_f64 v[{a[0]}:{a[0]+1}], v[{a[0]}:{a[0]+1}], v[44:45]
    v_add_f64 v[{a[2]}:{a[2]+1}], v[{a[2]}:{a[2]+1}], v[46:47]
    s_nop 1
    v_mov_b32_dpp v{a[0]}, v{a[2]} quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xc
    v_mov_b32_dpp v{a[0]+1}, v{a[2]+1} quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xc
    """)
    dst = f"v[{a[0]}:{a[0]+1}]"
    for ctrl in ("row_half_mirror", "quad_perm:[2,3,0,1]", "quad_perm:[1,0,3,2]"):
        E(f"""
        s_nop 1
        v_mov_b32_dpp v44, v{a[0]} {ctrl} row_mask:0xf bank_mask:0xf bound_ctrl:1
        v_mov_b32_dpp v45, v{a[0]+1} {ctrl} row_mask:0xf bank_mask:0xf bound_ctrl:1
        v_add_f64 {dst}, {dst}, v[44:45]
        """)
    if W4:
        # the ONE exchange of a move: every wave writes its 8 partial sums, s_barrier, every wave adds the waves' partials in
        # wave order -- bit-identical totals on all waves, so all take the same Metropolis decision.  Two buffers alternate
        # with the parity of the move counter: a wave already in the next move cannot overwrite what a slower one still reads.
        E(f"""
        s_and_b32 {st(1)}, {s('i')}, 1
        s_mul_i32 {st(1)}, {st(1)}, {WPR * 512}
        s_lshl_b32 {st(0)}, {WAVE}, 9
        s_add_u32 {st(0)}, {st(0)}, {st(1)}
        v_lshl_add_u32 v44, {LANE}, 3, {st(0)}
        v_lshl_add_u32 v45, {LANE}, 3, {st(1)}
        ds_write_b64 v44, {dst} offset:{LDS_X}
        s_waitcnt lgkmcnt(0)
        s_barrier
        """)
        for k in range(0, WPR, 4):           # four partial sums at a time in v36..v43
            for j in range(4):
                E(f"ds_read_b64 v[{36 + 2 * j}:{37 + 2 * j}], v45 offset:{LDS_X + 512 * (k + j)}")
            E("s_waitcnt lgkmcnt(0)")
            if k == 0:
                E(f"v_add_f64 {dst}, v[36:37], v[38:39]")
            else:
                E(f"v_add_f64 {dst}, {dst}, v[36:37]")
                E(f"v_add_f64 {dst}, {dst}, v[38:39]")
            E(f"v_add_f64 {dst}, {dst}, v[40:41]")
            E(f"v_add_f64 {dst}, {dst}, v[42:43]")
    E(f"""
    v_mov_b32 v20, v{a[0]}
    v_mov_b32 v21, v{a[0]+1}
    s_nop 1
    v_permlane32_swap_b32 v{a[0]}, v20
    v_permlane32_swap_b32 v{a[0]+1}, v21
    """)
    return dst


if MG:
    # log-uniform of move i+1 (scalar load into nxy:nzl, whose run-start values the first pass has consumed by now): it
    # travels during the whole pass; lu itself is still this move's until the decision
    G(f"""
    s_cmp_eq_u32 {s('hasB')}, 0
    s_cbranch_scc1 L_nolu
    """)
    (G if (T5 and PEEL) else E)(f"""
    s_add_u32 {st(1)}, {s('i')}, 1
    s_lshl_b32 {st(0)}, {st(1)}, 3
    s_load_dwordx2 {sp('nxy')}, {sp('uK')}, {st(0)}
    """)
    if T5 and PEEL:   # steady copy (i >= 0): register + immediate offset
        SO(f"""
        s_lshl_b32 {st(0)}, {s('i')}, 3
        s_load_dwordx2 {sp('nxy')}, {sp('uK')}, {st(0)} offset:8
        """)
    E(f"""
    L_nolu:
    s_or_b32 {st(0)}, {s('nearA')}, {s('nearB')}
    s_cbranch_scc1 L_mgN
    """)
    mg_round0(False)
    REDIR[0] = cold
    mg_round0(True)
    REDIR[0] = None
    E(f"""
    L_mgR0:
    s_bitcmp1_b32 {s('nearB')}, 1
    s_cbranch_scc1 L_mgMore
    """)
    mark(None, 23)       # (IVAL: the fp64 bodies of all rounds and the side capture are done)
    REDIR[0] = cold
    mg_more()
    REDIR[0] = None
    FnG = mg_reduce()
    mark(None, 24)       # (IVAL: the eight sums are reduced)
    if ELDS:
        SO(f"""
        v_mad_u32_u24 v49, {s('tl')}, 24, {KC}
        ds_read_b64 v[36:37], v49 offset:{LDS_P0 + 24}
        ds_read_b64 v[40:41], {KSD} offset:{LDS_SIDEM}
        ds_read_b64 v[38:39], {KSD} offset:{LDS_SIDEM + 32}
        """)
    # ---- Metropolis step in group layout (SMC.c:326-335); FmV, DdV: this move's Fm and displacement per group
    E(f"s_mov_b32 {s('accf')}, 0")
    G(f"""
    s_cmp_eq_u32 {s('hasA')}, 0
    s_cbranch_scc1 L_noA
    """)
    E(f"""
    v_add_f64 {vp('D',0)}, {FnG}, -{vp('FmV')}
    v_add_f64 {vp('D',1)}, {FnG}, {vp('FmV')}
    v_fma_f64 {vp('D',2)}, {vp('FmV')}, {sp('AoT')}, {vp('DdV')}
    v_mul_f64 {vp('D',2)}, {vp('D',2)}, 0.5
    v_fma_f64 {vp('D',2)}, {vp('D',0)}, {sp('Ao4T')}, {vp('D',2)}
    v_mul_f64 {vp('D',2)}, {vp('D',2)}, {vp('D',1)}
    s_mov_b64 exec, 0xff
    v_mul_f64 {vp('D',2)}, {vp('D',0)}, 4.0
    s_mov_b64 exec, -1
    s_nop 0
    v_mov_b32_dpp {v('T')}, {v('D',4)} row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1
    v_mov_b32_dpp {v('T',1)}, {v('D',5)} row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1
    v_add_f64 {vp('D',2)}, {vp('D',2)}, {vp('T')}
    v_mov_b32 {v('T')}, 0
    v_mov_b32 {v('T',1)}, 0
    s_nop 0
    v_mov_b32_dpp {v('T')}, {v('D',4)} row_bcast:15 row_mask:0xa bank_mask:0xf
    v_mov_b32_dpp {v('T',1)}, {v('D',5)} row_bcast:15 row_mask:0xa bank_mask:0xf
    v_add_f64 {vp('D',2)}, {vp('D',2)}, {vp('T')}
    v_mul_f64 {vp('D',2)}, {vp('D',2)}, -{sp('invT')}
    {"s_waitcnt vmcnt(0)" if (EVM and MG and PEEL and not W4) else ""}
    v_cmp_lt_f64 vcc, {sp('lu')}, {vp('D',2)}
    s_nop 0
    s_bitcmp1_b32 vcc_lo, 31
    {"s_branch" if "noaccept" in ABL else "s_cbranch_scc0"} L_reject
    """)
MUTE[0] = MG

# ---------------------------------------------------------------------------------------------- probe A + Metropolis
QP = [sp('Q', 0), sp('Q', 1), sp('Q', 2)]
BP = ["v[14:15]", "v[16:17]", "v[18:19]"]
FnV = vp('M', 0)          # v[36:37]: the body's M registers are free now
FbV = vp('dr2')           # z8t: v[40:41] = probe B's total without the side pair, from the exchange to the proposal
DdN = vp('M', 1)          # z8t: v[38:39] = displacement of move i+1 per row, loaded during the exchange


def side_sources():
    """the side pair's sources on the side lanes: particle n = p0[tl] (ma, zb: where the move left it, read after the
    decision; z8t: its CURRENT position, read before the decision, and the proposal Q on the second lane)"""
    GW(f"""
    s_cmp_eq_u32 {s_hasAw()}, 0
    s_cbranch_scc1 L_nosrc
    """)
    E(f"""
    s_mul_i32 {st(0)}, {s('tl')}, 24
    {f"v_add_u32 {v('T')}, {st(0)}, v1" if W4 else f"v_mov_b32 {v('T')}, {st(0)}"}
    {f"s_lshl_b64 {stp(2)}, 1, {s('sideL')}" if ZB else ""}
    s_mov_b64 exec, {stp(2) if ZB else sp('sideM')}
    ds_read_b64 v[{XB_}:{XB_+1}], {v('T')} offset:{LDS_P0}
    ds_read_b64 v[{XB_+2}:{XB_+3}], {v('T')} offset:{LDS_P0 + 8}
    ds_read_b64 v[{XB_+4}:{XB_+5}], {v('T')} offset:{LDS_P0 + 16}
    """)
    if TT:
        E(f"s_lshl_b64 exec, 1, {s('sideN')}")
        for j in range(6):
            E(f"v_mov_b32 v{XB_ + j}, {s('Q', j)}")
    E(f"""
    s_mov_b64 exec, -1
    L_nosrc:
    """)


if TT:
    # team A evaluates probe A, team B probe B (with the side pair for both outcomes, kept apart): both leave their
    # partial sums in FnV's registers; then the one exchange of the move
    E(f"""
    v_mov_b32 v{V['M']}, 0
    v_mov_b32 v{V['M'] + 1}, 0
    s_cmp_ge_u32 {WAVE}, {KS}
    s_cbranch_scc1 L_evalB
    """)
    G(f"""
    s_cmp_eq_u32 {s('hasA')}, 0
    s_cbranch_scc1 L_xchg
    """)
    mark(2)
    if TTP == "allA":
        E(f"s_setprio {TTPA}")
    probe("A", QP, True, (s('Q', 4), s('Q', 5)), V['wa0'], V['wa1'], XA_, CA_, sp('haveA'), False,
          "s_waitcnt lgkmcnt(0)" if LP else "s_waitcnt vmcnt(0)")
    mark(4)
    reduce4(FnV)
    if TTP == "allA":
        E("s_setprio 0")
    mark(11)
    E(f"""
    s_branch L_xchg
    L_evalB:
    """)
    G(f"""
    s_cmp_eq_u32 {s('hasB')}, 0
    s_cbranch_scc1 L_xchg
    """)
    mark(2)
    side_sources()
    E("s_waitcnt lgkmcnt(0)")
    probe("B", BP, False, "v[18:19]", V['wb0'], V['wb1'], XB_, CB_, sp('haveB'), True,
          "s_waitcnt lgkmcnt(0)" if LP else "s_waitcnt vmcnt(0)", sp('wallB'), sp('planeB'))
    mark(4)
    if TTP == "probe":
        E("s_setprio 0")
    reduce4(FnV)
    if TTP == "red":
        E("s_setprio 0")
    mark(11)
    E("L_xchg:")
    # displacement of move i+1 per row (rows 1..3 -> components 0, 8, 16 of displ[3 (i+1) ..]): asked for now, it
    # travels during the exchange and the Metropolis step (DdV itself is still this move's)
    G(f"""
    s_cmp_eq_u32 {s('hasB')}, 0
    s_cbranch_scc1 L_nodd
    """)
    # (one v_mad_u32_u24 + the offset field instead of the scalar arithmetic, as in the merged pass, measured here: config 5
    # 22.15 -> 22.51 ms per sweep -- a VALU result that the next memory instruction needs costs this kernel's lone wavefronts
    # more than two scalar instructions do; profiles/r04_address_arithmetic_ab.txt)
    E(f"""
    s_add_u32 {st(1)}, {s('i')}, 1
    s_mul_i32 {st(1)}, {st(1)}, 24
    v_add_u32 {v('S6')}, {st(1)}, {KPROW}
    global_load_dwordx2 {DdN}, {v('S6')}, {sp('dK')}{NT}
    L_nodd:
    // what the part after the exchange reads from LDS, per lane: component (row - 1) of particle n+1's position (row cache)
    // and the side pair's result [row][rejected | accepted] of this move's buffer
    s_add_u32 {st(0)}, {s('tl')}, 1
    s_mul_i32 {st(0)}, {st(0)}, 24
    v_add3_u32 v22, {st(0)}, {KPROW}, v1
    {"" if TL else f"s_and_b32 {st(1)}, {s('i')}, 1"}
    {"" if TL else f"s_lshl_b32 {st(1)}, {st(1)}, 6"}
    {"" if TL else f"v_add_u32 v23, {st(1)}, {KSIDE}"}
    v_mov_b32 v14, 0
    v_mov_b32 v15, 0
    """)
    TG = " @G" if PEEL else ""
    xchg2(FnV, FnV, FbV, f"""
    ds_read_b64 v[16:17], v22 offset:{LDS_P0}
    s_cmp_lg_u32 {s('hasA')}, 0{TG}
    s_cselect_b64 exec, -1, 0{TG}
    ds_read_b64 v[14:15], {VSR if TL else "v23"} offset:{LDS_SIDE}
    s_mov_b64 exec, -1{TG}
    """)
    # this move's dX = Fm A/T + displ (D2; harmless without a move to decide), then the displacement of move i+1 (asked
    # for before the exchange) takes DdV's place NOW, before the accepted move's stores: vmcnt counts loads and stores in
    # issue order, and a wait further down would cover those stores as well (config 5: 28.3 -> 27.5 ms per sweep; config 2,
    # where half the moves accept, unchanged: the stores' acknowledgements were not what it waited for)
    E(f"""
    v_fma_f64 {vp('D',2)}, {vp('FmV')}, {sp('AoT')}, {vp('DdV')}
    s_waitcnt vmcnt(0)
    v_mov_b32 {v('DdV')}, v{V['M'] + 2}
    v_mov_b32 {v('DdV',1)}, v{V['M'] + 3}
    s_mov_b64 {stp(0)}, 0xffff
    """)
    # Two chains that do not depend on each other, issued alternately (a wavefront with its SIMD almost to itself waits
    # out the latency of every dependent instruction): M = the Metropolis argument of this move (SMC.c:326-335; row 0:
    # 4 (eA - eB), by selection instead of an exec mask), P = the proposal of particle n+1 (SMC.c:307-316) for both
    # outcomes: Fm = team B's total + the side result of the lane's half, q = p + (Fm A/T + displ), wrap of rows 1-2 by the
    # per-row constants (0 in rows 0 and 3: no wrap), fixed point; the unsafe-z test of every lane goes to planeM's pair (list hand-over: planeM is
    # a constant of the launch there, so haveB's pair, dead once the probes are done)
    Mc = f"""
    v_add_f64 {vp('D',0)}, {FnV}, -{vp('FmV')}
    v_add_f64 {vp('D',1)}, {FnV}, {vp('FmV')}
    v_mul_f64 {vp('D',2)}, {vp('D',2)}, 0.5
    v_fma_f64 {vp('D',2)}, {vp('D',0)}, {sp('Ao4T')}, {vp('D',2)}
    v_mul_f64 {vp('D',2)}, {vp('D',2)}, {vp('D',1)}
    v_mul_f64 {vp('T')}, {vp('D',0)}, 4.0
    v_cndmask_b32 {v('D',4)}, {v('D',4)}, {v('T')}, {stp(0)}
    v_cndmask_b32 {v('D',5)}, {v('D',5)}, {v('T',1)}, {stp(0)}
    v_mov_b32 {v('T')}, 0
    v_mov_b32 {v('T',1)}, 0
    v_mov_b32_dpp {v('T')}, {v('D',4)} row_bcast:15 row_mask:0xa bank_mask:0xf
    v_mov_b32_dpp {v('T',1)}, {v('D',5)} row_bcast:15 row_mask:0xa bank_mask:0xf
    v_add_f64 {vp('D',2)}, {vp('D',2)}, {vp('T')}
    s_nop 1
    v_mov_b32_dpp {v('T')}, {v('D',4)} row_bcast:31 row_mask:0xc bank_mask:0xf
    v_mov_b32_dpp {v('T',1)}, {v('D',5)} row_bcast:31 row_mask:0xc bank_mask:0xf
    v_add_f64 {vp('D',2)}, {vp('D',2)}, {vp('T')}
    v_mul_f64 {vp('D',2)}, {vp('D',2)}, -{sp('invT')}
    """
    Pc = f"""
    v_add_f64 v[14:15], {FbV}, v[14:15]
    v_fma_f64 v[18:19], v[14:15], {sp('AoT')}, {vp('DdV')}
    v_add_f64 v[16:17], v[16:17], v[18:19]
    v_mul_f64 v[18:19], v[16:17], {KINV}
    v_rndne_f64 v[18:19], v[18:19]
    v_fma_f64 v[16:17], -v[18:19], {sp('L')}, v[16:17]
    v_mul_f64 v[18:19], v[16:17], {KFIX}
    v_cmp_nlt_f64 {sp('haveB' if TL else 'planeM')}, |v[16:17]|, {sp('zsafe')}
    v_rndne_f64 v[18:19], v[18:19]
    v_mov_b32 v24, 0x7fff
    v_mov_b32 v25, 0xffff8001
    v_cvt_i32_f64 v20, v[18:19]
    v_med3_i32 v21, v20, v24, v25
    """
    Ml = [l.strip() for l in Mc.strip().split("\n") if l.strip()]
    Pl = [l.strip() for l in Pc.strip().split("\n") if l.strip()]
    ILV = os.environ.get("SMCX_GEN_NO_ILV") != "1"
    if ILV:
        k = 0
        for j, m in enumerate(Ml):
            E(m)
            if not m.startswith("s_nop") and k < len(Pl):
                E(Pl[k]); k += 1
        for q in Pl[k:]:
            E(q)
    else:
        for q in Pl + Ml:
            E(q)
    E(f"s_mov_b32 {s('accf')}, 0")
    G(f"""
    s_cmp_eq_u32 {s('hasA')}, 0
    s_cbranch_scc1 L_noA
    """)
    E(f"v_cmp_lt_f64 vcc, {sp('lu')}, {vp('D',2)}")
    mark(7)
    E(f"""
    s_bitcmp1_b32 vcc_hi, 31
    s_cbranch_scc0 L_reject
    """)
else:
    G(f"s_cmp_eq_u32 {s('hasA')}, 0")
    G("s_cbranch_scc1 L_noA")
    probe("A", QP, True, (s('Q', 4), s('Q', 5)), V['wa0'], V['wa1'], XA_, CA_, sp('haveA'), False,
          ("" if (ZBC or Z8C) else "s_waitcnt vmcnt(4)") if ZB else "s_nop 0")
    reduce4(FnV)
    if W4:
        xchg(FnV, 0)
(E if not TT else (lambda t: None))(f"""
// ---- Metropolis step in row layout (SMC.c:326-335); DdV = displacement of this move per row (z8t: above, interleaved)
v_add_f64 {vp('D',0)}, {FnV}, -{vp('FmV')}
v_add_f64 {vp('D',1)}, {FnV}, {vp('FmV')}
v_fma_f64 {vp('D',2)}, {vp('FmV')}, {sp('AoT')}, {vp('DdV')}
v_mul_f64 {vp('D',2)}, {vp('D',2)}, 0.5
v_fma_f64 {vp('D',2)}, {vp('D',0)}, {sp('Ao4T')}, {vp('D',2)}
v_mul_f64 {vp('D',2)}, {vp('D',2)}, {vp('D',1)}
{"s_mov_b64 exec, 0xffff" if ZB else f"s_mov_b32 {st(0)}, 0xffff"}
{"" if ZB else f"s_mov_b32 {st(1)}, 0"}
{"" if ZB else f"s_mov_b64 exec, {stp(0)}"}
v_mul_f64 {vp('D',2)}, {vp('D',0)}, 4.0
s_mov_b64 exec, -1
""")
if TT:
    pass
elif ZB:
    # sum over the four rows into row 3 with the row broadcasts of DPP (the rows are uniform after reduce4):
    # rows 1, 3 += lane 15 of rows 0, 2; then rows 2, 3 += lane 31; the decision is row 3's
    E(f"""
    v_mov_b32 {v('T')}, 0
    v_mov_b32 {v('T',1)}, 0
    v_mov_b32_dpp {v('T')}, {v('D',4)} row_bcast:15 row_mask:0xa bank_mask:0xf
    v_mov_b32_dpp {v('T',1)}, {v('D',5)} row_bcast:15 row_mask:0xa bank_mask:0xf
    v_add_f64 {vp('D',2)}, {vp('D',2)}, {vp('T')}
    s_nop 1
    v_mov_b32_dpp {v('T')}, {v('D',4)} row_bcast:31 row_mask:0xc bank_mask:0xf
    v_mov_b32_dpp {v('T',1)}, {v('D',5)} row_bcast:31 row_mask:0xc bank_mask:0xf
    v_add_f64 {vp('D',2)}, {vp('D',2)}, {vp('T')}
    v_mul_f64 {vp('D',2)}, {vp('D',2)}, -{sp('invT')}
    v_cmp_lt_f64 vcc, {sp('lu')}, {vp('D',2)}
    """)
    mark(7)
    E(f"""
    s_bitcmp1_b32 vcc_hi, 31
    s_cbranch_scc0 L_reject
    """)
else:
    E(f"""
    // sum over the four rows: lanes ^32, then rows ^1
    v_mov_b32 {v('T')}, {v('D',4)}
    v_mov_b32 {v('T',1)}, {v('D',5)}
    s_nop 1
    v_permlane32_swap_b32 {v('D',4)}, {v('T')}
    v_permlane32_swap_b32 {v('D',5)}, {v('T',1)}
    s_nop 0
    v_add_f64 {vp('D',2)}, {vp('D',2)}, {vp('T')}
    s_nop 0
    v_mov_b32 {v('T')}, {v('D',4)}
    v_mov_b32 {v('T',1)}, {v('D',5)}
    s_nop 1
    v_permlane16_swap_b32 {v('D',4)}, {v('T')}
    v_permlane16_swap_b32 {v('D',5)}, {v('T',1)}
    s_nop 0
    v_add_f64 {vp('D',2)}, {vp('D',2)}, {vp('T')}
    v_mul_f64 {vp('D',2)}, {vp('D',2)}, -{sp('invT')}
    v_cmp_lt_f64 vcc, {sp('lu')}, {vp('D',2)}
    s_cmp_lg_u64 vcc, 0
    s_cbranch_scc0 L_reject
    """)
MUTE[0] = False
if ELDS:   # the side result of an accepted move takes the place of the rejected move's (both were asked for before the decision)
    SO("""
    s_waitcnt lgkmcnt(0)
    v_mov_b64 v[40:41], v[38:39]
    """)
E(f"""
// accepted: E += Un - Um = 4 (eA - eB) (row 0 of g; mg: the e group, lane 0 either way), particle n takes the proposal
{f"s_mov_b32 {s('accf')}, 1" if (TT or MG) else ""}
{"" if T5 else f"v_readlane_b32 {st(0)}, {v('D',0)}, 0"}
{"" if T5 else f"v_readlane_b32 {st(1)}, {v('D',1)}, 0"}
{f"v_fma_f64 {vp('T')}, {vp('D',0)}, 4.0, {sp('E')}" if T5 else ""}
s_add_u32 {s('jacc')}, {s('jacc')}, 1
{"" if T5 else "s_nop 0"}
{"" if T5 else f"v_mov_b32 {v('T')}, {st(0)}"}
{"" if T5 else f"v_mov_b32 {v('T',1)}, {st(1)}"}
{"" if T5 else f"v_fma_f64 {vp('T')}, {vp('T')}, 4.0, {sp('E')}"}
s_lshl_b64 {stp(0)}, 1, {s('tl')}
s_add_u32 {st(2)}, {s('first')}, {s('i')}
{"" if ZB else f"s_mul_i32 {st(2)}, {st(2)}, 24"}
v_readfirstlane_b32 {s('E')}, {v('T')}
v_readfirstlane_b32 {s('E',1)}, {v('T',1)}
""")
if not ZB:
    E(f"""
    s_mov_b64 exec, {stp(0)}
    v_mov_b32 {xy(0)}, {v('axy')}
    v_and_b32 {v('uns0')}, -2, {v('uns0')}
    v_or_b32 {v('uns0')}, {s('ua')}, {v('uns0')}
    v_mov_b32 {v('T')}, {s('az16')}
    v_mov_b32 v50, {s('Q',0)}
    v_mov_b32 v51, {s('Q',1)}
    v_mov_b32 v52, {s('Q',2)}
    v_mov_b32 v53, {s('Q',3)}
    v_mov_b32 v54, {s('Q',4)}
    v_mov_b32 v55, {s('Q',5)}
    v_mov_b32 {v('T',1)}, {st(2)}
    v_mul_u32_u24 {v('S6')}, 24, {LANE}
    ds_write_b16 {v('zaddr')}, {v('T')}
    global_store_dwordx4 {v('T',1)}, v[50:53], {sp('Rg')}{NT}
    global_store_dwordx2 {v('T',1)}, v[54:55], {sp('Rg')} offset:16{NT}
    ds_write_b64 {v('S6')}, v[50:51] offset:{LDS_P0}
    ds_write_b64 {v('S6')}, v[52:53] offset:{LDS_P0 + 8}
    ds_write_b64 {v('S6')}, v[54:55] offset:{LDS_P0 + 16}
    s_mov_b64 exec, -1
    s_nop 1
    """)
else:
    # lane tl: the fp64 position to R (particle order), Rs (cell order) and the row cache; then the owner lane of
    # the particle's cell: packed x,y (indexed register write), unsafe bit, int16 z; then lane g: the group's range
    E(f"""
    {f"s_and_b32 {st(3)}, {s('locA')}, {NS * 64 - 1}" if W4 else ""}
    s_mov_b64 exec, {stp(0)}
    {f"v_mov_b64 v[50:51], {sp('Q',0)}" if T5 else f"v_mov_b32 v50, {s('Q',0)}"}
    {f"v_mov_b64 v[52:53], {sp('Q',1)}" if T5 else f"v_mov_b32 v51, {s('Q',1)}"}
    {f"v_mov_b64 v[54:55], {sp('Q',2)}" if T5 else f"v_mov_b32 v52, {s('Q',2)}"}
    {"" if T5 else f"v_mov_b32 v53, {s('Q',3)}"}
    {"" if T5 else f"v_mov_b32 v54, {s('Q',4)}"}
    {"" if T5 else f"v_mov_b32 v55, {s('Q',5)}"}
    v_mad_u32_u24 {v('T',1)}, {st(2)}, 24, 0
    v_mad_u32_u24 {v('T')}, {st(3) if W4 else s('locA')}, 24, 0
    {f"v_mad_u32_u24 {v('S6')}, {LANE}, 24, v1" if W4 else f"v_mul_u32_u24 {v('S6')}, 24, {LANE}"}
    {"" if "nostoreR" in ABL else f"global_store_dwordx4 {v('T',1)}, v[50:53], {sp('Rg')}{NT}"}
    {"" if "nostoreR" in ABL else f"global_store_dwordx2 {v('T',1)}, v[54:55], {sp('Rg')} offset:16{NT}"}
    {"" if "nop0" in ABL else f"ds_write_b64 {v('S6')}, v[50:51] offset:{LDS_P0}"}
    {"" if "nop0" in ABL else f"ds_write_b64 {v('S6')}, v[52:53] offset:{LDS_P0 + 8}"}
    {"" if "nop0" in ABL else f"ds_write_b64 {v('S6')}, v[54:55] offset:{LDS_P0 + 16}"}
    """)
    if W4:   # the cell, its copy in Rs and its group's range belong to one wave
        E(f"""
        s_lshr_b32 {st(1)}, {s('locA')}, {WSH}
        {f"s_and_b32 {st(4)}, {WAVE}, {KS - 1}" if TT else ""}
        s_cmp_lg_u32 {st(1)}, {st(4) if TT else WAVE}
        s_cbranch_scc1 L_notmine
        """)
    E(f"""
    {"" if "nostoreRs" in ABL else f"global_store_dwordx4 {v('T')}, v[50:53], {sp('Rs')}"}
    {"" if "nostoreRs" in ABL else f"global_store_dwordx2 {v('T')}, v[54:55], {sp('Rs')} offset:16"}
    {f"ds_write_b64 {v('T')}, v[50:51] offset:{LDS_RS}" if LP else ""}
    {f"ds_write_b64 {v('T')}, v[52:53] offset:{LDS_RS + 8}" if LP else ""}
    {f"ds_write_b64 {v('T')}, v[54:55] offset:{LDS_RS + 16}" if LP else ""}
    {f"s_bfe_u32 {st(1)}, {s('locA')}, {SLOTF}" if W4 else f"s_lshr_b32 {st(1)}, {s('locA')}, 6"}
    s_lshl_b64 {stp(2)}, 1, {s('locA')}
    """)
    uns = f"""
    s_lshl_b64 {stp(4)}, 1, {st(1)}
    s_not_b64 {stp(6)}, {stp(4)}
    s_cmp_eq_u32 {s('ua')}, 0
    s_cselect_b64 {stp(4)}, 0, {stp(4)}
    s_mov_b64 exec, {stp(2)}
    v_mov_b32 {v('T')}, {s('axys')}
    v_and_b32 {v('uns0')}, {st(6)}, {v('uns0')}
    v_and_b32 {v('uns1')}, {st(7)}, {v('uns1')}
    {"s_mov_b32 s101, m0" if (IVAL and not TT) else ""}
    {"" if "nogpridx" in ABL else f"s_set_gpr_idx_on {st(1)}, gpr_idx(DST)"}
    {"" if "nogpridx" in ABL else f"v_mov_b32 {xy(0)}, {v('T')}"}
    {"" if "nogpridx" in ABL else "s_set_gpr_idx_off"}
    {"s_mov_b32 m0, s101" if (IVAL and not TT) else ""}
    v_or_b32 {v('uns0')}, {st(4)}, {v('uns0')}
    v_or_b32 {v('uns1')}, {st(5)}, {v('uns1')}
    """
    if T5:   # the unsafe-z bit of the cell changes only if the proposal is unsafe or some cell already is: cold piece
        E(f"""
        s_or_b32 {st(4)}, {s('ua')}, {ANYU}
        s_cbranch_scc1 L_unsC
        s_mov_b64 exec, {stp(2)}
        v_mov_b32 {v('T')}, {s('axys')}
        s_set_gpr_idx_on {st(1)}, gpr_idx(DST)
        v_mov_b32 {xy(0)}, {v('T')}
        s_set_gpr_idx_off
        L_unsR:
        """)
        COLD(f"""
        L_unsC:
        s_mov_b32 {ANYU}, 1
        {uns}
        s_branch L_unsR
        """)
    else:
        E(uns)
    if PS:   # the moved particle's cell is a candidate of the pre-screened probe, whatever its old bytes said (exec = its lane)
        assert not T5
        SO(f"""
        s_not_b64 {sp('hA')}, {stp(6)}
        v_or_b32 {v('wb0')}, {s('hA')}, {v('wb0')}
        v_or_b32 {v('wb1')}, {s('hA', 1)}, {v('wb1')}
        """)
    if not Z8:
        E(f"""
        s_lshr_b32 {st(2)}, {st(1)}, 1
        s_lshl_b32 {st(2)}, {st(2)}, 8
        s_and_b32 {st(3)}, {st(1)}, 1
        s_lshl_b32 {st(3)}, {st(3)}, 1
        s_add_u32 {st(2)}, {st(2)}, {st(3)}
        v_add_u32 {v('S6')}, {st(2)}, {v('zaddr')}
        v_mov_b32 {v('T',1)}, {s('az16')}
        ds_write_b16 {v('S6')}, {v('T',1)}
        """)
    E(f"""
    s_lshr_b32 {st(1)}, {st(1)}, {GSH}
    s_lshl_b64 {stp(2)}, 1, {st(1)}
    {"" if T5 else f"v_readfirstlane_b32 {st(6)}, {KRZ}" if TT else f"s_mov_b32 {st(6)}, {s('RZ')}" if (Z8 and not W4) else f"s_load_dword {st(6)}, {KARG}, {K_RZ}"}
    s_sext_i32_i16 {st(0)}, {s('axys') if Z8 else s('az16')}
    {"" if (TT or (Z8 and not W4)) else "s_waitcnt lgkmcnt(0)"}
    s_sub_i32 {st(4)}, {st(0)}, {s('RZ') if T5 else st(6)}
    s_add_i32 {st(5)}, {st(0)}, {s('RZ') if T5 else st(6)}
    s_mov_b64 exec, {stp(2)}
    v_min_i32 {v('gloR')}, {st(4)}, {v('gloR')}
    v_max_i32 {v('ghiR')}, {st(5)}, {v('ghiR')}
    {"L_notmine:" if W4 else ""}
    s_mov_b64 exec, -1
    s_nop 1
    """)
E(f"""
L_reject:
L_noA:
""")
mark(8)

# ---------------------------------------------------------------------------------------------- probe B
G(f"s_cmp_eq_u32 {s('hasB')}, 0")
G("s_cbranch_scc1 L_noB")
if MG:
    # ---- Fm of particle n+1 = probe B's sums + the side result that applies; then its proposal, in group layout
    E(f"""
    s_mov_b64 {sp('lu')}, {sp('nxy')}
    """)
    (G if PEEL else E)(f"""
    s_add_u32 {st(2)}, {s('tl')}, 1
    s_mul_i32 {st(2)}, {st(2)}, 24
    {f"v_add3_u32 v49, {st(2)}, {KC}, v1" if W4 else f"v_add_u32 v49, {st(2)}, {KC}"}
    ds_read_b64 {vp('D',0)}, v49 offset:{LDS_P0}
    """)
    if not ELDS:
        SO(f"""
        v_mad_u32_u24 v49, {s('tl')}, 24, {KC}
        {"v_add_u32 v49, v49, v1" if W4 else ""}
        ds_read_b64 {vp('D',0)}, v49 offset:{LDS_P0 + 24}
        """)
    # (a pass without the side pair -- generic copy only -- takes probe B's sums as they are)
    (G if PEEL else E)(f"""
    v_mov_b32 {v('FmV')}, v20
    v_mov_b32 {v('FmV',1)}, v21
    """)
    G(f"""
    s_cmp_eq_u32 {s('hasA') if W4 else s('sidesHi')}, 0
    s_cbranch_scc1 L_nsr
    """)
    (G if ELDS else E)(f"""
    {f"s_and_b32 {st(0)}, {s('i')}, 1" if W4 else ""}
    {f"s_lshl_b32 {st(0)}, {st(0)}, 1" if W4 else ""}
    {f"s_add_u32 {st(0)}, {st(0)}, {s('accf')}" if W4 else ""}
    v_lshl_add_u32 v48, {st(0) if W4 else s('accf')}, 5, {KSD}
    ds_read_b64 v[46:47], v48 offset:{LDS_SIDEM}
    """)
    # (steady copy: one wait for the side result and the displacement, which the next instruction but one needs anyway)
    (G if PEEL else E)("s_waitcnt lgkmcnt(0)")
    SO("s_waitcnt lgkmcnt(0)" if (EVM and MG and not W4) else "s_waitcnt vmcnt(0) lgkmcnt(0)")
    (G if ELDS else E)(f"v_add_f64 {vp('FmV')}, {MGW}, v[46:47]")
    if ELDS:
        SO(f"v_add_f64 {vp('FmV')}, {MGW}, v[40:41]")
    E("L_nsr:")
    (G if PEEL else E)("s_waitcnt vmcnt(0) lgkmcnt(0)")
    E(f"""
    v_mov_b32 {v('DdV')}, v12
    v_mov_b32 {v('DdV',1)}, v13
    v_fma_f64 {vp('D',1)}, {vp('FmV')}, {sp('AoT')}, {DdNm}
    """)
    (G if ELDS else E)(f"v_add_f64 {vp('D',0)}, {vp('D',0)}, {vp('D',1)}")
    if ELDS:
        SO(f"v_add_f64 {vp('D',0)}, v[36:37], {vp('D',1)}")
    E(f"""
    // wrap x and y (lanes 16..23, 8..15 of a half), fixed point with 256/L ; z (lanes 24..31): fixed point, safe range, near a wall
    s_bfm_b64 exec, 16, 8
    v_mul_f64 {vp('D',1)}, {vp('D',0)}, {sp('invL')}
    v_rndne_f64 {vp('D',1)}, {vp('D',1)}
    v_fma_f64 {vp('D',0)}, -{vp('D',1)}, {sp('L')}, {vp('D',0)}
    v_mul_f64 {vp('D',1)}, {vp('D',0)}, {sp('toFix')}
    s_bfm_b64 exec, 8, 24
    v_mul_f64 {vp('D',1)}, {vp('D',0)}, {sp('zFix')}
    v_cmp_nlt_f64 vcc, |{vp('D',0)}|, {sp('zsafe')}
    v_cmp_nlt_f64 {stp(4)}, |{vp('D',0)}|, {sp('znear')}
    s_mov_b64 exec, -1
    v_rndne_f64 {vp('D',1)}, {vp('D',1)}
    v_mov_b32 {v('T')}, 0x7fff
    v_mov_b32 {v('T',1)}, 0xffff8001
    v_cvt_i32_f64 {v('D',4)}, {vp('D',1)}
    s_bfe_u32 {s('ua')}, vcc_lo, 0x10018
    s_bfe_u32 {s('nearA')}, {st(4)}, 0x10018
    v_med3_i32 {v('D',5)}, {v('D',4)}, {v('T')}, {v('T',1)}
    v_readlane_b32 {s('Q',0)}, {v('D',0)}, 16
    v_readlane_b32 {s('Q',1)}, {v('D',1)}, 16
    v_readlane_b32 {s('Q',2)}, {v('D',0)}, 8
    v_readlane_b32 {s('Q',3)}, {v('D',1)}, 8
    v_readlane_b32 {s('Q',4)}, {v('D',0)}, 24
    v_readlane_b32 {s('Q',5)}, {v('D',1)}, 24
    v_readlane_b32 {st(0)}, {v('D',4)}, 16
    v_readlane_b32 {st(1)}, {v('D',4)}, 8
    v_readlane_b32 {st(2)}, {v('D',5)}, 24
    """)
MUTE[0] = MG
if TT:
    # probe B was evaluated before the decision (team B) and the proposal of particle n+1 formed for both outcomes (above):
    # the half of every row that worked for the outcome that did NOT happen takes the other half's Fm (it is the next move's
    # Fm and must be uniform in its row); position, fixed-point copies and the unsafe flag are read from the lanes
    # 16 row + 8 accf
    E(f"""
    s_add_u32 {st(1)}, {s('i')}, 1
    s_lshl_b32 {st(0)}, {st(1)}, 3
    s_load_dwordx2 {sp('lu')}, {sp('uK')}, {st(0)}
    v_mov_b32_dpp {v('T')}, v14 row_ror:8 row_mask:0xf bank_mask:0xf
    v_mov_b32_dpp {v('T',1)}, v15 row_ror:8 row_mask:0xf bank_mask:0xf
    s_xor_b32 {st(2)}, {s('accf')}, 1
    s_lshl_b32 {st(2)}, {st(2)}, 3
    s_lshl_b32 {st(2)}, 0x00ff00ff, {st(2)}
    s_mov_b32 {st(3)}, {st(2)}
    s_lshl_b32 {st(4)}, {s('accf')}, 3
    s_add_u32 {st(5)}, {st(4)}, 16
    s_add_u32 {st(6)}, {st(4)}, 32
    s_add_u32 {st(7)}, {st(4)}, 48
    v_cndmask_b32 {v('FmV')}, v14, {v('T')}, {stp(2)}
    v_cndmask_b32 {v('FmV',1)}, v15, {v('T',1)}, {stp(2)}
    s_lshr_b64 {stp(2)}, {sp('haveB' if TL else 'planeM')}, {st(7)}
    s_and_b32 {s('ua')}, {st(2)}, 1
    v_readlane_b32 {s('Q',0)}, v16, {st(5)}
    v_readlane_b32 {s('Q',1)}, v17, {st(5)}
    v_readlane_b32 {s('Q',2)}, v16, {st(6)}
    v_readlane_b32 {s('Q',3)}, v17, {st(6)}
    v_readlane_b32 {s('Q',4)}, v16, {st(7)}
    v_readlane_b32 {s('Q',5)}, v17, {st(7)}
    v_readlane_b32 {st(0)}, v20, {st(5)}
    v_readlane_b32 {st(1)}, v20, {st(6)}
    v_readlane_b32 {st(2)}, v21, {st(7)}
    """)
    # (no wait for lu here: every later wait on this counter is lgkmcnt(0), the first at the latest before the exchange)
    mark(9)
else:
    if not KROW:
        E(f"""
        // per-row component offset: rows 1..3 -> 0, 8, 16 (row 0 idles along with component 0)
        v_lshrrev_b32 {v('T')}, 4, {LANE}
        v_add_u32 {v('T')}, -1, {v('T')}
        v_max_i32 {v('T')}, 0, {v('T')}
        v_lshlrev_b32 {v('T')}, 3, {v('T')}
        """)
    E(f"""
    s_add_u32 {st(1)}, {s('i')}, 1
    {f"s_lshl_b32 {st(0)}, {st(1)}, 3" if Z8 else ""}
    {f"s_load_dwordx2 {sp('lu')}, {sp('uK')}, {st(0)}" if Z8 else ""}
    s_mul_i32 {st(1)}, {st(1)}, 24
    v_add_u32 {v('S6')}, {st(1)}, {KROW or v('T')}
    """)
    side_sources()
    E(f"""
    // displacement of move i+1 per row: asked for now, needed after probe B
    global_load_dwordx2 {vp('DdV')}, {v('S6')}, {sp('dK')}{NT}
    s_waitcnt lgkmcnt(0)
    """)
    probe("B", BP, False, "v[18:19]", V['wb0'], V['wb1'], XB_, CB_, sp('haveB'), True, "s_waitcnt vmcnt(1)",
          sp('wallB') if ZB else None, sp('planeB') if ZB else None)
    reduce4(vp('FmV'))
    if W4:
        xchg(vp('FmV'), 1)
(E if not TT else (lambda t: None))(f"""
// ---- proposal of particle n+1 in row layout (SMC.c:307-316): q = p + (Fm A/T + displ)   (z8t: before the decision, above)
// rows 1..3 read component row-1 of p0[rowB] and of displ[3 (i+1) ..]; row 0 idles along with component 0
{"" if KROW else f"v_lshrrev_b32 {v('T')}, 4, {LANE}"}
{"" if KROW else f"v_add_u32 {v('T')}, -1, {v('T')}"}
{"" if KROW else f"v_max_i32 {v('T')}, 0, {v('T')}"}
{"" if KROW else f"v_lshlrev_b32 {v('T')}, 3, {v('T')}"}
s_add_u32 {st(0)}, {s('tl')}, 1
s_mul_i32 {st(0)}, {st(0)}, 24
{f"v_add3_u32 {v('T',1)}, {st(0)}, {v('T')}, v1" if W4 else f"v_add_u32 {v('T',1)}, {st(0)}, {KROW or v('T')}"}
ds_read_b64 {vp('D',0)}, {v('T',1)} offset:{LDS_P0}
{"s_waitcnt lgkmcnt(0)" if TT else "s_waitcnt vmcnt(0) lgkmcnt(0)"}
{'' if Z8 else f"s_mov_b64 {sp('lu')}, {sp('nlu')}"}
v_fma_f64 {vp('D',1)}, {vp('FmV')}, {sp('AoT')}, {vp('DdV')}
v_add_f64 {vp('D',0)}, {vp('D',0)}, {vp('D',1)}
// wrap x and y (rows 1, 2 = lanes 16..47), fixed point with 65536/L
{"s_bfm_b64 exec, 32, 16" if ZB else f"s_mov_b32 {st(0)}, 0xffff0000"}
{"" if ZB else f"s_mov_b32 {st(1)}, 0x0000ffff"}
{"" if ZB else f"s_mov_b64 exec, {stp(0)}"}
v_mul_f64 {vp('D',1)}, {vp('D',0)}, {sp('invL')}
v_rndne_f64 {vp('D',1)}, {vp('D',1)}
v_fma_f64 {vp('D',0)}, -{vp('D',1)}, {sp('L')}, {vp('D',0)}
v_mul_f64 {vp('D',1)}, {vp('D',0)}, {sp('toFix')}
// z (row 3 = lanes 48..63): fixed point with 1/uz; outside the safe range?
{"s_bfm_b64 exec, 16, 48" if ZB else f"s_mov_b32 {st(0)}, 0"}
{"" if ZB else f"s_mov_b32 {st(1)}, 0xffff0000"}
{"" if ZB else f"s_mov_b64 exec, {stp(0)}"}
v_mul_f64 {vp('D',1)}, {vp('D',0)}, {sp('zFix')}
v_cmp_nlt_f64 vcc, |{vp('D',0)}|, {sp('zsafe')}
s_mov_b64 exec, -1
v_rndne_f64 {vp('D',1)}, {vp('D',1)}
v_mov_b32 {v('T')}, 0x7fff
v_mov_b32 {v('T',1)}, 0xffff8001
v_cvt_i32_f64 {v('D',4)}, {vp('D',1)}
s_bfe_u32 {s('ua')}, vcc_hi, 0x10010
v_med3_i32 {v('D',5)}, {v('D',4)}, {v('T')}, {v('T',1)}
v_readlane_b32 {s('Q',0)}, {v('D',0)}, 16
v_readlane_b32 {s('Q',1)}, {v('D',1)}, 16
v_readlane_b32 {s('Q',2)}, {v('D',0)}, 32
v_readlane_b32 {s('Q',3)}, {v('D',1)}, 32
v_readlane_b32 {s('Q',4)}, {v('D',0)}, 48
v_readlane_b32 {s('Q',5)}, {v('D',1)}, 48
v_readlane_b32 {st(0)}, {v('D',4)}, 16
v_readlane_b32 {st(1)}, {v('D',4)}, 32
v_readlane_b32 {st(2)}, {v('D',5)}, 48
""")
MUTE[0] = False
if Z8:
    E(f"""
    s_and_b32 {st(0)}, {st(0)}, 0xff
    s_lshl_b32 {st(1)}, {st(1)}, 8
    s_and_b32 {st(1)}, {st(1)}, 0xff00
    s_or_b32 {st(0)}, {st(0)}, {st(1)}
    s_pack_ll_b32_b16 {s('axys')}, {st(2)}, {st(0)}
    """)
else:
    E(f"""
    s_and_b32 {st(0)}, {st(0)}, 0xffff
    s_lshl_b32 {st(1)}, {st(1)}, 16
    s_or_b32 {st(0)}, {st(0)}, {st(1)}
    s_and_b32 {s('az16')}, {st(2)}, 0xffff
    s_mul_i32 {s('azz')}, {s('az16')}, 0x10001
    """)
E(f"""
{"" if Z8 else f"s_mov_b32 {s('axys')}, {st(0)}" if ZB else f"v_mov_b32 {v('axy')}, {st(0)}"}
""")
E(f"s_cmp_eq_u32 {s('tl')}, 63" if ZB else f"s_cmp_eq_u32 {s('cross')}, 1")
E("s_cbranch_scc0 L_nocross")
rotate("r2")
E(f"s_mov_b32 {s('tl')}, -1")
E("L_nocross:")
E(f"s_add_u32 {s('tl')}, {s('tl')}, 1")
E("L_noB:")
mark(10)
if ZB:
    E(f"s_mov_b32 {s('locA')}, {s('locB')}")
E(f"s_add_u32 {s('i')}, {s('i')}, 1")
if TL:
    E(f"""
    v_xor_b32 {VXR}, {WPR * 512}, {VXR}
    v_xor_b32 {VXW}, {WPR * 512}, {VXW}
    v_xor_b32 {VSR}, 64, {VSR}
    """)
G(f"""
s_mov_b32 {s('hasA')}, 1
{f"s_and_b32 {st(0)}, {WAVE}, {KS - 1}" if TT else ""}
{f"s_cmp_eq_u32 {st(0) if TT else WAVE}, {TT_SIDE if TT else 0}" if (W4 and not MG) else ""}
{f"s_cselect_b32 {s('hasAw')}, 1, 0" if (W4 and not MG) else ""}
""")
E(f"""
s_add_u32 {st(0)}, {s('i')}, 1
s_cmp_lt_i32 {st(0)}, {s('len')}
""")
G(f"s_cselect_b32 {s('hasB')}, 1, 0")
if MG:   # what the next pass has: plane of half A (a proposal and walls), of half B (a next particle and walls), the side pair
    G(f"""
    s_cmp_ge_i32 {s('M2')}, 0
    s_cselect_b32 {s('wlp')}, 1, 0
    s_mov_b32 {st(2)}, {s('hasB')}
    {f"s_cmp_eq_u32 {WAVE}, 0" if W4 else ""}
    {f"s_cselect_b32 {s('wlp')}, {s('wlp')}, 0" if W4 else ""}
    {f"s_cselect_b32 {st(2)}, {st(2)}, 0" if W4 else ""}
    s_and_b32 {s('wlp',1)}, {s('wlp')}, {st(2)}
    s_lshl_b32 {st(1)}, {st(2)}, 1
    s_add_u32 {s('stB')}, {s('wlp',1)}, {st(1)}
    s_mul_i32 {st(1)}, {st(2)}, 3
    s_lshl_b32 {s('sidesHi')}, {st(1)}, {s('wlp',1)}
    v_mov_b32 v22, {s('wlp')}
    v_mov_b32 v23, {s('stB')}
    """)
    mg_coeff_init(G)
    G(f"s_cmp_lt_i32 {st(0)}, {s('len')}")
# the next pass has a proposal to decide (hasA) and a next particle (hasB): the steady copy runs it
E("s_cbranch_scc1 L_S_move" if PEEL else "")
SO(f"""
s_mov_b32 {s('hasB')}, 0
{f"s_mov_b32 {s('wlp',1)}, 0" if MG else ""}
{f"s_mov_b32 {s('stB')}, 0" if MG else ""}
{f"s_mov_b32 {s('sidesHi')}, 0" if MG else ""}
{"v_mov_b32 v23, 0" if MG else ""}
s_branch L_G_move
""")
G(f"""
s_cmp_lt_i32 {s('i')}, {s('len')}
s_cbranch_scc1 L_move
""")
MOVE_END = len(out)
E(f"""
L_run_next:
s_add_u32 {s('run')}, {s('run')}, 1
s_cmp_lt_u32 {s('run')}, 2
s_cbranch_scc1 L_run
// C: rec[sw] = (E, accepted) for the bookkeeping kernel (SMC.c:194-195)
v_mov_b32 v14, {s('E')}
v_mov_b32 v15, {s('E',1)}
v_mov_b32 v16, {s('jacc')}
v_mov_b32 v17, 0
""")
if not ZB:
    E(f"""
    s_lshl_b32 {st(0)}, {s('sw')}, 4
    v_mov_b32 v18, {st(0)}
    s_mov_b64 exec, 1
    s_nop 1
    global_store_dwordx4 v18, v[14:17], {sp('rec')}
    s_mov_b64 exec, -1
    """)
else:
    E(f"""
    s_load_dwordx2 {stp(2)}, {KARG}, {K_REC}
    s_load_dword {st(0)}, {KARG}, {K_INTS + 4}
    s_waitcnt lgkmcnt(0)
    s_mul_i32 {st(0)}, {st(0)}, {REP}
    s_add_u32 {st(0)}, {st(0)}, {s('sw')}
    s_lshl_b32 {st(0)}, {st(0)}, 4
    v_mov_b32 v18, {st(0)}
    s_mov_b64 exec, 1
    s_nop 0
    global_store_dwordx4 v18, v[14:17], {stp(2)}
    s_mov_b64 exec, -1
    """)
E(f"""
s_add_u32 {s('sw')}, {s('sw')}, 1
s_cmp_lt_u32 {s('sw')}, {s('nsw')}
s_cbranch_scc1 L_sweep
""")
if Z8C:
    cnt_addr("v25")
    E(f"""
    ds_read_b32 v22, v25 offset:{LDS_CNT}
    ds_read_b32 v23, v25 offset:{LDS_CNT + 4}
    ds_read_b32 v24, v25 offset:{LDS_CNT + 8}
    ds_read_b32 v34, v25 offset:{LDS_CNT + 12}
    ds_read_b32 v35, v25 offset:{LDS_CNT + 16}
    ds_read_b32 v40, v25 offset:{LDS_CNT + 20}
    ds_read_b32 v42, v25 offset:{LDS_CNT + 24}
    ds_read_b32 v44, v25 offset:{LDS_CNT + 28}
    s_load_dwordx2 {stp(2)}, {KARG}, {K_DBG}
    v_mov_b32 v30, 0
    v_mov_b32 v27, 0
    s_waitcnt lgkmcnt(0)
    s_mov_b64 exec, 1
    v_mov_b32 v26, v22
    global_atomic_add_x2 v30, v[26:27], {stp(2)}
    v_mov_b32 v28, v23
    v_mov_b32 v29, 0
    global_atomic_add_x2 v30, v[28:29], {stp(2)} offset:8
    v_mov_b32 v32, v24
    v_mov_b32 v33, 0
    global_atomic_add_x2 v30, v[32:33], {stp(2)} offset:16
    v_mov_b32 v36, v34
    v_mov_b32 v37, 0
    global_atomic_add_x2 v30, v[36:37], {stp(2)} offset:24
    v_mov_b32 v38, v35
    v_mov_b32 v39, 0
    global_atomic_add_x2 v30, v[38:39], {stp(2)} offset:32
    v_mov_b32 v41, 0
    v_mov_b32 v43, 0
    global_atomic_add_x2 v30, v[40:41], {stp(2)} offset:40
    global_atomic_add_x2 v30, v[42:43], {stp(2)} offset:48
    v_mov_b32 v45, 0
    global_atomic_add_x2 v30, v[44:45], {stp(2)} offset:56
    s_mov_b64 exec, -1
    s_waitcnt vmcnt(0)
    """)
if ZBC:
    E(f"""
    v_lshlrev_b32 v25, 3, {LANE}
    ds_read_b64 v[22:23], v25 offset:{LDS_CNT}
    s_load_dwordx2 {stp(2)}, {KARG}, {K_DBG}
    v_mov_b32 v30, 0
    v_mov_b32 v27, 0
    v_mov_b32 v29, 0
    s_waitcnt lgkmcnt(0)
    v_mov_b32 v26, v22
    v_mov_b32 v28, v23
    global_atomic_add_x2 v30, v[26:27], {stp(2)} offset:8
    global_atomic_add_x2 v30, v[28:29], {stp(2)} offset:16
    s_waitcnt vmcnt(0)
    """)
CLK1 = clk_ptr(stp(2))
if TT and STAMPS:   # add this wave's 16 words to clk[rep][wave * 16 + k] (up to 16 waves) instead of the end stamp (no start stamp either)
    E(f"""
    s_lshl_b32 {st(4)}, {WAVE}, 6
    v_lshl_add_u32 v20, {LANE}, 2, {st(4)}
    s_lshl_b32 {st(5)}, {WAVE}, 7
    v_lshl_add_u32 v21, {LANE}, 3, {st(5)}
    v_mov_b32 v23, 0
    s_mov_b64 exec, 0xffff
    ds_read_b32 v22, v20 offset:{LDS_TM}
    s_waitcnt lgkmcnt(0)
    global_atomic_add_x2 v21, v[22:23], {CLK1}
    s_mov_b64 exec, -1
    s_waitcnt vmcnt(0)
    s_endpgm
    """)
E(f"""
// end stamp
s_memtime {stp(4)}
{f"s_mov_b32 {st(6)}, m0" if (IVA is not None or IVC is not None) else f"s_memrealtime {stp(6)}"}
{f"s_mov_b32 {st(7)}, 0" if (IVA is not None or IVC is not None) else ""}
s_waitcnt lgkmcnt(0)
v_mov_b32 v14, 16
v_mov_b32 v16, {st(4)}
v_mov_b32 v17, {st(5)}
v_mov_b32 v18, {st(6)}
v_mov_b32 v19, {st(7)}
s_mov_b64 exec, 1
s_nop 1
global_store_dwordx4 v14, v[16:19], {CLK1}
s_mov_b64 exec, -1
s_waitcnt vmcnt(0) lgkmcnt(0)
""")

if PEEL:
    # two copies of the move: [cold pieces + move] generic (all lines but the @S ones), and steady (all but the @G ones,
    # the labels it defines renamed L_x -> L_S_x); the steady copy ends in unconditional branches and lies first
    def variant(lines, drop, keep):
        return [l[:-3] if l.endswith(keep) else l for l in lines if not l.endswith(drop)]
    chunk = out[MOVE_AT:MOVE_END]
    gen = variant(cold, " @S", " @G") + variant(chunk, " @S", " @G")
    ste = variant(cold, " @G", " @S") + variant(chunk, " @G", " @S")
    defined = {l[2:-1] for l in ste if re.fullmatch(r"L_\w+:", l)} - {"jb"}
    ste = [re.sub(r"\bL_(\w+)", lambda m: ("L_S_" + m.group(1)) if m.group(1) in defined else m.group(0), l) for l in ste]
    assert not any(l.endswith((" @G", " @S")) for l in out[:MOVE_AT] + out[MOVE_END:])
    out[MOVE_AT:MOVE_END] = ste + gen
elif COLD_AT is not None:
    out[COLD_AT:COLD_AT] = cold
# peephole: a full write of exec that the next instruction (labels in between do not matter: whoever jumps there arrives behind
# it) overwrites in full is dead -- the helpers each restore exec = -1 at their end, and the next one sets its own mask
def _dead_exec_writes(lines):
    keep, n = [], 0
    for i, l in enumerate(lines):
        if l.startswith("s_mov_b64 exec, "):
            j = i + 1
            while j < len(lines) and re.fullmatch(r"L_\w+:", lines[j]):
                j += 1
            if j < len(lines) and lines[j].startswith("s_mov_b64 exec, "):
                n += 1
                continue
        keep.append(l)
    return keep, n
out, NDEAD = _dead_exec_writes(out)
with open(sys.argv[1] if len(sys.argv) > 1 else "smcx_sweep_ma_body.inc", "w") as f:
    f.write("// generated by gen_sweep_ma.py -- do not edit\n")
    for ln in out:
        # labels are per variant: the three bodies are assembled into one object
        ln = re.sub(r"\bL_(\w+)", r"L%d%s%s_\1" % (NS, MODE, "x%d" % WPR if W4 else ""), ln)
        f.write('"%s\\n\\t"\n' % ln)
print("%d lines (%d dead exec writes dropped)" % (len(out), NDEAD), file=sys.stderr)
