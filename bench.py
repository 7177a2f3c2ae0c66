#!/usr/bin/env python3
"""bench.py -- headline benchmark of the SMC hot path on MI355X.

Metric (BASELINE.json): pair-evals/s = nrep * sweeps * 2N(N-1) / time at N=4096.
One "step" = one Smart-Monte-Carlo sweep (N force-biased trial moves, SMC.c:278-351)
of every replica chain on every GPU.  Workload at any GPU count: BASELINE config 3 per
GPU (N=4096 + wall, 4096 replicas, fcc(8,16) start, T=A=1.1, M=3, W fixture, seeds
12345 + global replica index) -- weak scaling, config 4 at 8 GPUs.  Replicas are
independent (the reference's intended MPI fan-out), so ranks share nothing while
sampling; the only collective is the final RCCL all-gather of the observables.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--replicas R] [--no-cpu] [--equilibrate E]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

The JSON line carries numbers and kernel names only (round 5: < 6 KB, so that every entry survives in the driver's tail); what
each key means, how it is measured and the caveats that used to ride along as note strings are in DESIGN.md section 8.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# bytes of HBM traffic per byte of FETCH_SIZE for THIS path's reads (24-byte gathers of candidate positions, row fills of 64
# consecutive 24-byte records): calibrated by tools/ubench/gather24.hip on known record counts
# (profiles/r05_fetch_size_24B_gather.txt): a random 24-byte gather from a 3 GiB table reads FETCH_SIZE = 73.7 B per record.
# 64-byte requests counted in full could not give less than 80 (1.25 sectors per record); 128-byte requests tallied at 64 give 72
# (1.125 lines): the counter halves this pattern like the guide's streamed read; rows of consecutive records read 12.0 B per
# 24-byte record.  So x 2 throughout.  WRITE_SIZE is exact (guide).  bytes = 2 FETCH_SIZE + WRITE_SIZE.
HBM_FETCH_FACTOR = 2.0
BYTES_PER_PAIR_EVAL = 24  # one neighbour position = 3 fp64 per pair-eval
N_SIMD = 256 * 4          # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32
# issue cost of one wave64 VALU instruction on a SIMD-32, MI355X_MICROARCH.md ("v_fma_f32 (wave64) 2 cyc",
# fp64 at half rate: 4, transcendental 8 -> fp64 transcendental 16)
GUIDE_COST = {"b32": 2.0, "f64": 4.0, "trans_f64": 16.0}
# the same measured on this chip at four waves per SIMD (profiles/r02_issue_costs.txt): the fast 32-bit forms (VOP1/VOP2
# add/sub/and/mov/shift with VGPR or inline-constant sources, v_fma_f32) 1.87 cycles, every other 32-bit form (packed,
# dot, alignbit, 3-operand integer, conversions, compares, DPP, lane reads, any SGPR source) 3.44, fp64 3.43, v_rcp_f64 8.
# The PMC classes do not separate fast from slow 32-bit forms, so the VALU pipe's busy fraction is bracketed:
# every 32-bit instruction fast .. every one slow.
MEASURED_COST_LO = {"b32": 1.87, "f64": 3.43, "trans_f64": 8.0}
MEASURED_COST_HI = {"b32": 3.44, "f64": 3.43, "trans_f64": 8.0}


def issue_roofline(kname, sweep_ms_per_sweep, nrep, N, clock_ghz, start=None):
    """VALU-issue roofline of a sweep kernel: SIMD cycles the executed wave-instructions need at the
    guide's issue costs, over the SIMD cycles that passed (1024 SIMDs x in-kernel clock x time).
    Instruction counts per wave-move by class come from the committed PMC run of this kernel on this
    workload (profiles/kernel_counters.json, tools/profile_valu.sh + tools/pmc_to_json.py); a wave-move is
    one trial move seen by one wavefront (the several-wavefront kernels run every move on all their waves)."""
    path = os.path.join(ROOT, "profiles", "kernel_counters.json")
    if not os.path.exists(path) or not clock_ghz:
        return None
    allk = json.load(open(path))
    kc = (allk.get("%s@%s" % (kname, start)) if start else None) or allk.get(kname)   # a second start state of a kernel: "name@start"
    if not kc:
        return None
    if kc["workload"]["replicas"] != nrep or kc["workload"]["N"] != N or \
            (start is not None and kc["workload"].get("start") != start):
        return None                                                  # counters are of another workload / start state
    # ... or of another BUILD of the kernel: the counts carry the source identity of the library they were taken from
    # (sha256 of the generated body / the source files, smcx_kernel_source_id); a kernel edited since has another one
    try:
        import smcx_loader
        built = smcx_loader.load().kernel_source_id(kname)
    except Exception:
        built = None
    if not built or kc.get("source_id") != built:
        return {"bound": "valu_issue", "achieved": None, "peak": N_SIMD * clock_ghz, "unit": "G SIMD-cycles/s of VALU issue",
                "frac": None, "clock_ghz": clock_ghz,
                "note": "the committed PMC instruction counts (profiles/kernel_counters.json, source_id %s) are of another "
                        "build of %s than the loaded library (source_id %s): re-profile with tools/profile_configs.sh"
                        % (kc.get("source_id"), kname, built)}
    m = kc["per_wave_move"]
    f64 = m.get("SQ_INSTS_VALU_ADD_F64", 0) + m.get("SQ_INSTS_VALU_MUL_F64", 0) + m.get("SQ_INSTS_VALU_FMA_F64", 0)
    tr = m.get("SQ_INSTS_VALU_TRANS_F64", 0)
    b32 = m["SQ_INSTS_VALU"] - f64 - tr
    wpr = kc["workload"].get("waves_per_replica", 1)
    moves_per_s = nrep * N / (sweep_ms_per_sweep * 1e-3)             # trial moves per second, all replicas
    wave_moves_per_s = moves_per_s * wpr
    need = b32 * GUIDE_COST["b32"] + f64 * GUIDE_COST["f64"] + tr * GUIDE_COST["trans_f64"]
    need_lo = b32 * MEASURED_COST_LO["b32"] + f64 * MEASURED_COST_LO["f64"] + tr * MEASURED_COST_LO["trans_f64"]
    need_hi = b32 * MEASURED_COST_HI["b32"] + f64 * MEASURED_COST_HI["f64"] + tr * MEASURED_COST_HI["trans_f64"]
    peak = N_SIMD * clock_ghz                                        # G SIMD-cycles per second
    achieved = need * wave_moves_per_s / 1e9
    # Round 4 measured what ONE instruction of each kind costs the running kernel at four wavefronts per SIMD (40 extra
    # instructions per move, profiles/r04_instruction_costs_in_kernel.txt): scalar ALU 2.0 SIMD cycles, fast VALU form 1.8, slow
    # 32-bit VALU form 3.2, fp64 3.4 -- and their sum over the kernel's instructions IS its time: a SIMD issues one instruction
    # of any kind at a time.  So besides the VALU-only fraction (`frac`, the guide's costs) the line carries the fraction of
    # the SIMDs' cycles that ALL executed instructions account for at those measured costs, bracketed because the PMC classes do
    # not separate fast from slow 32-bit VALU forms (LDS / memory / scalar-memory instructions priced like a scalar one).
    other = ((m.get("SQ_INSTS_SALU") or 0) + (m.get("SQ_INSTS_BRANCH") or 0) + (m.get("SQ_INSTS_LDS") or 0) +
             (m.get("SQ_INSTS_VMEM_RD") or 0) + (m.get("SQ_INSTS_VMEM_WR") or 0) + (m.get("SQ_INSTS_SMEM") or 0))
    all_lo = b32 * 1.8 + f64 * 3.4 + tr * 8.0 + other * 2.0
    all_hi = b32 * 3.2 + f64 * 3.4 + tr * 8.0 + other * 2.0
    out = {"bound": "valu_issue", "achieved": achieved, "peak": peak, "unit": "G SIMD-cycles/s of VALU issue",
           "frac": achieved / peak,
           # all executed instructions at the per-kind costs measured inside this kernel, 32-bit VALU all-fast .. all-slow
           "frac_all_kinds": [all_lo * wave_moves_per_s / 1e9 / peak,
                                                                  all_hi * wave_moves_per_s / 1e9 / peak],
           "instr_per_move": m["SQ_INSTS_VALU"] + other,
           # the same with the 32-bit VALU split into fast and slow forms by opcode (static count of the generated steady copy,
           # tools/phase_table.py; only where the committed counters carry that split): ONE number instead of the bracket
           "frac_all_kinds_point": (((b32 * (kc["valu32_fast_fraction_static"] * 1.8 + (1.0 - kc["valu32_fast_fraction_static"]) * 3.2)
                                     + f64 * 3.4 + tr * 8.0 + other * 2.0) * wave_moves_per_s / 1e9 / peak)
                                    if kc.get("valu32_fast_fraction_static") else None),
           "clock_ghz": clock_ghz, "waves_per_replica": wpr,
           "valu_per_move": m["SQ_INSTS_VALU"], "fp64_per_move": f64, "fp64_trans_per_move": tr,
           "salu_per_move": m.get("SQ_INSTS_SALU"), "branch_per_move": m.get("SQ_INSTS_BRANCH"), "lds_per_move": m.get("SQ_INSTS_LDS"),
           "vmem_per_move": (m.get("SQ_INSTS_VMEM_RD") or 0) + (m.get("SQ_INSTS_VMEM_WR") or 0),
           "cycles_per_move": peak * 1e9 / wave_moves_per_s,
           "wait_any_frac": kc.get("wait_any_frac"), "wait_inst_any_frac": kc.get("wait_inst_any_frac")}
    hb = kc.get("hbm_bytes_per_sweep")
    if hb:
        sec = sweep_ms_per_sweep * 1e-3
        # ONE calibrated reading (round 5, tools/ubench/gather24.hip, profiles/r05_fetch_size_24B_gather.txt): see HBM_FETCH_FACTOR
        fetch = hb["fetch_x2_plus_write"] - hb["fetch_x1_plus_write"]          # FETCH_SIZE and WRITE_SIZE in bytes per sweep
        write = hb["fetch_x1_plus_write"] - fetch
        byt = HBM_FETCH_FACTOR * fetch + write
        out["hbm"] = {"bytes_per_sweep": byt, "gbs": byt / sec / 1e9, "frac_of_peak": byt / sec / 1e9 / HBM_PEAK_GBS,
                      "compulsory_bytes_per_sweep": nrep * (48.0 * N + 32.0 * N + 8)}
    return out


_EXEC_WORKER = r"""
import sys, os, ctypes as C, importlib.util, json
import numpy as np
root = sys.argv[1]
os.environ["SMCX_LIB"] = os.path.join(root, "montecarlo-surfacer_amd", "libsmcx_check.so")
spec = importlib.util.spec_from_file_location("smcx_chk", os.path.join(root, "montecarlo-surfacer_amd", "__init__.py"))
K = importlib.util.module_from_spec(spec); spec.loader.exec_module(K)
N, Na, Nz, nrep, nsw, slots, waves, dev = (int(v) for v in sys.argv[2:10])
state = sys.argv[10] if len(sys.argv) > 10 else ""     # positions [nrep][3N] of a run in progress instead of the lattice
p = K.default_params(N, nrep, tune_slots=slots, tune_waves=waves, device=dev)
with K.Engine(p) as eng:
    name = eng.kernel_form[1]
    eng.upload(np.load(state) if state else K.fcc_init(Na, Nz), K.W_REFERENCE)
    eng.run(0, nsw, 10)
    cnt = (C.c_uint64 * 8)()
    f = K._lib().smcx_debug_work_counts
    f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    assert f(eng._h, cnt) == 0
print(json.dumps({"name": name, "inside": int(cnt[0]), "cand": int(cnt[1]), "miss": int(cnt[2]), "groups": int(cnt[3]),
                  "passes": int(cnt[4]), "more_rounds": int(cnt[5]), "fold": int(cnt[6]), "unworked": int(cnt[7])}))
"""


def executed_work(kname, N, lattice, slots, waves, device, nrep=64, sweeps=2, state=None):
    """What the z-ordered kernels EXECUTE per probe, counted by the diagnostic build of the same sources
    (libsmcx_check.so, SMCX_CHECK_MB=2: counters beside every screen pass, plus the fp64 test of every cell) on a
    sample of the same start -- or of `state`, positions [nrep][3N] taken from a run in progress -- in a child process
    after the timed region: 4-slot groups screened per pass, hence cells tested per move, candidate bits per probe (the
    cells that are no neighbours by construction included: the kernels drop them by a compare of the hand-over item), pairs
    truly inside the cutoff, pairs the screen missed (0), rounds of the fp64 body beyond the first per probe."""
    import subprocess
    import tempfile
    if not os.path.exists(os.path.join(ROOT, "montecarlo-surfacer_amd", "libsmcx_check.so")):
        return {"note": "libsmcx_check.so not built"}
    env = {k: v for k, v in os.environ.items() if not (k.startswith("SMCX_") and k not in ("SMCX_FORCE_DEVICE",))}
    env["SMCX_CHECK_MB"] = "2"
    args = [sys.executable, "-c", _EXEC_WORKER, ROOT, str(N), str(lattice[0]), str(lattice[1]), str(nrep), str(sweeps), str(slots),
            str(waves), str(device)]
    tmp = None
    if state is not None:
        import numpy as np
        tmp = tempfile.NamedTemporaryFile(suffix=".npy", delete=False)
        np.save(tmp, state[:nrep])
        tmp.close()
        args.append(tmp.name)
    try:
        r = subprocess.run(args, env=env, capture_output=True, text=True, timeout=600)
    finally:
        if tmp is not None:
            os.unlink(tmp.name)
    if r.returncode != 0:
        return {"note": "diagnostic run failed: " + r.stderr[-300:]}
    d = json.loads(r.stdout.strip().splitlines()[-1])
    if d["name"] != kname:
        return {"note": "diagnostic build ran %s, not %s" % (d["name"], kname)}
    moves = float(nrep) * sweeps * N
    return {"groups_per_pass": d["groups"] / max(d["passes"], 1), "groups_per_wavefront": slots // 4,
            "wavefronts_per_replica": waves, "passes_per_move": d["passes"] / moves,
            "cells_per_move": d["groups"] * 256.0 / moves, "cells_per_move_all_pairs": 2.0 * (N - 1),
            "fraction_screened": d["groups"] * 256.0 / moves / (2.0 * (N - 1)),
            "candidate_bits": d["cand"] / (2.0 * moves), "pairs_in_cutoff": d["inside"] / (2.0 * moves),
            "missed": d["miss"], "unworked": d.get("unworked", 0),
            "further_rounds": d.get("more_rounds", 0) / (2.0 * moves),
            "sample": "%d replicas x %d sweeps" % (nrep, sweeps)}


def progress(msg):
    """one line per stage on stderr (a long silent run looks hung to the job runner; stdout carries only the JSON line)"""
    print("[bench %.0fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cores():
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(cores, 16))  # the GPU box's CPU share for one GPU


def cpu_baseline_reference(N, Na, Nz, seconds_target=12.0):
    """The REAL reference (kind "reference"): oneParticleMoves of SMC.c compiled where it lies by oracle/build_ref.sh into
    oracle/_ref/libref_smc_N<n>_O3.so (prebuilt; travels with the snapshot), one chain per PROCESS and core -- the
    reference's own MPI fan-out, and its libc rand() state is per process.  None when the library is not there."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "_ref", "libref_smc_N%d_O3.so" % N)
    script = os.path.join(ROOT, "oracle", "time_ref.py")
    if not (os.path.exists(so) and os.path.exists(script)):
        return None
    cores = host_cores()

    def run(seed, sweeps):
        return subprocess.Popen([sys.executable, script, str(N), str(Na), str(Nz), str(seed), str(sweeps)],
                                stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    try:
        out, _ = run(12345, 1).communicate(timeout=120)          # calibrate: one sweep on one core
        t1 = float(out.split()[0])
        sweeps = max(1, int(seconds_target / max(t1, 1e-3) / 1.3))  # cores slow down when all are busy
        procs = [run(12345 + i, sweeps) for i in range(cores)]
        secs = [float(pr.communicate(timeout=600)[0].split()[0]) for pr in procs]
    except Exception:
        return None
    wall = max(secs)                                             # the slowest chain's time inside its sweep loop
    pe = cores * sweeps * 2.0 * N * (N - 1.0)
    return {"value": pe / wall, "unit": "pair-evals/s", "cores": cores, "kind": "reference", "cpu_model": cpu_model(),
            "per_core": pe / wall / cores,
            "sample": "%d chains (one process per core) x %d sweeps, the reference's oneParticleMoves, slowest %.1f s" % (cores, sweeps, wall)}


def cpu_baseline(N, Na, Nz, seconds_target=12.0):
    """The oracle (CPU restatement, kind "port") timed on this box's host cores: one
    independent chain per core, the reference's intended MPI fan-out."""
    import ctypes as C
    import subprocess
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    # a -O3 -march=native build of the same source for the timing leg (the -O2
    # -ffp-contract=off build stays the parity checker)
    lib = None
    try:
        tmp = tempfile.mkdtemp(prefix="smcx_cpu_")
        so = os.path.join(tmp, "liboracle_fast.so")
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-shared", "-o", so,
                               os.path.join(O.ORACLE_DIR, "smc_oracle.c"), "-lm"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lib = C.CDLL(so)
        lib.orc_time_sweeps.argtypes = O.lib().orc_time_sweeps.argtypes
        lib.orc_time_sweeps.restype = C.c_double
        flags = "gcc -O3 -march=native"
    except Exception:
        lib = O.lib()
        flags = "gcc -O2 -ffp-contract=off"
    cores = host_cores()
    R0 = O.fcc(Na, Nz)
    s = O.make_sys(N)

    def one(args):
        seed, sweeps = args
        R = R0.copy()
        acc = C.c_uint64(0)
        return lib.orc_time_sweeps(C.byref(s), seed, R.ctypes.data_as(C.POINTER(C.c_double)),
                                   O.W_FIXTURE.ctypes.data_as(C.POINTER(C.c_double)), 1.1, 1.1, sweeps,
                                   C.byref(acc))
    t1 = one((12345, 1))                      # calibrate: one sweep on one core
    sweeps = max(1, int(seconds_target / max(t1, 1e-3) / 1.3))  # cores slow down when all are busy
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:     # ctypes releases the GIL
        list(ex.map(one, [(12345 + i, sweeps) for i in range(cores)]))
    wall = time.perf_counter() - t0
    pe = cores * sweeps * 2.0 * N * (N - 1.0)
    return {"value": pe / wall, "unit": "pair-evals/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "per_core": pe / wall / cores,
            "sample": "%d chains (one per core) x %d sweeps of N=%d, fcc(%d,%d), oracle/smc_oracle.c built %s, %.1f s wall"
                      % (cores, sweeps, N, Na, Nz, flags, wall)}


def compact(o, digits=5):
    """the JSON line in few bytes: floats to `digits` significant digits, None-valued keys dropped"""
    if isinstance(o, float):
        return float("%.*g" % (digits, o))
    if isinstance(o, dict):
        return {k: compact(v, digits) for k, v in o.items() if v is not None or k in ("vs_baseline", "frac", "traffic", "value")}
    if isinstance(o, (list, tuple)):
        return [compact(v, digits) for v in o]
    return o


SIDE_ROOFLINE_KEYS = ("frac", "frac_all_kinds", "frac_all_kinds_point", "valu_per_move", "fp64_per_move",
                      "salu_per_move", "branch_per_move", "instr_per_move", "cycles_per_move", "wait_any_frac",
                      "wait_inst_any_frac", "note")
SIDE_EXECUTED_KEYS = ("groups_per_pass", "cells_per_move", "candidate_bits",
                      "pairs_in_cutoff", "missed", "unworked",
                      "further_rounds", "note")


def z_profile_width(ob, p):
    """standard deviation of z over the gathers of the last run (the wall-normal profile's width: identifies the state)"""
    import numpy as np
    zh = ob["zhist"].sum(axis=0).astype(float)
    if zh.sum() == 0:
        return None
    zc = (np.arange(p.Ncz) + 0.5) / p.Ncz * p.Lz - p.Lz / 2
    zm = (zh * zc).sum() / zh.sum()
    return float(np.sqrt((zh * (zc - zm) ** 2).sum() / zh.sum()))


def side_config(S, label, N, nrep, lattice, sweeps, device, kernel=0, executed=True, equilibrate=0):
    """one of the other BASELINE configurations (or the headline workload through another kernel, or after `equilibrate`
    sweeps of the chain: the state a production run sits in), run briefly AFTER the timed region (not the headline)"""
    p = S.default_params(N, nrep, device=device, tune_kernel=kernel)
    state = None
    progress("side config: %s" % label)
    with S.Engine(p) as e:
        e.upload(S.fcc_init(*lattice), S.W_REFERENCE)
        e.run(0, max(1, equilibrate), 10)
        e.run(0, sweeps, min(10, sweeps))              # (a gather inside the measured run: the z profile of THIS state)
        ms, launches = e.last_kernel_ms()
        run_ms = e.last_run_ms()
        ob = e.observables()
        try:
            ghz, _ = e.last_clock()
        except Exception:
            ghz = None
        s_, w_, _ = e.geometry
        kname = e.kernel_form[1]
        zipped = "kernel_mc" in kname or "kernel_mt" in kname or "kernel_ml" in kname
        if equilibrate and executed and zipped:
            state = e.positions()[:64 if N <= 4096 else 8].copy()
    pe = nrep * sweeps * 2.0 * N * (N - 1.0)
    start = "fcc(%d,%d)" % tuple(lattice) + ("+%d sweeps" % equilibrate if equilibrate else "")
    out = {"workload": label, "N": N, "replicas": nrep, "start": start, "sweeps": sweeps, "value": pe / (run_ms * 1e-3),
           "ms_per_sweep": ms / sweeps, "kernel": kname.replace("smcx::sweep_kernel_", ""),
           "acceptance": float(ob["acceptance_ratio"].mean()), "mean_E_last": float(ob["E_last"].mean()), "z_std": z_profile_width(ob, p)}
    rl = issue_roofline(kname, ms / sweeps, nrep, N, ghz, start=start) or {"bound": "valu_issue", "frac": None, "clock_ghz": ghz}
    out["roofline"] = {k: rl[k] for k in SIDE_ROOFLINE_KEYS if k in rl}
    if rl.get("hbm"):
        out["roofline"]["hbm_gbs"] = rl["hbm"]["gbs"]
        out["roofline"]["hbm_frac_of_peak"] = rl["hbm"]["frac_of_peak"]
    if executed and zipped:
        ex = executed_work(kname, N, lattice, s_, w_, device, nrep=min(nrep, 64 if N <= 4096 else 8), state=state)
        out["executed"] = {k: ex[k] for k in SIDE_EXECUTED_KEYS if k in ex}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--replicas", type=int, default=4096, help="replica chains per GPU")
    ap.add_argument("--N", type=int, default=4096, choices=[256, 1024, 4096, 16384])
    ap.add_argument("--slots", type=int, default=0)
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--resort", type=int, default=0, help="measurement switch: smcx_params.tune_resort (sweeps per z sort)")
    ap.add_argument("--kernel", type=int, default=0, help="measurement switch: smcx_params.tune_kernel (SMCX_KERNEL_*)")
    ap.add_argument("--lattice", default="", help="measurement switch: fcc start Na,Nz other than the configuration's (e.g. 16,4: the "
                                                  "dense film of other_configs, for its PMC passes)")
    ap.add_argument("--equilibrate", type=int, default=0,
                    help="measurement switch: this many sweeps of the chain BEFORE the warm-up (the PMC passes of the state a "
                         "production run sits in; the bench line's other_configs carry it without the switch)")
    ap.add_argument("--no-cpu", action="store_true",
                    help="skip the reference legs after the timed region (cpu_baseline, all-fp64 kernels): profiling runs")
    a = ap.parse_args()

    import importlib.util
    spec = importlib.util.spec_from_file_location("smcx_dist", os.path.join(ROOT, "montecarlo-surfacer_amd", "dist.py"))
    D = importlib.util.module_from_spec(spec); spec.loader.exec_module(D)

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher.  N fresh ranks (one per GPU) are
        # started before anything in this process touches the GPU; rank 0 prints the JSON line.
        sys.exit(D.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:   # never run a different rank count than asked for and label it --gpus
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d; launch one rank per GPU "
                         "(python bench.py --gpus N does it by itself)" % (a.gpus, world))

    import numpy as np
    import torch
    import smcx_loader
    S = smcx_loader.load()   # raises if libsmcx.so is missing: no fallback

    # rehearsal knobs (one-GPU box): SMCX_DIST_BACKEND=gloo, SMCX_FORCE_DEVICE=0
    backend = os.environ.get("SMCX_DIST_BACKEND", "nccl")
    if "SMCX_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["SMCX_FORCE_DEVICE"])
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world)  # "nccl" = RCCL over xGMI
    lattice = {256: (4, 4), 1024: (8, 4), 4096: (8, 16), 16384: (16, 16)}[a.N]
    if a.lattice:
        lattice = tuple(int(x) for x in a.lattice.split(","))
        if 4 * lattice[0] * lattice[0] * lattice[1] != a.N:
            raise SystemExit("--lattice %s does not hold N=%d particles" % (a.lattice, a.N))
    N, nrep = a.N, a.replicas
    first, _ = D.shard(nrep * world, rank, world)
    p = S.default_params(N, nrep, device=local_rank, first_replica=first,
                         tune_slots=a.slots, tune_waves=a.waves, tune_resort=a.resort, tune_kernel=a.kernel)
    eng = S.Engine(p)
    granule, granule_note = eng.replica_granule()      # replicas the device runs at once with this kernel (sizing rule)
    eng.upload(S.fcc_init(*lattice), S.W_REFERENCE)   # inputs resident in HBM before timing
    gather_lapse = 10                                  # SURVEY.md 8d throughput runs

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if a.equilibrate > 0:
        eng.run(0, a.equilibrate, gather_lapse)
    progress("warm-up and timed region")
    warm_ms, warm_launches = 0.0, 0
    if a.warmup > 0:
        eng.run(0, a.warmup, gather_lapse)
        warm_ms, warm_launches = eng.last_kernel_ms()
    barrier()
    t0 = time.perf_counter()
    eng.run(0, a.steps, gather_lapse)                  # exactly K steps; returns after stream sync
    barrier()
    dt = time.perf_counter() - t0
    sweep_ms, launches = eng.last_kernel_ms()          # HIP events around the sweep launches
    run_ms = eng.last_run_ms()
    try:
        clock_ghz, wave_cycles = eng.last_clock()      # measured inside the last sweep launch of the timed run
    except Exception:
        clock_ghz, wave_cycles = None, None

    # the one exchange step: all-gather of the per-replica observables
    nbytes = eng.obs_device_bytes()
    buf = torch.zeros(nbytes // 8, dtype=torch.float64, device="cuda:%d" % local_rank)
    eng.export_observables_device(buf.data_ptr(), nbytes)
    tg = time.perf_counter()
    obs = D.gather_observables(buf if backend == "nccl" or world == 1 else buf.cpu(), nrep, p.Ncz)
    torch.cuda.synchronize()
    gather_ms = (time.perf_counter() - tg) * 1e3
    summ = D.summarise(obs, N, a.steps)

    tdev = ("cuda:%d" % local_rank) if backend == "nccl" else "cpu"
    tmax = torch.tensor([dt], dtype=torch.float64, device=tdev)
    per_rank = None
    if world > 1:
        # every rank's own numbers, for the record (a straggler or a slow gather must be visible in SCALE_r*.json): its wall
        # time over the K steps, the device time of its sweep kernels and of its whole run, its side of the gather
        mine = torch.tensor([dt * 1e3 / a.steps, sweep_ms / a.steps, run_ms / a.steps, gather_ms], dtype=torch.float64, device=tdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = np.array([t.cpu().numpy() for t in allr])
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        pe_per_gpu_launchset = nrep * a.steps * 2.0 * N * (N - 1.0)
        value = world * pe_per_gpu_launchset / dt
        S_, W_, _ = eng.geometry
        kform, kname = eng.kernel_form
        algo_bytes_per_launch = pe_per_gpu_launchset * BYTES_PER_PAIR_EVAL / max(launches, 1)
        launch_s = sweep_ms * 1e-3 / max(launches, 1)
        achieved = algo_bytes_per_launch / launch_s / 1e9
        start = "fcc(%d,%d)" % tuple(lattice) + ("+%d sweeps" % a.equilibrate if a.equilibrate else "")
        out = {
            "metric": "reference-equivalent pair-evals/s (MC sweeps/s x replicas x 2N(N-1)) at N=%d" % N,
            "value": value, "unit": "pair-evals/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt * 1e3 / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config 3 per GPU: N=%d LJ + wall (M=3), %d replicas per GPU, %s, L=33 Lz=240 T=A=1.1"
                                   % (N, nrep, start),
                       "N": N, "replicas_per_gpu": nrep, "replicas_total": nrep * world, "gather_lapse": gather_lapse,
                       "replicas_resident_at_once": granule, "geometry": "S=%d x %d wavefront(s)" % (S_, W_),
                       "parallelism": "replicas sharded x%d, %s all-gather of observables at the end"
                                      % (world, "RCCL" if backend == "nccl" else backend)},
            "roofline": None,
            # SURVEY 8d's streaming model (24 B per pair-eval): NOT the binding bound -- positions are register-resident
            "roofline_hbm_model": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": achieved / HBM_PEAK_GBS},
            "device_ms": {"sweep_kernels": sweep_ms, "whole_run": run_ms, "helpers": run_ms - sweep_ms},
            "observables": {"mean_acceptance": summ["mean_acceptance"], "mean_energy": summ["mean_of_meanE"],
                            "replicas_gathered": int(len(obs["accepted"]))},
        }
        if granule_note:
            out["config"]["replica_count_advice"] = granule_note
        if per_rank is not None:
            def stats(col):
                v = per_rank[:, col]
                return {"min": float(v.min()), "median": float(np.median(v)), "max": float(v.max()),
                        "ranks": [float(x) for x in v]}
            out["per_rank"] = {"ms_per_step": stats(0), "sweep_kernel_ms_per_step": stats(1), "device_ms_per_step": stats(2),
                               "gather_ms": stats(3)}
        out["gather_ms"] = gather_ms
        rl = issue_roofline(kname, sweep_ms / a.steps, nrep, N, clock_ghz, start=start)
        base = {"kernel": kname, "launches": launches, "avg_launch_ms": sweep_ms / max(launches, 1),
                # what `rocprofv3 --stats` averages over: the warm-up launches as well
                "avg_launch_ms_incl_warmup": (sweep_ms + warm_ms) / max(launches + warm_launches, 1),
                "ms_per_sweep": sweep_ms / a.steps, "traffic": None}
        if rl is None:   # no committed counters for this kernel / workload: the clock and the time are still live
            rl = {"bound": "valu_issue", "achieved": None, "peak": N_SIMD * clock_ghz if clock_ghz else None,
                  "unit": "G SIMD-cycles/s of VALU issue", "frac": None, "clock_ghz": clock_ghz,
                  "note": "no PMC instruction counts committed for this kernel and workload (profiles/kernel_counters.json)"}
        elif rl.get("hbm"):
            base["traffic"] = rl["hbm"]["bytes_per_sweep"] * (a.steps / max(launches, 1))
        rl.update(base)
        out["roofline"] = rl
        if world == 1 and kform == 2 and not a.no_cpu:
            # for reference, outside the timed region: the same workload through the all-fp64 sweep kernels
            progress("timed region done: %.3f ms per step; all-fp64 kernels" % (dt * 1e3 / a.steps))
            try:
                eng.close()
                p64 = S.default_params(N, nrep, device=local_rank, first_replica=first, tune_kernel=1)
                with S.Engine(p64) as e64:
                    e64.upload(S.fcc_init(*lattice), S.W_REFERENCE)
                    e64.run(0, 1, gather_lapse)
                    nk = min(a.steps, 5)
                    e64.run(0, nk, gather_lapse)
                    ms64, l64 = e64.last_kernel_ms()
                    out["fp64_only_kernels"] = {"kernel": e64.kernel_form[1], "sweeps": nk, "ms_per_sweep": ms64 / nk,
                                                "value": nrep * nk * 2.0 * N * (N - 1.0) / (ms64 * 1e-3)}
            except Exception as e:
                out["fp64_only_kernels"] = {"value": None, "note": "failed: %r" % (e,)}
        zipped = "kernel_mc" in kname or "kernel_mt" in kname or "kernel_ml" in kname
        if world == 1 and not a.no_cpu and zipped:
            # what the timed kernel executed per probe (diagnostic build, sample of the same start)
            eng.close()
            progress("executed-work counters (diagnostic build)")
            ex = executed_work(kname, N, lattice, S_, W_, local_rank, nrep=64 if N <= 4096 else 8)
            out["executed"] = {k: ex[k] for k in SIDE_EXECUTED_KEYS + ("fraction_screened", "sample") if k in ex}
        if world == 1 and not a.no_cpu and N == 4096:
            # the like-for-like kernel: the same workload through sweep_kernel_ma64, whose screen visits EVERY cell for
            # every probe as the reference's loops do (SMC.c:563-578, 597-612)
            try:
                eng.close()
                apk = side_config(S, "the timed workload through the last kernel that tests every pair", N, nrep, lattice, 5,
                                  local_rank, kernel=S.KERNEL_MA, executed=False)
                out["all_pairs_kernel"] = {k: apk[k] for k in ("kernel", "value", "ms_per_sweep")}
            except Exception as e:
                out["all_pairs_kernel"] = {"value": None, "note": "failed: %r" % (e,)}
            # the other single-GPU BASELINE configurations, one adverse state, and the states a PRODUCTION run sits in (the
            # reference thermalises for ~4e6 sweeps, main.c:15-18, SMC.c:110-126): briefly, after the timed region
            out["other_configs"] = []
            for label, n_, r_, lat_, sw_, eq_ in (
                    ("config 3 equilibrated", 4096, 4096, (8, 16), 20, 2000),
                    ("config 2", 1024, 1024, (8, 4), 40, 0),
                    ("config 5 per GPU", 16384, 256, (16, 16), 4, 0),
                    ("config 5 per GPU equilibrating", 16384, 256, (16, 16), 4, 200),
                    ("dense film (adverse)", 4096, 4096, (16, 4), 5, 0)):
                try:
                    out["other_configs"].append(side_config(S, label, n_, r_, lat_, sw_, local_rank, equilibrate=eq_))
                except Exception as e:
                    out["other_configs"].append({"workload": label, "value": None, "note": "failed: %r" % (e,)})
        if world == 1 and not a.no_cpu:
            try:
                # the reference itself where its prebuilt library is at hand (kind "reference"), with the oracle's rate on
                # the same cores beside it; the oracle alone (kind "port") otherwise
                progress("cpu baseline (reference, then oracle port)")
                ref = cpu_baseline_reference(N, *lattice)
                port = cpu_baseline(N, *lattice, seconds_target=6.0 if ref else 12.0)
                if ref:
                    ref["oracle_port"] = {k: port[k] for k in ("value", "per_core")}
                out["cpu_baseline"] = ref or port
            except Exception as e:  # the baseline leg must never take the GPU number down
                out["cpu_baseline"] = {"value": None, "unit": "pair-evals/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        value_full = out["value"]
        out = compact(out)
        out["value"] = value_full
        print(json.dumps(out, separators=(",", ":")), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
