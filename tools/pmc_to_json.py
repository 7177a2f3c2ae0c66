#!/usr/bin/env python3
"""pmc_to_json.py PROF_DIR OUT_JSON [N nrep sweeps [waves_per_replica]] -- turn the rocprofv3 --pmc csv files of
tools/profile_valu.sh into per-kernel counters of ONE full launch of every sweep kernel and per-wave-move figures:
the LARGEST launch (default: the 2-sweep launch between two z sorts of the default profile command, or the one launch per
sweep of the several-wavefront kernels), or with PMC_PICK=last the LAST full launch (within 25 % of the largest) -- for profiles
of `bench.py --equilibrate E`, where the launches in front of the timed ones belong to other states of the system.
bench.py reads the JSON (profiles/kernel_counters.json) for the instruction mix behind its roofline line."""
import csv, glob, json, os, sys

d, out = sys.argv[1], sys.argv[2]
N, nrep, sweeps = (int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (4096, 4096, 9)
wpr = int(sys.argv[6]) if len(sys.argv) > 6 else 1
res = {}
for f in glob.glob(os.path.join(d, "*", "*", "*_counter_collection.csv")):
    best, allv = {}, {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "sweep_kernel" not in k:
            continue
        name = k.split("(")[0].replace("void ", "").strip()
        key = (name, r["Counter_Name"])
        v = float(r["Counter_Value"])
        row = (v, int(r["VGPR_Count"]), int(r["SGPR_Count"]), int(r["Scratch_Size"]), int(r["LDS_Block_Size"]),
               int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        allv.setdefault(key, []).append((int(r["Start_Timestamp"]), row))
        if v >= best.get(key, (-1,))[0]:
            best[key] = row
    if os.environ.get("PMC_PICK") == "last":
        for key, rows in allv.items():
            full = [row for _, row in sorted(rows) if row[5] >= 0.75 * best[key][5]]     # by duration: a full (2-sweep) launch
            best[key] = full[-1]
    for (name, cnt), (v, vg, sg, scr, lds, ns) in best.items():
        e = res.setdefault(name, {"counters": {}, "launch_ns": {}})
        e["counters"][cnt] = v
        e["vgpr"], e["sgpr"], e["scratch_bytes"], e["lds_bytes"] = vg, sg, scr, lds
        e["launch_ns"][cnt] = ns
for name, e in res.items():
    c = e["counters"]
    if "kernel_mt" in name or ("kernel_mc" in name and "x" in name.split("kernel_")[1]):   # several wavefronts per replica:
        sweeps = 1                                                                          # a z sort and a launch per sweep
    elif "kernel_mb" in name or "kernel_mc" in name or "kernel_ml" in name:   # one launch per z sort: tune_resort sweeps (default 2)
        sweeps = int(os.environ.get("SMCX_RESORT", "2"))
    moves = float(nrep) * sweeps * N * wpr              # wave-moves: every wavefront of a replica runs every move
    e["workload"] = {"N": N, "replicas": nrep, "sweeps_in_launch": sweeps, "wave_moves": moves, "waves_per_replica": wpr}
    if "SQ_ACTIVE_INST_VALU" in c and "SQ_BUSY_CYCLES" in c:
        # SQ_ACTIVE_INST_VALU: per SIMD, quad-cycles with a VALU instruction in flight; SQ_BUSY_CYCLES: per SE/XCD busy
        # cycles summed (guide).  Reported against the launch time at the clock bench.py measures, see DESIGN section 6
        e["active_inst_valu_quadcycles"] = c["SQ_ACTIVE_INST_VALU"]
    e["per_wave_move"] = {k: v / moves for k, v in c.items() if k.startswith("SQ_INSTS")}
    if c.get("SQ_WAVE_CYCLES"):
        e["wait_any_frac"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        e["wait_inst_any_frac"] = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # gfx950: FETCH_SIZE counts 64 B per 128-B request of wide coalesced reads (guide, section HBM); these
        # are 8/16-byte scattered and scalar reads, for which the factor is uncalibrated: both readings kept
        e["hbm_bytes_per_sweep"] = {"fetch_x2_plus_write": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / sweeps,
                                    "fetch_x1_plus_write": (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / sweeps}
    e["launch_ms"] = sorted(set(round(v / 1e6, 3) for v in e.pop("launch_ns").values()))
# the identity of the library these counts were taken from (bench.py uses them only for the same build of the kernel)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import smcx_loader
    _S = smcx_loader.load()
    for name, e in res.items():
        e["source_id"] = _S.kernel_source_id(name)
except Exception as ex:   # counts without an identity are never used for a fraction
    print("pmc_to_json: no source ids (%r)" % (ex,))
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
for name, e in res.items():
    print(name, "VGPR", e["vgpr"], "SGPR", e["sgpr"], "scratch", e["scratch_bytes"], "LDS", e["lds_bytes"], "launch ms", e["launch_ms"])
    for k, v in sorted(e["per_wave_move"].items()):
        print("   %-28s %10.2f per wave-move" % (k, v))
    for k in ("SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "FETCH_SIZE", "WRITE_SIZE"):
        if k in e["counters"]:
            print("   %-28s %12.5g" % (k, e["counters"][k]))
