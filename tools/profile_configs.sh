#!/bin/bash
# PMC instruction mix, activity and HBM counters of the sweep kernel of BASELINE configs 3, 2 and 5 (per-GPU share):
#   tools/profile_configs.sh <tag>      (through gpurun, from the repo root; 30 rocprofv3 --pmc passes, counters only)
# Writes gpurun_out/prof_<tag>_c{3,2,5}/ and merges the three kernel_counters.json into
# gpurun_out/kernel_counters_<tag>.json (copy to profiles/kernel_counters.json to make bench.py use it).
set -e
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
tools/profile_valu.sh ${TAG}_c3 > gpurun_out/pmc_${TAG}_c3.txt 2>&1
PMC_N=1024 PMC_NREP=1024 PMC_WPR=1 tools/profile_valu.sh ${TAG}_c2 --N 1024 --replicas 1024 > gpurun_out/pmc_${TAG}_c2.txt 2>&1
PMC_N=16384 PMC_NREP=256 PMC_WPR=8 tools/profile_valu.sh ${TAG}_c5 --N 16384 --replicas 256 > gpurun_out/pmc_${TAG}_c5.txt 2>&1
# the adverse start of other_configs (dense film, most groups in reach): same kernel as config 3, keyed "kernel@start"
tools/profile_valu.sh ${TAG}_dense --lattice 16,4 > gpurun_out/pmc_${TAG}_dense.txt 2>&1
# the states a production run sits in (round 5): config 3 after 2000 sweeps, config 5's share after 200; the LAST full launch counts
PMC_PICK=last tools/profile_valu.sh ${TAG}_c3eq --equilibrate 2000 > gpurun_out/pmc_${TAG}_c3eq.txt 2>&1
PMC_PICK=last PMC_N=16384 PMC_NREP=256 PMC_WPR=8 tools/profile_valu.sh ${TAG}_c5eq --N 16384 --replicas 256 --equilibrate 200 > gpurun_out/pmc_${TAG}_c5eq.txt 2>&1
python3 - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
out = {}
START = {"c3": "fcc(8,16)", "c2": "fcc(8,4)", "c5": "fcc(16,16)", "dense": "fcc(16,4)", "c3eq": "fcc(8,16)+2000 sweeps",
         "c5eq": "fcc(16,16)+200 sweeps"}
for c in ("c3", "c2", "c5", "dense", "c3eq", "c5eq"):
    d = json.load(open("gpurun_out/prof_%s_%s/kernel_counters.json" % (tag, c)))
    for k, v in d.items():
        v["source"] = "tools/profile_configs.sh %s (rocprofv3 --pmc, 5 passes of bench.py --steps 9 --warmup 1 --no-cpu), %s" % (
            tag, "config " + c[1] if c != "dense" else "dense film")
        v["workload"]["start"] = START[c]
        out[k if c in ("c3", "c2", "c5") else k + "@" + START[c]] = v
# the opcode-level split of the 32-bit VALU instructions of sweep_kernel_mc64 (fast / slow issue forms), from the generated body
try:
    import re, subprocess
    txt = subprocess.run([sys.executable, "tools/phase_table.py", "montecarlo-surfacer_amd/csrc/build/smcx_sweep_mc_body64.inc"],
                         capture_output=True, text=True).stdout
    m = re.search(r"32-bit VALU: ([\d.]+) fast \+ ([\d.]+) slow", txt)
    fast, slow = float(m.group(1)), float(m.group(2))
    for k, v in out.items():
        if k.startswith("smcx::sweep_kernel_mc64"):
            v["valu32_fast_fraction_static"] = round(fast / (fast + slow), 4)
            v["valu32_fast_fraction_source"] = "tools/phase_table.py over the generated steady copy: %.1f fast + %.1f slow 32-bit VALU forms per move" % (fast, slow)
except Exception as ex:
    print("no static fast/slow split:", ex)
json.dump(out, open("gpurun_out/kernel_counters_%s.json" % tag, "w"), indent=1, sort_keys=True)
print("kernels:", sorted(out))
PY
