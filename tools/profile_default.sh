#!/bin/bash
# kernel trace + stats of the DEFAULT bench command (run through gpurun from the repo root):
#   tools/profile_default.sh <tag>     -> gpurun_out/prof_<tag>/stats/...  and the bench line of that same run
set -e
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --no-cpu > "$OUT/bench_traced.log" 2>&1
f=$(find "$OUT/stats" -name "*_kernel_stats.csv" | head -1)
echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu   (default workload, 2 warm-up + 20 timed sweeps)"
cat "$f"
t=$(find "$OUT/stats" -name "*_kernel_trace.csv" | head -1)
echo "# per-launch durations of the sweep kernel (ns), from the kernel trace"
python3 - "$t" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "sweep_kernel" in r["Kernel_Name"]:
        print(r["Kernel_Name"].split("(")[0], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
PY
echo "# the bench line of this traced run"
grep '^{' "$OUT/bench_traced.log"
