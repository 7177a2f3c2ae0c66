# round-4 evidence run A (through gpurun, repo root): PMC counters of configs 3, 2, 5 and the dense film (with the source ids of
# this build), kernel trace + stats of the default bench command, the default bench line, the 200-step bench line
set -o pipefail
tools/profile_configs.sh r04 > gpurun_out/r04_profile_configs.log 2>&1 || { tail -20 gpurun_out/r04_profile_configs.log; exit 1; }
cp gpurun_out/kernel_counters_r04.json profiles/kernel_counters.json
tools/profile_default.sh r04 > gpurun_out/r04_kernel_stats_bench_default.txt 2>&1 || { tail -20 gpurun_out/r04_kernel_stats_bench_default.txt; exit 1; }
python bench.py > gpurun_out/r04_bench_default.log 2> gpurun_out/r04_bench_default.err || { tail -5 gpurun_out/r04_bench_default.err; exit 1; }
python bench.py --steps 200 --warmup 5 --no-cpu > gpurun_out/r04_bench_200_steps.log 2> gpurun_out/r04_bench_200_steps.err || exit 1
python - <<'PY'
import json
for f in ("gpurun_out/r04_bench_default.log", "gpurun_out/r04_bench_200_steps.log"):
    j = json.loads([l for l in open(f) if l.startswith("{")][0])
    r = j["roofline"]
    print(f, "value %.4e ms/step %.3f sweep %.3f frac %s clock %s" % (j["value"], j["ms_per_step"], r["ms_per_sweep"], r.get("frac"), r.get("clock_ghz")))
    for c in j.get("other_configs", []):
        print("   ", c["workload"][:44], c.get("kernel"), "%.4e" % (c.get("value") or 0), "ms/sweep %.3f" % (c.get("ms_per_sweep") or 0), "frac", (c.get("roofline") or {}).get("frac"))
    if "cpu_baseline" in j: print("    cpu", j["cpu_baseline"].get("value"), j["cpu_baseline"].get("kind"), j["cpu_baseline"].get("cores"))
PY
python tools/soak_stats.py > gpurun_out/r04_soak_stats.txt 2>&1; tail -6 gpurun_out/r04_soak_stats.txt
