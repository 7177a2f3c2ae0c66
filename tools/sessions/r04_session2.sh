# round-4 session 2 (through gpurun, repo root): the whole -m gpu suite on the kernels with the two-copy move loop, the row
# registers of the NEXT particles, the cheaper exclusions and the row-layout constant; then the default bench line (its
# `executed` objects now carry further rounds per probe and the lanes-32-apart statistic)
set -o pipefail
python -m pytest tests -x -q -m gpu > gpurun_out/r04_gputests2.log 2>&1; tail -4 gpurun_out/r04_gputests2.log
python bench.py > gpurun_out/r04_bench2.log 2> gpurun_out/r04_bench2.err || { tail -5 gpurun_out/r04_bench2.err; exit 1; }
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04_bench2.log") if l.startswith("{")][0])
print("value %.4e  ms/step %.3f  sweep %.3f" % (j["value"], j["ms_per_step"], j["roofline"]["ms_per_sweep"]))
print("executed", json.dumps(j.get("executed")))
for c in j.get("other_configs", []):
    print(c["workload"][:40], c.get("kernel"), "%.4e" % (c.get("value") or 0), "ms/sweep %.3f" % (c.get("ms_per_sweep") or 0), json.dumps({k: v for k, v in (c.get("executed") or {}).items() if "round" in k or "apart" in k or "candidate" in k}))
PY
