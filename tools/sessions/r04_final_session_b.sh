# round-4 evidence run B (through gpurun, repo root): replica counts around the chip's wave slots, the soak of the diagnostic
# build, the whole -m gpu suite, smoke()
set -o pipefail
for r in 2048 4096 4097 6144 8192; do
python bench.py --no-cpu --steps 10 --warmup 2 --replicas $r 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); c = j['config']
        print('%5d replicas: %8.3f ms per step  %.4e pair-evals/s   resident at once: %s   %s' % ($r, j['ms_per_step'], j['value'], c.get('replicas_resident_at_once'), c.get('replica_count_advice') or ''))
"
done | tee gpurun_out/r04_replica_cliff.txt
python tools/soak_check.py > gpurun_out/r04_soak_check.txt 2>&1; tail -3 gpurun_out/r04_soak_check.txt
python -m pytest tests -x -q -m gpu > gpurun_out/r04_gputests_final.log 2>&1; tail -4 gpurun_out/r04_gputests_final.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.log 2>&1; tail -3 gpurun_out/r04_smoke.log
