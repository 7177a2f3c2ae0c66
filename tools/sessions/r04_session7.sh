# round-4 session 7 (through gpurun, repo root): the -m gpu suite (sweep_kernel_ml16 in the plan; log-uniform asked for at the
# start of the pass), config 2 through ml16 against the two-team kernel, config 3 with a z sort every second sweep
set -o pipefail
python -m pytest tests -x -q -m gpu > gpurun_out/r04_gputests4.log 2>&1; tail -4 gpurun_out/r04_gputests4.log
for args in "--N 1024 --replicas 1024" "--N 1024 --replicas 1024 --slots 16 --waves 1" "--N 1024 --replicas 1024" "--N 1024 --replicas 1024 --slots 16 --waves 1" "--replicas 4096" "--replicas 4096 --resort 2" "--replicas 4096" "--replicas 4096 --resort 2"; do
python bench.py --no-cpu --steps 40 --warmup 4 $args 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('%-50s %8.4f ms/step  %.4e  sweep %.4f ms  %s' % ('$args', j['ms_per_step'], j['value'], r['ms_per_sweep'], r['kernel']))
"
done | tee gpurun_out/r04_config2_ml16b.txt
