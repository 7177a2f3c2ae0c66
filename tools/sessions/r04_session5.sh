# round-4 session 5 (through gpurun, repo root): the -m gpu suite; config 2 through the one-wavefront merged kernel against the
# two-team kernel the geometry rule picks
set -o pipefail
python -m pytest tests -x -q -m gpu > gpurun_out/r04_gputests3.log 2>&1; tail -4 gpurun_out/r04_gputests3.log
for args in "--N 1024 --replicas 1024" "--N 1024 --replicas 1024 --slots 16 --waves 1" "--N 1024 --replicas 1024" "--N 1024 --replicas 1024 --slots 16 --waves 1" "--N 1024 --replicas 2048" "--N 1024 --replicas 2048 --slots 16 --waves 2" "--N 1024 --replicas 512" "--N 1024 --replicas 512 --slots 16 --waves 1"; do
python bench.py --no-cpu --steps 40 --warmup 4 $args 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('%-50s %8.4f ms/step  %.4e  sweep %.4f ms  %s' % ('$args', j['ms_per_step'], j['value'], r['ms_per_sweep'], r['kernel']))
"
done | tee gpurun_out/r04_config2_forms.txt
