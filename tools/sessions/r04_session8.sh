# round-4 session 8 (through gpurun, repo root): the merged pass in the several-wavefront kernels (mc64x4, mc32x8, mc32x4: one
# exchange per move): the -m gpu suite under a timeout, then config 5's share through them against the two-team kernel
set -o pipefail
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gputests5.log 2>&1; tail -6 gpurun_out/r04_gputests5.log
grep -q "passed" gpurun_out/r04_gputests5.log || exit 1
grep -q "failed" gpurun_out/r04_gputests5.log && exit 1
for args in "--N 16384 --replicas 256" "--N 16384 --replicas 256 --slots 32 --waves 8" "--N 16384 --replicas 256 --slots 64 --waves 4" "--N 16384 --replicas 256" "--N 16384 --replicas 256 --slots 32 --waves 8" "--N 16384 --replicas 512" "--N 16384 --replicas 512 --slots 32 --waves 8"; do
python bench.py --no-cpu --steps 6 --warmup 2 $args 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('%-50s %8.4f ms/step  %.4e  sweep %.4f ms  %s' % ('$args', j['ms_per_step'], j['value'], r['ms_per_sweep'], r['kernel']))
"
done | tee gpurun_out/r04_config5_forms.txt
