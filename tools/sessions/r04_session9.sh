# round-4 session 9 (through gpurun, repo root): the two-team kernel with the hand-over list (TL) against the oracle and against
# round 3's form (libsmcx_nottl.so) at config 5
set -o pipefail
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "mt64x8 or config5 or 64-8 or 9216 or several_wavefront" > gpurun_out/r04_tl_tests.log 2>&1; tail -5 gpurun_out/r04_tl_tests.log
grep -q "failed" gpurun_out/r04_tl_tests.log && exit 1
for lib in smcx smcx_nottl smcx smcx_nottl; do
SMCX_LIB=$PWD/montecarlo-surfacer_amd/lib$lib.so python bench.py --no-cpu --steps 6 --warmup 2 --N 16384 --replicas 256 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('%-12s %8.4f ms/step  %.4e  sweep %.4f ms  %s' % ('$lib', j['ms_per_step'], j['value'], r['ms_per_sweep'], r['kernel']))
"
done | tee gpurun_out/r04_tl_config5.txt
