# round-4 session 21 (through gpurun, repo root): several-wavefront merged kernels with the exclusions by compare -- their tests, then
# N = 16384 x 512 and N = 8192 x 1024 against the build before (libsmcx_prev.so)
set -o pipefail
timeout -k 10 800 python -m pytest tests -q -m gpu -k "config5 or several_wavefront or x4 or x8 or 9216 or 6144 or ragged or 32-4 or 64-4 or 32-8 or screen_ab" > gpurun_out/r04_xcw_tests.log 2>&1; tail -3 gpurun_out/r04_xcw_tests.log
grep -q "failed\|error" gpurun_out/r04_xcw_tests.log && { grep "^FAILED" gpurun_out/r04_xcw_tests.log; exit 1; }
for args in "--N 16384 --replicas 512"; do
for lib in smcx smcx_prev smcx smcx_prev; do
SMCX_LIB=$PWD/montecarlo-surfacer_amd/lib$lib.so python bench.py --no-cpu --steps 6 --warmup 2 $args 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('%-40s %-10s %8.4f ms/step  %.4e  sweep %.4f ms  %s' % ('$args', '$lib', j['ms_per_step'], j['value'], r['ms_per_sweep'], r['kernel']))
"
done; done | tee gpurun_out/r04_xcw_ab.txt
