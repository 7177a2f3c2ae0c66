# round-4 session 12 (through gpurun, repo root): exclusions by compare + screen prologue/epilogue trims -- whole GPU suite, then the
# headline workload and config 2 against the build before them (libsmcx_prev.so), twice each in one session
set -o pipefail
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/r04_trims_tests.log 2>&1; tail -4 gpurun_out/r04_trims_tests.log
grep -q "failed\|error" gpurun_out/r04_trims_tests.log && { grep "^FAILED" gpurun_out/r04_trims_tests.log; exit 1; }
for args in "" "--N 1024 --replicas 1024"; do
for lib in smcx smcx_prev smcx smcx_prev; do
SMCX_LIB=$PWD/montecarlo-surfacer_amd/lib$lib.so python bench.py --no-cpu --steps 10 --warmup 2 $args 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('%-24s %-10s %8.4f ms/step  %.4e  sweep %.4f ms  %s' % ('$args', '$lib', j['ms_per_step'], j['value'], r['ms_per_sweep'], r['kernel']))
"
done; done | tee gpurun_out/r04_trims_ab.txt
