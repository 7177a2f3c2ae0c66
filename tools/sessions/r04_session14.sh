# round-4 session 14 (through gpurun, repo root): the trimmed kernels against the build before them (libsmcx_prev.so): the kernel-against-
# kernel test that failed on rounding drift, then the headline workload, config 2 and config 5, twice each in one session
set -o pipefail
timeout -k 10 600 python -m pytest tests -q -m gpu -k "hand_scheduled_kernel_matches" > gpurun_out/r04_trims_tests2.log 2>&1; tail -3 gpurun_out/r04_trims_tests2.log
grep -q "failed\|error" gpurun_out/r04_trims_tests2.log && exit 1
for args in "" "--N 1024 --replicas 1024" "--N 16384 --replicas 256"; do
for lib in smcx smcx_prev smcx smcx_prev; do
SMCX_LIB=$PWD/montecarlo-surfacer_amd/lib$lib.so python bench.py --no-cpu --steps 10 --warmup 2 $args 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('%-26s %-10s %8.4f ms/step  %.4e  sweep %.4f ms  %s' % ('$args', '$lib', j['ms_per_step'], j['value'], r['ms_per_sweep'], r['kernel']))
"
done; done | tee gpurun_out/r04_trims_ab.txt
