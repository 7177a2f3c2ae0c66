# round-4 session 11 (through gpurun, repo root): two timing probes of the two-team kernel at config 5 -- streaming hint on the
# displacement loads (libsmcx_nt.so), and every candidate fetch reading cell 0 (libsmcx_fake.so: TIMING ONLY, wrong results)
set -o pipefail
for lib in smcx smcx_nt smcx_fake smcx smcx_nt smcx_fake; do
SMCX_LIB=$PWD/montecarlo-surfacer_amd/lib$lib.so python bench.py --no-cpu --steps 6 --warmup 2 --N 16384 --replicas 256 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('%-12s %8.4f ms/step  %.4e  sweep %.4f ms  %s  acceptance %.4f' % ('$lib', j['ms_per_step'], j['value'], r['ms_per_sweep'], r['kernel'], j['observables']['mean_acceptance']))
"
done | tee gpurun_out/r04_config5_memory_probes.txt
