# round-4 session 22 (through gpurun, repo root): team A's priority level during its probe (3 = the product, 2, 1; team B always 3) at config 5
set -o pipefail
for lib in smcx smcx_allA2 smcx_allA1 smcx smcx_allA2 smcx_allA1; do
SMCX_LIB=$PWD/montecarlo-surfacer_amd/lib$lib.so python bench.py --no-cpu --steps 6 --warmup 2 --N 16384 --replicas 256 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('%-12s %8.4f ms/step  %.4e  sweep %.4f ms  %s' % ('$lib', j['ms_per_step'], j['value'], r['ms_per_sweep'], r['kernel']))
"
done | tee gpurun_out/r04_team_a_priority.txt
