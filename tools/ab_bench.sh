#!/bin/bash
# ab_bench.sh [bench args] -- the default bench workload through every montecarlo-surfacer_amd/libsmcx*.so
# variant named in $LIBS (space separated, without the lib prefix / .so suffix; "smcx" = the product), in ONE
# session so that the numbers are comparable.  Variants are loaded through SMCX_LIB, never copied over the product.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for v in ${LIBS:-smcx}; do
    lib=$ROOT/montecarlo-surfacer_amd/lib$v.so
    [ "$v" = smcx ] || [ -f "$lib" ] || { echo "$v: not built"; continue; }
    SMCX_LIB=$lib python3 $ROOT/bench.py --no-cpu "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        r = j['roofline']
        print('%-14s %8.3f ms/step  %.4e pair-evals/s  sweep %.3f ms  clock %.3f GHz  %s' % ('$v', j['ms_per_step'], j['value'], r['ms_per_sweep'], r.get('clock_ghz') or 0, r['kernel']))
"
done
