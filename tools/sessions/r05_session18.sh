# round-5 session 18 (through gpurun, repo root): TIMING-ONLY ablations of the ACCEPT path of sweep_kernel_mc64 (wrong results by construction):
# moves are accepted as usual but parts of what an accepted move does are left out
set -o pipefail
mkdir -p gpurun_out
for lib in smcx smcx_abl_nostR smcx_abl_nostRs smcx_abl_nostRRs smcx_abl_nogpr smcx_abl_nop0 smcx_abl_noall smcx_abl_noaccept smcx; do
SMCX_LIB=$PWD/montecarlo-surfacer_amd/lib$lib.so python bench.py --no-cpu --steps 10 --warmup 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('%-28s sweep kernel %8.4f ms per sweep   acceptance %.3f' % ('$lib', r['ms_per_sweep'], j['observables']['mean_acceptance']))
"
done | tee gpurun_out/r05_ablation_accept_mc64.txt
