# round-5 session 17 (through gpurun, repo root): TIMING-ONLY ablations of sweep_kernel_mc64 (wrong results by construction): what does a
# phase of the move cost the running kernel, measured by leaving it out?  Every variant rejects every move (the state stays the
# lattice start, so the candidates are the same in all of them); config 3, sweep kernel ms per sweep.
set -o pipefail
mkdir -p gpurun_out
for lib in smcx smcx_abl_noaccept smcx_abl_na_nobody smcx_abl_na_nosA smcx_abl_na_nosB smcx_abl_na_nosAB smcx_abl_na_nosAB_nobody smcx smcx_abl_noaccept; do
SMCX_LIB=$PWD/montecarlo-surfacer_amd/lib$lib.so python bench.py --no-cpu --steps 10 --warmup 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('%-28s sweep kernel %8.4f ms per sweep   acceptance %.3f' % ('$lib', r['ms_per_sweep'], j['observables']['mean_acceptance']))
"
done | tee gpurun_out/r05_ablation_mc64.txt
