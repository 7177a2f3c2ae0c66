# round-5 session 8 (through gpurun, repo root): the two-team kernel with team B's wall lanes / side pair on other slabs' waves
# (variants tt12: walls on slab 2, side pair on slab 1; tt01; tt20) -- parity against the oracle, then config 5's time per sweep, A/B/A/B
set -o pipefail
mkdir -p gpurun_out
for lib in tt12 tt01 tt20; do
  echo "== parity, lib$lib" | tee -a gpurun_out/r05_tt_slabs.txt
  SMCX_LIB=$PWD/montecarlo-surfacer_amd/libsmcx_$lib.so python -m pytest tests/test_gpu_rare_paths.py tests/test_gpu_configs.py -q -m gpu -k "mt64x8 and not two_team_list and not windows and not byte_screen" 2>&1 | tail -2 | tee -a gpurun_out/r05_tt_slabs.txt
done
for lib in smcx smcx_tt12 smcx_tt01 smcx_tt20 smcx smcx_tt12 smcx_tt01 smcx_tt20; do
SMCX_LIB=$PWD/montecarlo-surfacer_amd/lib$lib.so python bench.py --no-cpu --steps 8 --warmup 2 --N 16384 --replicas 256 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('%-12s %8.4f ms/step  %.4e  sweep %.4f ms  %s' % ('$lib', j['ms_per_step'], j['value'], r['ms_per_sweep'], r['kernel']))
"
done | tee -a gpurun_out/r05_tt_slabs.txt
python tools/probes/replica_cliff.py --gl 40 --steps 40 2>&1 | tee gpurun_out/r05_replica_cliff_gl40.txt
