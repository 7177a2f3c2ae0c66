# round-5 session 16 (through gpurun, repo root): five trims of the one-wavefront merged kernels (gen_sweep_ma.py T5) -- A/B against the
# library without them (libsmcx_notrim.so), config 3 and config 2; then the whole GPU suite on the product
set -o pipefail
mkdir -p gpurun_out
for lib in smcx_notrim smcx smcx_notrim smcx smcx_notrim smcx; do
SMCX_LIB=$PWD/montecarlo-surfacer_amd/lib$lib.so python bench.py --no-cpu --steps 20 --warmup 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('%-12s config 3  %8.4f ms/step  %.4e  sweep %.4f ms  %s' % ('$lib', j['ms_per_step'], j['value'], r['ms_per_sweep'], r['kernel']))
"
done | tee gpurun_out/r05_trim5_ab.txt
for lib in smcx_notrim smcx smcx_notrim smcx; do
SMCX_LIB=$PWD/montecarlo-surfacer_amd/lib$lib.so python bench.py --no-cpu --steps 40 --warmup 2 --N 1024 --replicas 1024 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('%-12s config 2  %8.4f ms/step  %.4e  sweep %.4f ms  %s' % ('$lib', j['ms_per_step'], j['value'], r['ms_per_sweep'], r['kernel']))
"
done | tee -a gpurun_out/r05_trim5_ab.txt
python -m pytest tests -q -m gpu > gpurun_out/r05_gputests_trim5.log 2>&1
echo "gpu tests rc=$?"; tail -5 gpurun_out/r05_gputests_trim5.log
