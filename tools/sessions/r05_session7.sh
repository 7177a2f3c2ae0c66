# round-5 session 7 (through gpurun, repo root): the z sort's sentinel fix -- the regression test against a library built BEFORE the fix
# (libsmcx_nowin.so: must fail) and against the product; the drift hunt again (no event expected); whole GPU suite; default bench
set -o pipefail
mkdir -p gpurun_out
echo "== regression test through the library built before the fix (expected: failures)" | tee gpurun_out/r05_s7_regress.txt
SMCX_LIB=$PWD/montecarlo-surfacer_amd/libsmcx_nowin.so SMCX_ALLOW_ENV_TUNING=1 python -m pytest tests/test_gpu_configs.py -q -m gpu -k "last_particle_in_the_top_corner" 2>&1 | grep -E "passed|failed|FAILED|assert .*jj|worst" | cut -c1-260 | tee -a gpurun_out/r05_s7_regress.txt
echo "== the same through the product library" | tee -a gpurun_out/r05_s7_regress.txt
python -m pytest tests/test_gpu_configs.py -q -m gpu -k "last_particle_in_the_top_corner" 2>&1 | tail -3 | tee -a gpurun_out/r05_s7_regress.txt
python tools/probes/drift_hunt.py 500 10 4096 2>&1 | cut -c1-400 | grep -v '"changed_in_this_chunk": \[\]' | tee gpurun_out/r05_drift_hunt_after_fix.txt
python tools/probes/drift_hunt.py 200 10 256 16384 16 16 2>&1 | cut -c1-400 | grep -v '"changed_in_this_chunk": \[\]' | tee -a gpurun_out/r05_drift_hunt_after_fix.txt
python -m pytest tests -q -m gpu > gpurun_out/r05_s7_gputests.log 2>&1
echo "gpu tests rc=$?"; tail -6 gpurun_out/r05_s7_gputests.log
timeout -k 10 700 python bench.py > gpurun_out/r05_bench_default_b.log 2> gpurun_out/r05_bench_default_b.err
echo "bench rc=$? bytes=$(wc -c < gpurun_out/r05_bench_default_b.log)"
