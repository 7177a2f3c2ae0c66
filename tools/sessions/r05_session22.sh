# round-5 session 22 (through gpurun, repo root): the record on the FINAL sources -- PMC passes again (the kernels are the same; their
# source ids changed with smcx_sweep_ma.hip), whole GPU suite, smoke, default bench, kernel trace of the default bench
set -o pipefail
mkdir -p gpurun_out
bash tools/profile_configs.sh r05 > gpurun_out/r05_profile_configs.log 2>&1
echo "profile_configs rc=$?"; tail -2 gpurun_out/r05_profile_configs.log | cut -c1-300
cp gpurun_out/kernel_counters_r05.json profiles/kernel_counters.json
python -m pytest tests -q -m gpu > gpurun_out/r05_gputests_final.log 2>&1
echo "gpu tests rc=$?"; tail -3 gpurun_out/r05_gputests_final.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 800 python bench.py > gpurun_out/r05_bench_default.log 2> gpurun_out/r05_bench_default.err
echo "bench rc=$? bytes=$(wc -c < gpurun_out/r05_bench_default.log)"
bash tools/profile_default.sh r05_default > gpurun_out/r05_kernel_stats_bench_default.txt 2>&1; echo "stats rc=$?"
