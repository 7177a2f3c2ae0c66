# round-5 session 29 (run twice: after the pre-screen and again after the early displacement wait): the record on the final library --
# source ids) -- PMC passes, smoke, default bench, kernel trace of the default bench, 200-step bench
set -o pipefail
mkdir -p gpurun_out
bash tools/profile_configs.sh r05 > gpurun_out/r05_profile_configs.log 2>&1
echo "profile_configs rc=$?"; tail -2 gpurun_out/r05_profile_configs.log | cut -c1-300
cp gpurun_out/kernel_counters_r05.json profiles/kernel_counters.json
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 800 python bench.py > gpurun_out/r05_bench_default.log 2> gpurun_out/r05_bench_default.err
echo "bench rc=$? bytes=$(wc -c < gpurun_out/r05_bench_default.log)"
bash tools/profile_default.sh r05_default > gpurun_out/r05_kernel_stats_bench_default.txt 2>&1; echo "stats rc=$?"
timeout -k 10 300 python bench.py --steps 200 --no-cpu > gpurun_out/r05_bench_200_steps.log 2> gpurun_out/r05_bench_200.err; echo "bench200 rc=$?"
