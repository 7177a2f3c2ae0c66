# round-3 evidence run (through gpurun, repo root): PMC counters of configs 3, 2, 5 and the dense film; kernel trace + stats of the default
# bench command; the default bench line; smoke()
set -o pipefail
tools/profile_configs.sh r03g || exit 1
tools/profile_default.sh r03g > gpurun_out/r03_kernel_stats_bench_default.txt 2>&1 || exit 1
python bench.py > gpurun_out/r03_bench_default.log 2> gpurun_out/r03_bench_default.err || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke.log 2>&1 || { tail -5 gpurun_out/r03_smoke.log; exit 1; }
tail -3 gpurun_out/r03_smoke.log
