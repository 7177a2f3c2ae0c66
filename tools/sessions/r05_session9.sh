# round-5 session 9 (through gpurun, repo root): PMC instruction mix / activity / HBM counters of the sweep kernels of configs 3, 2, 5,
# the dense film and the two production states (30 rocprofv3 --pmc passes, counters only), then the kernel trace of the default bench
set -o pipefail
mkdir -p gpurun_out
bash tools/profile_configs.sh r05 > gpurun_out/r05_profile_configs.log 2>&1
echo "profile_configs rc=$?"; tail -3 gpurun_out/r05_profile_configs.log
bash tools/profile_default.sh r05_default > gpurun_out/r05_kernel_stats_bench_default.txt 2>&1
echo "profile_default rc=$?"; tail -2 gpurun_out/r05_kernel_stats_bench_default.txt | cut -c1-300
