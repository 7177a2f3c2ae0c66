# round-5 session 25 (through gpurun, repo root): the record on the final library (total_energy_zk, DPP rand blocks): smoke, default
# bench, kernel trace of the default bench.  The sweep kernels and their source ids are those of session 22 (PMC passes not repeated).
set -o pipefail
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 800 python bench.py > gpurun_out/r05_bench_default.log 2> gpurun_out/r05_bench_default.err
echo "bench rc=$? bytes=$(wc -c < gpurun_out/r05_bench_default.log)"
bash tools/profile_default.sh r05_default > gpurun_out/r05_kernel_stats_bench_default.txt 2>&1; echo "stats rc=$?"
timeout -k 10 600 python bench.py --steps 200 --no-cpu > gpurun_out/r05_bench_200_steps.log 2> gpurun_out/r05_bench_200.err; echo "bench200 rc=$?"
