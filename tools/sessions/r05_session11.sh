# round-5 session 11 (through gpurun, repo root): the record on the round's kernels -- whole GPU suite, smoke, default bench line (rooflines
# from the committed counters), 200-step line
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/r05_gputests_final.log 2>&1
echo "gpu tests rc=$?"; tail -4 gpurun_out/r05_gputests_final.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1
echo "smoke rc=$?"; tail -2 gpurun_out/r05_smoke.log
timeout -k 10 800 python bench.py > gpurun_out/r05_bench_default.log 2> gpurun_out/r05_bench_default.err
echo "bench rc=$? bytes=$(wc -c < gpurun_out/r05_bench_default.log)"
timeout -k 10 300 python bench.py --steps 200 --warmup 2 --no-cpu > gpurun_out/r05_bench_200_steps.log 2> gpurun_out/r05_bench_200_steps.err
echo "bench200 rc=$?"; cut -c1-300 gpurun_out/r05_bench_200_steps.log
