# round-5 session 2 (through gpurun, repo root): the whole GPU suite on the new tolerance schedule (+ the full-occupancy test from
# per-replica states), the FETCH_SIZE calibration for 24-byte gathers, the default bench line (slim form, equilibrated entries)
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r05_s2_gputests.log 2>&1
echo "gpu tests rc=$?" | tee -a gpurun_out/r05_s2_gputests.log
tail -4 gpurun_out/r05_s2_gputests.log
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05_gather24 -- $GRAFT_REPO_ROOT/tools/ubench/gather24 > $GRAFT_REPO_ROOT/gpurun_out/r05_gather24_stdout.txt 2>&1 )
python tools/ubench/gather24_report.py gpurun_out/r05_gather24 gpurun_out/r05_gather24_stdout.txt 2>&1 | tee gpurun_out/r05_fetch_size_24B_gather.txt
python bench.py > gpurun_out/r05_bench_default_a.log 2> gpurun_out/r05_bench_default_a.err
echo "bench rc=$? bytes=$(wc -c < gpurun_out/r05_bench_default_a.log)"
