# round-5 session 3 (through gpurun, repo root): hunt for the replicas whose incremental energy parts from the recomputed one;
# the FETCH_SIZE calibration (asm constraints fixed); the default bench line with progress on stderr
set -o pipefail
mkdir -p gpurun_out
python tools/probes/drift_hunt.py 500 10 4096 2>&1 | tee gpurun_out/r05_drift_hunt.txt
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05_gather24 -- $GRAFT_REPO_ROOT/tools/ubench/gather24 > $GRAFT_REPO_ROOT/gpurun_out/r05_gather24_stdout.txt 2>&1 ) &&
python tools/ubench/gather24_report.py gpurun_out/r05_gather24 gpurun_out/r05_gather24_stdout.txt 2>&1 | tee gpurun_out/r05_fetch_size_24B_gather.txt &&
timeout -k 10 700 python bench.py > gpurun_out/r05_bench_default_a.log 2> gpurun_out/r05_bench_default_a.err
echo "bench rc=$? bytes=$(wc -c < gpurun_out/r05_bench_default_a.log)"; tail -3 gpurun_out/r05_bench_default_a.err
