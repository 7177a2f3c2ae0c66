#!/bin/bash
# instruction-fetch and LDS counters of the bench workload (run through gpurun from the repo root):
#   tools/profile_sq2.sh <tag> [bench args...]
set -e
TAG=${1:-run}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 9 --warmup 1 --no-cpu $@"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc_if" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_if.log" 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/pmc_lds" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_lds.log" 2>&1
python3 "$ROOT/tools/pmc_summary.py" "$OUT" | grep -v prepass
