#!/bin/bash
# SQ counters of the bench workload for one geometry (run through gpurun from the repo root):
#   tools/profile_sq.sh <tag> [bench args...]        e.g.  SMCX_MX=1 tools/profile_sq.sh mx64 --slots 64 --waves 1
set -e
TAG=${1:-run}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 9 --warmup 1 --no-cpu $@"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD --output-format csv -d "$OUT/pmc_sq2" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_sq2.log" 2>&1
python3 "$ROOT/tools/pmc_summary.py" "$OUT" | grep -v prepass
