#!/bin/bash
# VALU instruction mix and issue counters of the bench workload (run through gpurun from the repo root):
#   tools/profile_valu.sh <tag> [bench args...]
# Three rocprofv3 --pmc passes (8 SQ counters each, no trace domains); summary printed by tools/pmc_summary.py.
set -e
TAG=${1:-run}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 9 --warmup 1 --no-cpu $@"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES --output-format csv -d "$OUT/pmc_mix" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_mix.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_cls" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_cls.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/pmc_act" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_act.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1
python3 "$ROOT/tools/pmc_summary.py" "$OUT" | grep -v prepass
# PMC_N / PMC_NREP / PMC_WPR: the workload of the bench args (default: config 3, one wavefront per replica)
python3 "$ROOT/tools/pmc_to_json.py" "$OUT" "$OUT/kernel_counters.json" ${PMC_N:-4096} ${PMC_NREP:-4096} 9 ${PMC_WPR:-1}
