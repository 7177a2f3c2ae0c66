set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_ref_golden.py -m gpu -q -k "ragged or byte_screen_kernel or multi_gpu or smcx_main or chain_against" > gpurun_out/r03_gputests4.log 2>&1
tail -12 gpurun_out/r03_gputests4.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
$R/tools/ubench/active_valu > $R/gpurun_out/r03_active_valu.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES --output-format csv -d $R/gpurun_out/prof_active_valu -- $R/tools/ubench/active_valu >> $R/gpurun_out/r03_active_valu.log 2>&1
cat $R/gpurun_out/r03_active_valu.log
cd $R
for r in 2048 4096 4097 6144 8192; do python bench.py --no-cpu --replicas $r --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('replicas', d['config']['replicas_per_gpu'], 'ms_per_step %.3f'%d['ms_per_step'], 'value %.4g'%d['value'])"; done > gpurun_out/r03_replica_cliff.log 2>&1
cat gpurun_out/r03_replica_cliff.log
for n in 8192; do python bench.py --no-cpu --N 8192 --replicas 512 --steps 10 2>&1 | tail -c 400; done
