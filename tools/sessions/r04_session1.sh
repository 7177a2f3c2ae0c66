# round-4 session 1 (through gpurun, repo root): the instruction-cost experiment (40 padding instructions of one kind per move in
# sweep_kernel_mc64, builds without the two-copy move loop) and the two-copy ("peeled") move loop against the oracle
set -o pipefail
LIBS="smcx_nopeel smcx_pad_S40 smcx_pad_V40 smcx_pad_W40 smcx_pad_N40 smcx_pad_D40 smcx smcx_nopeel" tools/ab_bench.sh --steps 10 --warmup 2 > gpurun_out/r04_pad.txt 2>&1 || exit 1
cat gpurun_out/r04_pad.txt
python -m pytest tests/test_gpu_configs.py tests/test_gpu_rare_paths.py -x -q -m gpu > gpurun_out/r04_peel_tests.log 2>&1; tail -4 gpurun_out/r04_peel_tests.log
