# round-4 session 4 (through gpurun, repo root): merged-pass variant: A/B bench, then the whole -m gpu suite through the variant
set -o pipefail
LIBS="smcx smcx_mg smcx smcx_mg" tools/ab_bench.sh --steps 10 --warmup 2 > gpurun_out/r04_mg_ab.txt 2>&1
cat gpurun_out/r04_mg_ab.txt
export SMCX_LIB=$PWD/montecarlo-surfacer_amd/libsmcx_mg.so
export SMCX_CHECK_LIB=$PWD/montecarlo-surfacer_amd/libsmcx_mgc.so
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04_mg_tests.log 2>&1; tail -15 gpurun_out/r04_mg_tests.log
