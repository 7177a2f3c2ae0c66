# round-4 session 13 (through gpurun, repo root): rounding drift between two kernels at the ragged N = 4000 case, this build and the one before
set -o pipefail
for lib in smcx smcx_prev; do echo "== lib$lib.so"; SMCX_LIB=$PWD/montecarlo-surfacer_amd/lib$lib.so timeout -k 10 500 python tools/probes/ragged_divergence.py 2>&1 | grep -v Warning; done | tee gpurun_out/r04_ragged_divergence.txt
