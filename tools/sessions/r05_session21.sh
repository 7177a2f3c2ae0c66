# round-5 session 21: wavefronts per SIMD capped by the LDS block size (variants cap4: 16 workgroups per CU, cap3: 12) against the product
set -o pipefail
for lib in smcx smcx_cap4 smcx_cap3; do echo "== lib$lib"; SMCX_LIB=$PWD/montecarlo-surfacer_amd/lib$lib.so python tools/probes/waves_per_simd.py 2>&1; done | tee gpurun_out/r05_waves_per_simd_capped.txt
