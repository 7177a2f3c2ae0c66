# round-5 session 31 (through gpurun, repo root): the early wait for the displacement (SMCX_GEN_EVM=1, libsmcx_evm.so): GPU suite and
# the energy soak on the variant (the diagnostic build is the product's: the suite's rare-path and soak tests use it as before)
set -o pipefail
mkdir -p gpurun_out
SMCX_LIB=$PWD/montecarlo-surfacer_amd/libsmcx_evm.so python -m pytest tests -q -m gpu > gpurun_out/r05_gputests_evm.log 2>&1
echo "gpu tests rc=$?"; tail -3 gpurun_out/r05_gputests_evm.log
SMCX_LIB=$PWD/montecarlo-surfacer_amd/libsmcx_evm.so timeout -k 10 600 python tools/soak_energy.py > gpurun_out/r05_soak_energy_evm.txt 2>&1; echo "soak rc=$?"; tail -1 gpurun_out/r05_soak_energy_evm.txt
