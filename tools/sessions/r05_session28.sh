# round-5 session 28 (through gpurun, repo root): the pre-screen as the default of the product and the diagnostic build -- A/B over
# several configurations against libsmcx_nops.so, the whole GPU suite, the energy soak, the long soak of the diagnostic build
set -o pipefail
mkdir -p gpurun_out
{ for i in 1 2; do
    SMCX_LIB=$PWD/montecarlo-surfacer_amd/libsmcx_nops.so python tools/probes/ps_ab.py
    python tools/probes/ps_ab.py
  done; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ps_ab_configs.txt
echo "ab rc=$?"; cat gpurun_out/r05_ps_ab_configs.txt
python -m pytest tests -q -m gpu > gpurun_out/r05_gputests_ps.log 2>&1
echo "gpu tests rc=$?"; tail -4 gpurun_out/r05_gputests_ps.log
timeout -k 10 600 python tools/soak_energy.py > gpurun_out/r05_soak_energy_ps.txt 2>&1; echo "soak energy rc=$?"; tail -1 gpurun_out/r05_soak_energy_ps.txt
