# round-5 session 27 (through gpurun, repo root): the pre-screen of the next move's probe B inside the fetch wait (generator switch
# SMCX_GEN_PS=1, libsmcx_ps.so): timing against the product, alternating, then the GPU suite on the variant
set -o pipefail
mkdir -p gpurun_out
{ for i in 1 2; do
    SMCX_LIB=$PWD/montecarlo-surfacer_amd/libsmcx_ps.so python tools/probes/helpers_probe.py
    python tools/probes/helpers_probe.py
  done; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ps_ab.txt
echo "ab rc=$?"; cat gpurun_out/r05_ps_ab.txt
SMCX_LIB=$PWD/montecarlo-surfacer_amd/libsmcx_ps.so python -m pytest tests -q -m gpu -x > gpurun_out/r05_gputests_ps.log 2>&1
echo "gpu tests rc=$?"; tail -5 gpurun_out/r05_gputests_ps.log
