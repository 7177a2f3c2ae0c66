# round-5 session 24 (through gpurun, repo root): rand() blocks of the pre-pass through DPP row shifts instead of LDS shuffles --
# A/B of the helpers' time (libsmcx_shfl.so = the same sources with -DSMCX_PREPASS_SHUFFLE), then the whole GPU suite on the new library
set -o pipefail
mkdir -p gpurun_out
{ for i in 1 2; do
    SMCX_LIB=$PWD/montecarlo-surfacer_amd/libsmcx_shfl.so python tools/probes/helpers_probe.py
    python tools/probes/helpers_probe.py
  done; } > gpurun_out/r05_prepass_dpp_ab.txt 2>&1
echo "ab rc=$?"; cat gpurun_out/r05_prepass_dpp_ab.txt
python -m pytest tests -q -m gpu > gpurun_out/r05_gputests_energy.log 2>&1
echo "gpu tests rc=$?"; tail -3 gpurun_out/r05_gputests_energy.log
