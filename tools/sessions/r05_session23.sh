# round-5 session 23 (through gpurun, repo root): smcx_total_energy with the particles ranked by z (total_energy_zk) -- its own
# tests, its duration at the bench's configurations, then the whole GPU suite on the new library
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "total_energy" > gpurun_out/r05_total_energy_tests.log 2>&1
echo "energy tests rc=$?"; tail -3 gpurun_out/r05_total_energy_tests.log
timeout -k 10 300 python tools/probes/total_energy_time.py > gpurun_out/r05_total_energy_time.txt 2>&1; echo "time rc=$?"; cat gpurun_out/r05_total_energy_time.txt
python -m pytest tests -q -m gpu > gpurun_out/r05_gputests_energy.log 2>&1
echo "gpu tests rc=$?"; tail -3 gpurun_out/r05_gputests_energy.log
