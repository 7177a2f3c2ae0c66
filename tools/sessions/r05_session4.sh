# round-5 session 4 (through gpurun, repo root): the drift hunt on the fixed wrappers; the window-launch tests; the whole GPU suite
set -o pipefail
mkdir -p gpurun_out
python tools/probes/drift_hunt.py 500 10 4096 2>&1 | cut -c1-600 | tee gpurun_out/r05_drift_hunt.txt
python -m pytest tests/test_gpu_configs.py -q -m gpu -k "windows_of_units" > gpurun_out/r05_s4_windows.log 2>&1
echo "window tests rc=$?"; tail -5 gpurun_out/r05_s4_windows.log
python -m pytest tests -q -m gpu --deselect tests/test_gpu_configs.py::test_config3_full_occupancy_from_equilibrating_per_replica_states > gpurun_out/r05_s4_gputests.log 2>&1
echo "gpu tests rc=$?"; tail -8 gpurun_out/r05_s4_gputests.log
