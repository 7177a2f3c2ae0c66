# round-5 session 5 (through gpurun, repo root): pin the first deviating move of the drift events; the replica-count cliff with windows
set -o pipefail
mkdir -p gpurun_out
# (the saved states of tools/probes/drift_hunt.py, copied from gpurun_out/ into a tracked scratch directory for this one session and removed afterwards)
python tools/probes/drift_pin.py tests/golden/drift_events/r05_drift_3128_25.npz tests/golden/drift_events/r05_drift_1125_28.npz tests/golden/drift_events/r05_drift_1150_32.npz tests/golden/drift_events/r05_drift_2902_46.npz 2>&1 | tee gpurun_out/r05_drift_pin.txt
python tools/probes/replica_cliff.py 2>&1 | tee gpurun_out/r05_replica_cliff.txt
