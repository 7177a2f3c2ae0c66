# round-5 session 5 (through gpurun, repo root): pin the first deviating move of the drift events; the replica-count cliff with windows
set -o pipefail
mkdir -p gpurun_out
cp tests/golden/drift_events/*.npz gpurun_out/ 2>/dev/null
python tools/probes/drift_pin.py tests/golden/drift_events/r05_drift_3128_25.npz tests/golden/drift_events/r05_drift_1125_28.npz tests/golden/drift_events/r05_drift_1150_32.npz tests/golden/drift_events/r05_drift_2902_46.npz 2>&1 | tee gpurun_out/r05_drift_pin.txt
python tools/probes/replica_cliff.py 2>&1 | tee gpurun_out/r05_replica_cliff.txt
