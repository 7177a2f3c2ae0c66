# round-5 session 6 (through gpurun, repo root): which feature of the kernel produces the phantom neighbour?  The first drift event through
# other kernel forms (tune_kernel) and through generator variants of sweep_kernel_mc64
set -o pipefail
F="tests/golden/drift_events/r05_drift_3128_25.npz tests/golden/drift_events/r05_drift_1150_32.npz"
for k in 7 6 5 4 1; do echo "== tune_kernel $k"; SMCX_PIN_KERNEL=$k python tools/probes/drift_pin.py $F 2>&1 | grep '"file"' | cut -c1-330; done | tee gpurun_out/r05_drift_bisect.txt
for v in nomerge noxc nopeel; do echo "== variant $v"; SMCX_LIB=$PWD/montecarlo-surfacer_amd/libsmcx_$v.so python tools/probes/drift_pin.py $F 2>&1 | grep '"file"' | cut -c1-330; done | tee -a gpurun_out/r05_drift_bisect.txt
