# round-5 session 1 (through gpurun, repo root): the list-overflow fix of the two-team kernel (tests + evidence against the round-4
# generator), then where a production run spends its time: config 3 and config 5's share over thousands of sweeps
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_rare_paths.py tests/test_gpu_configs.py -x -q -m gpu -k "mt64x8 or two_team_list" > gpurun_out/r05_s1_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r05_s1_tests.log
tail -5 gpurun_out/r05_s1_tests.log
python tools/probes/tt_list_overflow.py 2>&1 | tee gpurun_out/r05_two_team_list_overflow.txt
python tools/probes/equil_probe.py 4096 4096 8 16 25,75,100,300,500,1000,2000,4000 --save gpurun_out/r05_equil_c3:64 2>&1 | tee gpurun_out/r05_equil_config3.txt
python tools/probes/equil_probe.py 16384 256 16 16 4,46,150,300,500 --save gpurun_out/r05_equil_c5:8 2>&1 | tee gpurun_out/r05_equil_config5.txt
