# round-5 session 30 (through gpurun, repo root): whole GPU suite and the long soak of the diagnostic build on the library with the pre-screen
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/r05_gputests_final.log 2>&1
echo "gpu tests rc=$?"; tail -3 gpurun_out/r05_gputests_final.log
timeout -k 10 700 python tools/soak_check.py long > gpurun_out/r05_soak_check_long_ps.txt 2>&1; echo "soak check rc=$?"; tail -8 gpurun_out/r05_soak_check_long_ps.txt | cut -c1-230
