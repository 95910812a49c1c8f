set -o pipefail
python -m pytest tests -x -q -m gpu > gpurun_out/t7.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/t7.log
tools/gpu_profile.sh b1024 > gpurun_out/prof_b1024.log 2>&1; echo "prof b1024 rc=$?"
tools/gpu_profile.sh b64 --blocks 64 --no-pc > gpurun_out/prof_b64.log 2>&1; echo "prof b64 rc=$?"
tools/gpu_profile.sh randomize_16_20 --randomize 16 20 > gpurun_out/prof_rand.log 2>&1; echo "prof rand rc=$?"
python tools/merkle_levels.py 1024 > gpurun_out/merkle_levels.txt 2>&1; grep -c merkle gpurun_out/merkle_levels.txt
