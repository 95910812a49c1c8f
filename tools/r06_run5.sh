set -o pipefail
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "vp_warm or hundreds_of_deferred or fft_gkr" > gpurun_out/t5.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t5.log
tools/seam_x1024.sh 1024 > gpurun_out/seam1024.log 2>&1; tail -17 gpurun_out/seam1024.log
tools/seam_x1024.sh 64 > gpurun_out/seam64.log 2>&1; tail -14 gpurun_out/seam64.log
tools/gpu_profile.sh b1024 > gpurun_out/prof_b1024.log 2>&1; echo "prof b1024 rc=$?"
tools/gpu_profile.sh b64 --blocks 64 --no-pc > gpurun_out/prof_b64.log 2>&1; echo "prof b64 rc=$?"
tools/gpu_profile.sh randomize_16_20 --randomize 16 20 > gpurun_out/prof_rand.log 2>&1; echo "prof rand rc=$?"
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/bench_default.json
