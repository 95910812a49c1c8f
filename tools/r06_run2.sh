set -o pipefail
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fri_commit_phase or protocol_pass or deferred or commit_public or full_transcript_with_commitment or reference_binary or reference_verifier or complete_protocol or tensor or plan_tuner" > gpurun_out/t2.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t2.log
tools/seam_x1024.sh 1024 > gpurun_out/seam1024.log 2>&1; tail -16 gpurun_out/seam1024.log
tools/seam_x1024.sh 64 > gpurun_out/seam64.log 2>&1; tail -12 gpurun_out/seam64.log
tools/gpu_profile.sh b1024 > gpurun_out/prof_b1024.log 2>&1; echo "prof b1024 rc=$?"
tools/gpu_profile.sh b64 --blocks 64 --no-pc > gpurun_out/prof_b64.log 2>&1; echo "prof b64 rc=$?"
tools/gpu_profile.sh randomize_16_20 --randomize 16 20 > gpurun_out/prof_rand.log 2>&1; echo "prof rand rc=$?"
VP_LIBGPU=tools/_build/stamps/libvpgpu.so python tools/leaf_in_step.py 1024 > gpurun_out/leaf_in_step.txt 2>&1; echo "leaf rc=$?"; grep "^==" gpurun_out/leaf_in_step.txt | head -8
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/bench_default.json
