set -o pipefail
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hundreds_of_deferred or vp_warm or fft_vs_oracle or fft_known or commit_private_root or commit_private_two_real or split_transforms or fri_commit_phase or protocol_pass_matches or reference_binary or x1024_full" > gpurun_out/t3.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t3.log
tools/seam_x1024.sh 1024 > gpurun_out/seam1024.log 2>&1; tail -17 gpurun_out/seam1024.log
tools/ab_libs.sh - tools/_build/ntt_ps0/libvpgpu.so > gpurun_out/ab_ntt_ps.txt 2>&1; cat gpurun_out/ab_ntt_ps.txt
python tools/pc_ab.py 1024 - tools/_build/ntt_ps0/libvpgpu.so > gpurun_out/ab_ntt_ps_kernels.txt 2>&1; tail -20 gpurun_out/ab_ntt_ps_kernels.txt
