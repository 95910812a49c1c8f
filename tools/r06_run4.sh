set -o pipefail
python -m pytest tests -x -q -m gpu > gpurun_out/t4.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/t4.log
tools/seam_x1024.sh 1024 > gpurun_out/seam1024.log 2>&1; tail -17 gpurun_out/seam1024.log
tools/seam_x1024.sh 64 > gpurun_out/seam64.log 2>&1; tail -14 gpurun_out/seam64.log
T0=$(date +%s.%N)
VP_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 2 --steps 20 --warmup 5 --detail-file gpurun_out/bench_detail_n2_rehearsal.json > gpurun_out/rehearsal2.json 2> gpurun_out/rehearsal2.err; echo "rehearsal rc=$? wall $(echo "$(date +%s.%N) - $T0" | bc) s"; cut -c1-400 gpurun_out/rehearsal2.json
