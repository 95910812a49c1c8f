set -o pipefail
python -m pytest tests -x -q -m gpu > gpurun_out/t6.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/t6.log
tools/seam_x1024.sh 1024 > gpurun_out/seam1024.log 2>&1; tail -14 gpurun_out/seam1024.log
tools/seam_x1024.sh 64 > gpurun_out/seam64.log 2>&1; tail -12 gpurun_out/seam64.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/bench_default.json
python -c "import __graft_entry__ as g; g.smoke()"
