set -o pipefail
python -m pytest tests -x -q -m gpu > gpurun_out/t_final.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/t_final.log
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/bench_default.json
