R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/prof_fftsync; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
VPH_FFT_GKR_SYNC=1 rocprofv3 --kernel-trace --stats -d $O/stats -o stats -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-x64-leg > $O/line.json 2> $O/err.txt || exit 1
python3 $R/tools/pmc_summary.py --stats $O/stats/stats_results.db $O/kernel_stats.csv
rm -rf $O/stats
grep "k_fg\|k_emit\"\|k_seg<\|k_sumfold3b<" $O/kernel_stats.csv
