#!/bin/bash
# kernel statistics of the headline bench with the commitment at a given size:  tools/prof_pc.sh BLOCKS TAG
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/prof_pc_$2; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$O/st" -o s -- python3 $R/bench.py --blocks $1 --steps 4 --warmup 2 --no-cpu-baseline --no-x64-leg > "$O/bench.json" 2> "$O/err.txt" || { tail -5 "$O/err.txt"; exit 1; }
python3 $R/tools/pmc_summary.py --stats "$O/st/s_results.db" "$O/kernel_stats.csv"; rm -rf "$O/st"
head -30 "$O/kernel_stats.csv" | cut -c1-70,150-
