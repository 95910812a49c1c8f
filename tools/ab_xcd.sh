#!/bin/bash
# A/B of the XCD-aware block map (VP_XCD_MAP=0|1): bench line + per-kernel durations of the same run
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/ab_xcd; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
for B in 64 1024; do for X in 0 1; do
  VP_XCD_MAP=$X rocprofv3 --kernel-trace --stats -d "$O/b${B}_x$X" -o s -- python3 $R/bench.py --blocks $B --no-pc --steps 6 --warmup 2 --no-cpu-baseline > "$O/bench_b${B}_x$X.json" 2> /dev/null || exit 1
  python3 $R/tools/pmc_summary.py --stats "$O/b${B}_x$X/s_results.db" "$O/stats_b${B}_x$X.csv"; rm -rf "$O/b${B}_x$X"
  echo "== blocks $B xcd_map $X"; grep -E "k_light_multi|k_chunks_multi|k_sumfold3b" "$O/stats_b${B}_x$X.csv" | cut -c1-60,100-
done; done
