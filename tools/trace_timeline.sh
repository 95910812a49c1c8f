#!/bin/bash
# Development tool: kernel trace of the bench at BLOCKS and the timelines of its last four proofs (tools/timeline.py):  tools/trace_timeline.sh BLOCKS
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/trace_$1; B=$1; shift
mkdir -p "$O"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$O/kt" -o kt -- python3 $R/bench.py --blocks $B --steps 3 --warmup 2 --no-cpu-baseline --no-x1024-leg "$@" > "$O/bench.json" 2> "$O/err.log" || { tail -5 "$O/err.log"; exit 1; }
D=$(find "$O/kt" -name "*_results.db" | head -1)
for w in 1 2 3 4; do python3 $R/tools/timeline.py $D $w > "$O/timeline_$w.txt"; done
find "$O" -name "*.db" -size +16M -delete
