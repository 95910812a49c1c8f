#!/bin/bash
# Profiles of the headline bench on the GPU box (run through gpurun from the repo root):
#   tools/gpu_profile.sh TAG [bench args; default = the headline (x1024 + commitment); e.g. --blocks 64 --no-pc]   -> gpurun_out/prof_TAG/{stats,stats_serial,fetch,write,sq}/..., pmc_summary.json
# Kernel-trace stats and each PMC counter set are separate rocprofv3 runs (MI355X_MICROARCH.md, HBM / rocprofv3 section).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$1
shift
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-x64-leg --no-randomize-leg --no-pass-modes $*"
rocprofv3 --kernel-trace --stats -d "$O/stats" -o stats -- $B > "$O/bench_stats.json" 2> "$O/stats.err" || exit 1
VP_GKR_SERIAL=1 rocprofv3 --kernel-trace --stats -d "$O/stats_serial" -o stats -- $B > "$O/bench_stats_serial.json" 2> "$O/stats_serial.err" || exit 1
rocprofv3 --pmc FETCH_SIZE -d "$O/fetch" -o pmc -- $B > /dev/null 2> "$O/fetch.err" || exit 1
rocprofv3 --pmc WRITE_SIZE -d "$O/write" -o pmc -- $B > /dev/null 2> "$O/write.err" || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS -d "$O/sq" -o pmc -- $B > /dev/null 2> "$O/sq.err"
# rocprofv3 writes rocpd SQLite databases by default; the summaries are made from them
python3 "$R/tools/pmc_summary.py" "$O/pmc_summary.json" "$O/fetch" "$O/write" "$O/sq" > "$O/pmc_summary.txt"
python3 "$R/tools/pmc_summary.py" --stats "$O/stats/stats_results.db" "$O/kernel_stats.csv"
python3 "$R/tools/pmc_summary.py" --stats "$O/stats_serial/stats_results.db" "$O/kernel_stats_serial.csv"
# only the summaries travel back (gpurun merges at most 64 MiB): the raw databases stay on the box
rm -rf "$O/stats" "$O/stats_serial" "$O/fetch" "$O/write" "$O/sq"
ls -la "$O"
