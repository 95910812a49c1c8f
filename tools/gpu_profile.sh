#!/bin/bash
# Profiles of ONE bench configuration on the GPU box (run through gpurun from the repo root), the batched passes ALONE in the profiled process
# (bench.py --batched-only: no profiled replay, no interactive run, no verification, no nested legs — per-kernel counters are the headline's launches):
#   tools/gpu_profile.sh TAG [bench args]      TAG names the output: gpurun_out/prof_TAG/{kernel_stats.csv, kernel_stats_serial.csv, pmc_summary.json, bench_*.json}
#     tools/gpu_profile.sh b1024                                  the headline (BASELINE configs[2]: x1024 + commitment)
#     tools/gpu_profile.sh b64 --blocks 64 --no-pc                BASELINE configs[1]
#     tools/gpu_profile.sh randomize_16_20 --randomize 16 20      BASELINE configs[4]
# Kernel-trace stats and each PMC counter set are separate rocprofv3 runs (MI355X_MICROARCH.md, HBM / rocprofv3 section); the program follows `--` directly.
# The plan tuner runs once, unprofiled, and leaves its choice in a VP_PLAN_CACHE file: every profiled process replays THAT plan and nothing of the tuner.
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=$1
O=$R/gpurun_out/prof_$T
shift
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
export VP_PLAN_CACHE=$O/plan_cache.txt
B="python3 $R/bench.py --steps 3 --warmup 2 --batched-only --no-cpu-baseline $*"
$B > "$O/bench_untraced.json" 2> "$O/untraced.err" || exit 1
rocprofv3 --kernel-trace --stats -d "$O/stats" -o stats -- $B > "$O/bench_stats.json" 2> "$O/stats.err" || exit 1
VP_GKR_SERIAL=1 rocprofv3 --kernel-trace --stats -d "$O/stats_serial" -o stats -- $B > "$O/bench_stats_serial.json" 2> "$O/stats_serial.err" || exit 1
rocprofv3 --pmc FETCH_SIZE -d "$O/fetch" -o pmc -- $B > /dev/null 2> "$O/fetch.err" || exit 1
rocprofv3 --pmc WRITE_SIZE -d "$O/write" -o pmc -- $B > /dev/null 2> "$O/write.err" || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS -d "$O/sq" -o pmc -- $B > /dev/null 2> "$O/sq.err"
# rocprofv3 writes rocpd SQLite databases by default; the summaries are made from them
python3 "$R/tools/pmc_summary.py" "$O/pmc_summary.json" "$O/fetch" "$O/write" "$O/sq" --command "bench.py --steps 3 --warmup 2 --batched-only $*" > "$O/pmc_summary.txt"
python3 "$R/tools/pmc_summary.py" --stats "$O/stats/stats_results.db" "$O/kernel_stats.csv"
python3 "$R/tools/pmc_summary.py" --stats "$O/stats_serial/stats_results.db" "$O/kernel_stats_serial.csv"
python3 "$R/tools/pmc_summary.py" --add-floor "$O/pmc_summary.json" "$O/kernel_stats_serial.csv"
# only the summaries travel back (gpurun merges at most 64 MiB): the raw databases stay on the box
rm -rf "$O/stats" "$O/stats_serial" "$O/fetch" "$O/write" "$O/sq"
ls -la "$O"
