#!/bin/bash
# Effective shader clock per kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration (MI355X_MICROARCH.md, DVFS note).
#   tools/clock_probe.sh TAG [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/clock_$1; shift; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE -d "$O/p" -o pmc -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline $* > /dev/null 2> "$O/err.txt" || { tail -3 "$O/err.txt"; exit 1; }
python3 - "$O/p" <<'PY'
import glob, sqlite3, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True):
    db = sqlite3.connect(f)
    q = "select kernel_name, dispatch_id, sum(value), max(duration) from counters_collection where counter_name='GRBM_GUI_ACTIVE' group by kernel_name, dispatch_id"
    for name, _, v, dur in db.execute(q):
        a = agg[name.split('(')[0]]; a[0] += 1; a[1] += v; a[2] += dur
for name, (n, v, dur) in sorted(agg.items(), key=lambda kv: -kv[1][2])[:14]:
    if dur > 0: print("%-44s launches %4d  total %9.1f us  effective clock %.2f GHz" % (name[:44], n, dur / 1e3, v / 8 / dur))
PY
