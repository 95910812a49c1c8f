#!/bin/bash
# Effective shader clock and instruction counts of the two leaf-hash kernels in isolation (tools/_build/micro_leaf, tools/build_micro_keccak.sh first):
# GRBM_GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md, DVFS note), SQ_INSTS_VALU, SQ_WAVE_CYCLES — separate rocprofv3 --pmc passes.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/clock_leaf; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE -d "$O/clk" -o pmc -- $R/tools/_build/micro_leaf 17 > "$O/run_clk.txt" 2> "$O/err_clk.txt" || { tail -3 "$O/err_clk.txt"; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d "$O/sq" -o pmc -- $R/tools/_build/micro_leaf 17 > "$O/run_sq.txt" 2> "$O/err_sq.txt" || { tail -3 "$O/err_sq.txt"; exit 1; }
python3 - "$O" <<'PY'
import glob, sqlite3, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("clk", "sq"):
    for f in glob.glob(O + "/" + sub + "/**/*_results.db", recursive=True):
        db = sqlite3.connect(f)
        q = "select kernel_name, counter_name, dispatch_id, sum(value), max(duration) from counters_collection group by kernel_name, counter_name, dispatch_id"
        for name, counter, _, v, dur in db.execute(q):
            agg[name.split('(')[0]][counter].append((v, dur))
for name, cs in agg.items():
    if "leaf_hash" not in name: continue
    line = "%-28s" % name.replace("vp::", "")
    if "GRBM_GUI_ACTIVE" in cs:
        v = cs["GRBM_GUI_ACTIVE"][1:]          # first launch: warm-up
        line += " %d launches  %.3f ms  effective clock %.2f GHz" % (len(v), sum(d for _, d in v) / len(v) / 1e6, sum(x for x, _ in v) / 8 / sum(d for _, d in v))
    for c in ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY"):
        if c in cs:
            v = cs[c][1:]
            line += "  %s %.3e" % (c, sum(x for x, _ in v) / len(v))
    print(line)
PY
rm -rf "$O/clk" "$O/sq"
