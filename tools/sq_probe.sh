#!/bin/bash
# Issue-side counters of the plan kernels (one rocprofv3 --pmc pass per counter group):  tools/sq_probe.sh TAG [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/sq_$1; shift; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline $*"
i=0
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
         "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $G -d "$O/g$i" -o pmc -- $B > /dev/null 2> "$O/g$i.err" || { echo "group $i failed: $(tail -2 $O/g$i.err)"; continue; }
done
python3 - "$O" <<'PY'
import glob, sqlite3, sys, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True):
    db = sqlite3.connect(f)
    for name, counter, _, v in db.execute("select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection group by kernel_name, counter_name, dispatch_id"):
        a = agg[name.split('(')[0]][counter]; a[0] += 1; a[1] += v
out = {k: {c: v[1] / v[0] for c, v in cs.items()} for k, cs in agg.items() if any(x in k for x in ("sumfold3b", "k_light_multi", "k_chunks_multi", "k_seg_multi", "k_emit_multi"))}
json.dump(out, open(sys.argv[1] + "/sq_summary.json", "w"), indent=1)
for k, cs in out.items():
    print(k); print("   ", {c: round(v / 1e6, 2) for c, v in sorted(cs.items())})
PY
rm -rf "$O"/g[0-9]*          # the databases are large (gpurun merges at most 64 MiB back); the summary is what is kept
