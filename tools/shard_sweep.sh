#!/bin/bash
# Development: per-rank compute of one proof over W ranks, chain shard vs chain shard + index split, same call.  tools/shard_sweep.sh BLOCKS W [MIN_LOGS...]
B=$1; W=$2; shift 2
mkdir -p gpurun_out
for m in 0 "$@"; do
  python bench.py --blocks $B --no-x64-leg --no-randomize-leg --no-cpu-baseline --steps 6 --shard-sim $W --shard-split $m --detail-file gpurun_out/ss_${B}_${W}_$m.detail.json > gpurun_out/ss_${B}_${W}_$m.json 2> /dev/null || exit 1
  python - $B $W $m <<'PY'
import json,sys; d=json.load(open("gpurun_out/ss_%s_%s_%s.detail.json"%tuple(sys.argv[1:4]))); s=d["sharded_proof_simulation"]
print("x%s W=%s split_min_log %s: unsharded %.3f ms, max over ranks %.3f ms (%.2fx), per rank %s, chains split %s, finish %.3f ms, assembled ok %s" % (sys.argv[1], sys.argv[2], sys.argv[3], d["prover_sec_device"]*1e3, s["max_device_ms"], d["prover_sec_device"]*1e3/s["max_device_ms"], [round(x["device_ms"],3) for x in s["per_rank"]], s["chains_split_by_index"], s["host_finish_ms"] or 0, s["assembled_equals_unsharded"]), flush=True)
PY
done
