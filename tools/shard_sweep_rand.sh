#!/bin/bash
# as tools/shard_sweep.sh for a `randomize(LAYERS, LOG)` circuit (few, large chains).  tools/shard_sweep_rand.sh LAYERS LOG W [MIN_LOGS...]
L=$1; G=$2; W=$3; shift 3
mkdir -p gpurun_out
for m in 0 "$@"; do
  python bench.py --randomize $L $G --no-x64-leg --no-randomize-leg --steps 6 --shard-sim $W --shard-split $m --detail-file gpurun_out/ssr_${L}_${G}_${W}_$m.detail.json > gpurun_out/ssr_${L}_${G}_${W}_$m.json 2> /dev/null || exit 1
  python - $L $G $W $m <<'PY'
import json,sys; d=json.load(open("gpurun_out/ssr_%s_%s_%s_%s.detail.json"%tuple(sys.argv[1:5]))); s=d["sharded_proof_simulation"]
print("randomize(%s,%s) W=%s split_min_log %s: unsharded %.3f ms, max over ranks %.3f ms (%.2fx), per rank %s, chains split %s, assembled ok %s" % (sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], d["prover_sec_device"]*1e3, s["max_device_ms"], d["prover_sec_device"]*1e3/s["max_device_ms"], [round(x["device_ms"],3) for x in s["per_rank"]], s["chains_split_by_index"], s["assembled_equals_unsharded"]), flush=True)
PY
done
