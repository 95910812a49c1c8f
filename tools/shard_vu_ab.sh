#!/bin/bash
# Index-split proof, per-rank compute with V_u of the split phase-2 chains (a) added up whole on every rank (round 4) and (b) from the ranks' partial
# inner products + one exchange (round 5), same box, alternating.  tools/shard_vu_ab.sh BLOCKS W [MIN_LOG]
B=$1; W=$2; M=${3:-11}
mkdir -p gpurun_out
for rep in 1 2; do for mode in whole exchange; do
  flag=""; [ $mode = whole ] && flag="--no-vu-exchange"
  python bench.py --blocks $B --no-x64-leg --no-randomize-leg --no-cpu-baseline --steps 20 --shard-sim $W --shard-split $M $flag --detail-file gpurun_out/svu_${B}_${W}_$mode.detail.json > gpurun_out/svu_${B}_${W}_$mode.json 2> gpurun_out/svu_err.txt || { tail -5 gpurun_out/svu_err.txt; exit 1; }
  python - $B $W $mode <<'PY'
import json,sys; d=json.load(open("gpurun_out/svu_%s_%s_%s.detail.json"%tuple(sys.argv[1:4]))); s=d["sharded_proof_simulation"]
print("x%s W=%s V_u %-8s: unsharded %.3f ms, max over ranks %.3f ms (%.2fx), per rank %s, chains split %s, V_u exchanged %s, assembled ok %s" % (sys.argv[1], sys.argv[2], sys.argv[3], d["prover_sec_device"]*1e3, s["max_device_ms"], d["prover_sec_device"]*1e3/s["max_device_ms"], [round(x["device_ms"],3) for x in s["per_rank"]], s["chains_split_by_index"], s["v_u_by_partial_inner_products"], s["assembled_equals_unsharded"]), flush=True)
PY
done; done
