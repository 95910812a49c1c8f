#!/bin/bash
# Development sweep: fuse-init threshold (VP_FUSE_MIN_LOG) per circuit size, same call.  tools/sweep_fuse.sh "BLOCKS:v1,v2,..." ...
mkdir -p gpurun_out
for spec in "$@"; do
  b=${spec%%:*}; vs=${spec#*:}
  for v in ${vs//,/ }; do
    VP_FUSE_MIN_LOG=$v python bench.py --blocks $b --no-x1024-leg --no-cpu-baseline --steps 12 > gpurun_out/sw_${b}_$v.json 2> /dev/null || exit 1
    python - $b $v <<'PY'
import json,sys; d=json.loads(open("gpurun_out/sw_%s_%s.json"%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1]); print("x%s fuse_min_log %s device %.4f ms wall %.4f exact %s"%(sys.argv[1], sys.argv[2], d["prover_sec_device"]*1e3, d["ms_per_step"], d["bit_exact_vs_reference_golden"]), flush=True)
PY
  done
done
