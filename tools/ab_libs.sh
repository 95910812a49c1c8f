#!/bin/bash
# tools/ab_libs.sh lib1 lib2 ... : the headline step with each library ("-" = product), alternating, twice
for rep in 1 2; do for L in "$@"; do
  if [ "$L" = "-" ]; then unset VP_LIBGPU; else export VP_LIBGPU=$(pwd)/$L; fi
  python bench.py --no-cpu-baseline --no-x64-leg --no-randomize-leg --no-pass-modes --steps 20 --detail-file gpurun_out/_ab_detail.json > gpurun_out/_ab.json 2>/dev/null || { echo "$L failed"; continue; }
  python - "$L" <<'PY'
import json,sys; d=json.loads(open("gpurun_out/_ab.json").read().strip().splitlines()[-1]); p=d["prover_sec"]
print("%-36s step %.2f ms  gkr %.2f  private %.2f  public %.2f  fri %.2f  bit_exact %s" % (sys.argv[1], d["ms_per_step"], 1e3*p["gkr"], 1e3*p["commit_private"], 1e3*p["commit_public"], 1e3*p["fri_commit"], d["bit_exact"]), flush=True)
PY
done; done
