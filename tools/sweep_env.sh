#!/bin/bash
# Development A/B of one environment switch in one call:  tools/sweep_env.sh VAR "v1 v2 ..." BLOCKS [bench args]
# (for the plan-layout switches — VP_FUSE_COMBINE, VP_FOLD_BRANCHES, VP_PLAN_ALIGN, VP_FUSE_MIN_LOG — export VP_PLAN_AUTOTUNE=0 first)
V=$1; VALS=$2; B=$3; shift 3
mkdir -p gpurun_out
for v in $VALS; do
  env $V=$v python bench.py --blocks $B --no-x1024-leg --no-cpu-baseline --steps 12 "$@" > gpurun_out/se_${V}_${B}_$v.json 2> /dev/null || exit 1
  python - $V $B $v <<'PY'
import json,sys; d=json.loads(open("gpurun_out/se_%s_%s_%s.json"%tuple(sys.argv[1:4])).read().strip().splitlines()[-1])
ks={k["kernel"]: k["total_us"] for k in d["kernels"]}
print("%s=%s x%s device %.4f ms wall %.4f single-stream %.4f exact %s light %.1f gen %.1f upload %.3f"%(sys.argv[1], sys.argv[3], sys.argv[2], d["prover_sec_device"]*1e3, d["ms_per_step"], d["roofline"]["single_stream_proof_ms"], d["bit_exact_vs_reference_golden"], ks.get("k_light_multi",0), ks.get("k_sumfold3b_gen_multi",0), d["circuit_upload_sec"]), flush=True)
PY
done
