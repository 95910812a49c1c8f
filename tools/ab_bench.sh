#!/bin/bash
# Development A/B of two builds of libvpgpu.so inside ONE gpurun call (clocks differ from box to box by several per cent, so only same-call
# numbers are compared):  tools/ab_bench.sh TAG path/to/variantA/libvpgpu.so [bench flags...]   ("" = the product library)
# Alternates A, product, A, product; one JSON line each into gpurun_out/TAG_{a,b}{1,2}.json
T=$1; A=$2; shift 2
mkdir -p gpurun_out
for i in 1 2; do
  VP_LIBGPU=$A python bench.py "$@" > gpurun_out/${T}_a$i.json || exit 1
  python bench.py "$@" > gpurun_out/${T}_b$i.json || exit 1
done
python - "$T" <<'PY'
import json, sys
t = sys.argv[1]
for k in ("a1", "b1", "a2", "b2"):
    d = json.loads(open("gpurun_out/%s_%s.json" % (t, k)).read().strip().splitlines()[-1])
    pc = d.get("polynomial_commitment") or {}
    extra = (" pc_commit_side %.2f ms roots_exact %s" % (pc["pc_commit_side_device_ms"], pc.get("fri_roots_bit_exact"))) if "pc_commit_side_device_ms" in pc else ""
    print(k, "ms_per_step %.4f device %.4f single-stream %.4f exact %s%s" % (d["ms_per_step"], d["prover_sec_device"] * 1e3, d["roofline"]["single_stream_proof_ms"], d.get("bit_exact_vs_reference_golden", d.get("bit_exact_vs_oracle_fixture")), extra))
PY
