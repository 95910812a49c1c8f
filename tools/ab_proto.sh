#!/bin/bash
# A/B of run-time switches on the headline (x1024 + commitment): one bench run per setting, same box, same call.
#   [AB_BLOCKS=64] tools/ab_proto.sh "VAR=val ..." "VAR=val ..." ...      ("-" = defaults)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
i=0
for E in "$@"; do
  i=$((i+1)); [ "$E" = "-" ] && E=""
  env $E python3 bench.py ${AB_BLOCKS:+--blocks $AB_BLOCKS} --steps 10 --warmup 3 --no-cpu-baseline --no-x64-leg --detail-file gpurun_out/ab_proto_$i.json > gpurun_out/ab_proto_$i.line 2> gpurun_out/ab_proto_$i.err || { tail -5 gpurun_out/ab_proto_$i.err; exit 1; }
  python3 - "$i" "$E" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_proto_%s.json" % sys.argv[1]))
ps = d["prover_sec"]
print("[%s] %-40s step %.2f ms | priv %.2f gkr %.2f pub %.2f fft %.2f fri %.2f | exact %s" % (sys.argv[1], sys.argv[2] or "defaults", 1e3 * ps["step_wall"], 1e3 * ps["commit_private"],
      1e3 * ps["gkr"], 1e3 * ps["commit_public"], 1e3 * ps["fft_gkr"], 1e3 * ps["fri_commit"], all(v is not False for v in d["bit_exact"].values())))
ip = d["interactive_path"]
print("     interactive %.2f ms (init %.2f, rounds %.2f) equal %s | " % (1e3 * ip["prover_sec"], 1e3 * ip["init_calls_sec"], 1e3 * ip["round_calls_sec"], ip["transcript_equals_batched"])
      + "  ".join("%s: %d x %.1f us" % (c["served_by"][:22], c["rounds"], c["avg_us"]) for c in ip["per_round"]["by_path"]))
print("     " + "  ".join("%s %.2f" % (k["kernel"], k["total_us"] / 1e3) for k in d["kernels"][:9]))
PY
done
