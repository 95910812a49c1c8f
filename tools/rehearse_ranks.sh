#!/bin/bash
# Rehearsal of bench.py's multi-rank modes on a ONE-GPU box (both ranks on the same card; timings are meaningless, the
# code path — rendezvous, barrier, max-over-ranks, sharded proof + transcript all-reduce, JSON line — is what is exercised).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 5 --warmup 2 --blocks 16 --no-pc > gpurun_out/rehearse_weak.json 2> gpurun_out/rehearse_weak.err || { tail -20 gpurun_out/rehearse_weak.err; exit 1; }
VP_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 2 --steps 5 --warmup 2 --blocks 16 --no-pc --shard-chains > gpurun_out/rehearse_shard.json 2> gpurun_out/rehearse_shard.err || { tail -20 gpurun_out/rehearse_shard.err; exit 1; }
python3 - <<'PY'
import json
for f in ("rehearse_weak","rehearse_shard"):
    d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["n_gpus"], d["ranks"], d["scaling"], d["config"]["proofs_per_step"], "bit-exact", d["bit_exact"], "verifier", d.get("host_verifier_accepts"), d.get("sharded_proof"))
PY
