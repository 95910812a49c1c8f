#!/usr/bin/env python3
"""bench.py — headline benchmark: GKR prover seconds and field-ops/s on the SHA-256 64-block circuit
(BASELINE.json configs[1]: "SHA-256 64-block circuit, 1xMI355X, sumcheck rounds on GPU, Virgo PC off").

A step = one complete GKR proof (Vres + three sumchecks per layer, 691 rounds) by the device prover, with
the circuit, the witness and the verifier tape already resident in HBM.  Each rank proves its own instance
(rank r draws its witness after srandom(1 + r); rank 0's instance is the golden one), so N GPUs produce N
independent proofs per step with no data-path collective ("scaling": "weak").  Rank 0's transcript is
compared byte for byte with the real reference's golden transcript on every run.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--blocks B] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import gzip
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
PMC_SUMMARY = "r01_l_pmc_summary_b64.json"   # made by tools/pmc_summary.py from three rocprofv3 --pmc passes of this script


def dist_setup(n_gpus, rccl=False):
    """One process per GPU.  Default mode: the data path has no collective (independent proofs per rank); the control
    plane (barriers, max-over-ranks of the elapsed time) runs over torch.distributed/gloo.  rccl=True (--shard-chains):
    device tensors additionally go over RCCL (backend "nccl" = RCCL over xGMI) for the transcript all-reduce."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if rccl and os.environ.get("VP_BENCH_BACKEND", "") != "gloo":
            import torch
            torch.cuda.set_device(local)
            dist.init_process_group(backend="cpu:gloo,cuda:nccl", rank=rank, world_size=world)
        else:       # VP_BENCH_BACKEND=gloo: rehearsal of the sharded mode on a box with fewer GPUs than ranks (RCCL needs one GPU per rank)
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    return world, rank, local


def device_of(local):
    """GPU of this rank: LOCAL_RANK, folded onto the visible devices when a rehearsal runs more ranks than the box has GPUs."""
    import torch
    n = torch.cuda.device_count()
    return local % n if n else local


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def aggregate(world, elapsed, units):
    """(max elapsed over ranks, total units over ranks)."""
    if world == 1:
        return elapsed, units
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed], dtype=torch.float64)
    u = torch.tensor([units], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def gpu_sync(device):
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize(device)


def unpack_pws(tmpdir):
    p = os.path.join(tmpdir, "SHA256_64.pws")
    with gzip.open(os.path.join(ROOT, "tests", "golden", "SHA256_64.pws.gz"), "rb") as f, open(p, "wb") as g:
        g.write(f.read())
    return p


def cpu_baseline(pws, blocks, ref_ops):
    """The reference's CPU prover on this box's host cores (single thread: the reference has no threading).
    Prefers the REAL reference binary (oracle/_ref/ref_run, built from /root/reference in the build
    container); falls back to the repo's restatement (oracle/, bit-identical by tests/test_oracle_golden.py)."""
    ref_run = os.path.join(ROOT, "oracle", "_ref", "ref_run")
    sample = "one full GKR proof of the same %d-block circuit (PC off), single thread" % blocks
    if os.path.exists(ref_run):
        try:
            out = subprocess.run([ref_run, "--pws", pws, "--blocks", str(blocks), "--pc", "0"], stdout=subprocess.PIPE,
                                 stderr=subprocess.DEVNULL, text=True, timeout=600)
            m = re.search(r"Prove Time ([0-9.]+)", out.stdout)
            c = re.search(r"mult counter (-?\d+), add counter (-?\d+)", out.stdout)
            if out.returncode == 0 and m and c:
                sec = float(m.group(1))
                ops = int(c.group(1)) + int(c.group(2))
                return {"value": ops / sec, "unit": "field-ops/s", "cores": 1, "kind": "reference", "sample": sample,
                        "prover_sec": sec, "field_ops": ops}
        except Exception:
            pass
    import oracle_binding as ob
    c = ob.Circuit.from_pws(pws, blocks, seed=1)
    _, st = c.prove_gkr()
    c.close()
    ops = st["mult_count"] + st["add_count"]
    return {"value": ops / st["prove_sec"], "unit": "field-ops/s", "cores": 1, "kind": "port", "sample": sample,
            "prover_sec": st["prove_sec"], "field_ops": ops}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--randomize", type=int, nargs=2, metavar=("LAYERS", "LOG_SIZE"), default=None,
                    help="BASELINE configs[4] flavour: layeredCircuit::randomize(LAYERS, LOG_SIZE) instead of the SHA-256 circuit (no CPU baseline, no golden)")
    ap.add_argument("--shard-chains", action="store_true",
                    help="strong scaling: ONE proof per step; its independent sumcheck chains are dealt out to the ranks (vp_set_shard) and the "
                         "transcript is assembled by one RCCL all-reduce per proof (launch under torch.distributed.run)")
    ap.add_argument("--shard-sim", type=int, default=0, metavar="W",
                    help="single GPU: run the W shards of a chain-sharded proof one after the other and report each shard's device time "
                         "(the per-rank compute of a W-GPU run; outside the timed region)")
    ap.add_argument("--with-pc", action="store_true", help="also time the Virgo commitment (commit_private + commit_public + FRI commit phase)")
    a = ap.parse_args()

    # The native libraries are loaded BEFORE torch so that every rank count uses the same HIP runtime load order
    # (libvpgpu.so first; torch then reuses the already loaded libamdhip64).  Only local rank 0 may (re)build.
    import vp_loader
    vp = vp_loader.load()
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        vp.build()
        vp.lib_host()
    world, rank, local = dist_setup(a.gpus, rccl=a.shard_chains)
    local = device_of(local)
    shard = a.shard_chains and world > 1
    seed = 1 if shard else 1 + rank          # a sharded proof: every rank holds the same instance
    barrier(world)
    vp.lib_host()
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
    gname = "sha256_x%d" % a.blocks if not a.randomize else "randomize_%d_%d" % tuple(a.randomize)
    if a.randomize:
        a.no_cpu_baseline = True
    with tempfile.TemporaryDirectory() as tmp:
        pws = unpack_pws(tmp)
        if a.randomize:
            circ = vp.Circuit.randomize(a.randomize[0], a.randomize[1], seed=seed)
        else:
            circ = vp.Circuit.from_pws(pws, a.blocks, seed=seed)
        t_up = time.perf_counter()
        sess = vp.Session(circ, device=local)            # raises without the HIP library / GPU
        upload_sec = time.perf_counter() - t_up
        sess.draw_tape()
        if shard:
            sess.set_shard(rank, world)
        for _ in range(a.warmup):
            tr, _ = sess.prove_gkr()
            if shard:
                vp.allreduce_transcript(tr, local)
        gpu_sync(local)
        barrier(world)
        t0 = time.perf_counter()
        dev_ms = 0.0
        res = None
        for k in range(a.steps):
            tr, res = sess.prove_gkr()
            if shard:       # the one data-path collective: u64 sum of the ranks' disjoint transcript slices over RCCL
                tr = vp.allreduce_transcript(tr, local)
            dev_ms += res["gkr_device_ms"]
        gpu_sync(local)
        barrier(world)
        elapsed = time.perf_counter() - t0
        elapsed, proofs = aggregate(world, elapsed, float(a.steps))
        shard_info = None
        if shard:
            proofs = float(a.steps)                    # all ranks worked on the same proof
            import torch, torch.distributed as dist
            dm = torch.tensor([dev_ms / a.steps if r == rank else 0.0 for r in range(world)], dtype=torch.float64)
            dist.all_reduce(dm)
            owner, cost = sess.shard_chains()
            shard_info = {"device_ms_per_rank": [float(x) for x in dm], "chains": int((cost > 0).sum()),
                          "chains_per_rank": [int(((owner == r) & (cost > 0)).sum()) for r in range(world)],
                          "collective": "one all-reduce (sum, int64) of the %d-byte transcript per proof, backend %s" % (len(tr), dist.get_backend())}
            sess.set_shard(0, 1)                       # the roofline / verifier legs below run the whole proof on every rank
            tr_full, _ = sess.prove_gkr()
            assert tr_full == tr, "assembled sharded transcript differs from the unsharded proof"
        # Roofline pass (outside the timed region, same process, same resident state): the proof is replayed on ONE
        # stream with HIP events around every launch of the dominant kernel.  In the timed steps the independent
        # sumchecks overlap on separate streams, which makes per-kernel event times meaningless there.
        sess.set_profiling(1)
        tr_p, res_p = sess.prove_gkr()
        sess.set_profiling(0)
        assert tr_p == tr
        for key in ("fold_ms", "fold_launches", "fold_bytes"):
            res[key] = res_p[key]
        res["serial_device_ms"] = res_p["gkr_device_ms"]

        shard_sim = None
        if a.shard_sim > 1 and world == 1:
            # per-rank compute of a chain-sharded proof on W GPUs, measured shard by shard on this one GPU (no collective here)
            per = []
            parts = []
            for r in range(a.shard_sim):
                sess.set_shard(r, a.shard_sim)
                for _ in range(2):
                    sess.prove_gkr()
                ms = []
                for _ in range(max(3, a.steps // 2)):
                    t_s = time.perf_counter()
                    tr_s, res_s = sess.prove_gkr()
                    ms.append((res_s["gkr_device_ms"], 1e3 * (time.perf_counter() - t_s)))
                parts.append(tr_s)
                per.append({"rank": r, "device_ms": sum(m[0] for m in ms) / len(ms), "wall_ms": sum(m[1] for m in ms) / len(ms)})
            owner, cost = sess.shard_chains()
            sess.set_shard(0, 1)
            shard_sim = {"world": a.shard_sim, "per_rank": per, "max_device_ms": max(x["device_ms"] for x in per),
                         "max_wall_ms": max(x["wall_ms"] for x in per),
                         "cost_share_per_rank": [float(cost[owner == r].sum() / cost.sum()) for r in range(a.shard_sim)],
                         "assembled_equals_unsharded": vp.sum_transcripts(parts) == tr,
                         "note": "each shard run alone on this GPU; a W-GPU run adds one all-reduce of the transcript per proof"}

        interactive = None
        if rank == 0:
            # the drop-in path of the reference's own call pattern (one vp_round per verifier message), outside the timed region
            t_i = time.perf_counter()
            tr_i, res_i, ok_i = sess.prove_interactive()
            interactive = {"prover_sec": res_i["prove_sec"], "wall_sec_with_host_verifier": time.perf_counter() - t_i,
                           "transcript_equals_batched": tr_i == tr, "verified": ok_i,
                           "note": "reference definition of Prove Time (sum of prover-method spans), one launch + sync per round"}
            sess.draw_tape()

        pc = None
        if a.with_pc and rank == 0:
            # BASELINE.json configs[2] flavour: the commitment's commit side on the GPU (not part of `value`)
            import numpy as np
            t1 = time.perf_counter()
            full, okf = sess.prove_full(batched=True)
            t_full = time.perf_counter() - t1
            gname_ = gname
            pc = {"full_proof_wall_sec": t_full, "full_transcript_verified": okf}
            if gname_ in golden and "fri" in golden[gname_]:      # recorded runs of the real reference (full transcript + FRI steps)
                from conftest import GOLDEN
                gg = golden[gname_]
                pc["full_transcript_bit_exact"] = (full == open(os.path.join(GOLDEN, gg["transcript"]), "rb").read())
                fri = open(os.path.join(GOLDEN, gg["fri"]), "rb").read()
                st = gg["fri_steps"]
                rec = np.frombuffer(fri[:48 * st], dtype=np.uint64).reshape(st, 6)
                sess.fri_commit(np.ascontiguousarray(rec[:, :2]))          # first call allocates the FRI buffers
                sess.prove_full(batched=True)
                t2 = time.perf_counter()
                roots, fin = sess.fri_commit(np.ascontiguousarray(rec[:, :2]))
                pc["fri_commit_wall_sec"] = time.perf_counter() - t2
                pc["fri_commit_device_ms"] = sess.commit_device_ms()
                pc["fri_roots_bit_exact"] = (roots == b"".join(rec[i, 2:].tobytes() for i in range(st)))
            else:
                # no recorded reference run at this size: fold with fresh challenges (any challenges exercise the same work)
                st = circ.layer_bitlen(0) - 6
                rr = np.random.default_rng(1).integers(0, (1 << 61) - 1, size=(st, 2), dtype=np.uint64)
                sess.fri_commit(rr)                                         # first call allocates the FRI buffers
                sess.prove_full(batched=True)
                t2 = time.perf_counter()
                sess.fri_commit(rr)
                pc["fri_commit_wall_sec"] = time.perf_counter() - t2
                pc["fri_commit_device_ms"] = sess.commit_device_ms()
                pc["fri_steps"] = st
            t3 = time.perf_counter(); _, ms_priv = sess.commit_private(); pc["commit_private_device_ms"] = ms_priv
            pc["commit_private_wall_sec"] = time.perf_counter() - t3
            pub = np.random.default_rng(2).integers(0, (1 << 61) - 1, size=(1 << circ.layer_bitlen(0), 2), dtype=np.uint64)
            t4 = time.perf_counter(); ms_pub = sess.commit_public(pub)[3]; pc["commit_public_device_ms"] = ms_pub
            pc["commit_public_wall_sec"] = time.perf_counter() - t4
            pc["pc_commit_side_device_ms"] = ms_priv + ms_pub + pc.get("fri_commit_device_ms", 0.0)
            pc["reference_pc_prove_sec_build_container"] = golden.get(gname_, {}).get("reference_pc_prove_sec_here")

        bit_exact = None
        ref_ops = None
        if gname in golden:
            g = golden[gname]
            ref_ops = g["mult_counter"] + g["add_counter"]
            if rank == 0:
                gold = open(os.path.join(ROOT, "tests", "golden", g["transcript"]), "rb").read()[g["gkr_slice"][0]:g["gkr_slice"][1]]
                bit_exact = (tr == gold)
        ok, _ = sess.check(tr, skip_predicates=True)
        verify = None
        if rank == 0:      # the verifier's side of the same proof (outside the timed region): O(|C|) predicate loops on host vs on the GPU
            ok_h, sec_h = sess.check(tr)
            ok_d, sec_d = sess.check(tr, device_predicates=True)
            verify = {"host_predicates_sec": sec_h, "device_predicates_sec": sec_d, "accepted": bool(ok_h and ok_d),
                      "note": "the verifier's O(|C|) loops on the device: wiring predicates (vp_predicates), gr of verifyLiu (vp_liu_gr), input-layer MLE (vp_layer_mle); the per-round checks stay on the host"}

        if rank == 0:
            sec_per_proof_job = elapsed / a.steps                      # wall time of one step (all ranks in parallel)
            ops_total = (ref_ops or 0) * proofs
            line = {
                "metric": "prover sec + field-ops/sec, SHA-256 circuit, 1/2/4/8 MI355X (bit-exact)",
                "value": ops_total / elapsed if ref_ops else None,
                "unit": "field-ops/s",
                "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": 1e3 * sec_per_proof_job,
                "higher_is_better": True, "scaling": "strong" if shard else "weak", "vs_baseline": None,
                "dtype": "u64 (F_p^2, p=2^61-1)", "data": "synthetic",
                "config": {"workload": ("SHA-256 %d-block circuit (SHA256_64.pws x%d, %d gates, %d layers), GKR sumcheck on GPU, Virgo PC off"
                                        % (a.blocks, a.blocks, circ.gates, circ.layers)) if not a.randomize else
                                       ("layeredCircuit::randomize(%d, %d) (%d gates, %d layers), GKR sumcheck on GPU, Virgo PC off"
                                        % (a.randomize[0], a.randomize[1], circ.gates, circ.layers)),
                           "mode": "batched (verifier tape pre-drawn; transcript identical to the interactive run)",
                           "proofs_per_step": 1 if shard else world, "field_ops_per_proof": ref_ops},
                "prover_sec": sec_per_proof_job,
                "prover_sec_device": 1e-3 * dev_ms / a.steps,
                "rounds": res["rounds"], "kernel_launches_per_proof": res["launches"],
                "bit_exact_vs_reference_golden": bit_exact, "host_verifier_accepts": ok,
                "golden_origin": (golden[gname].get("origin", "the real reference binary (oracle/_ref/ref_run)") if gname in golden else None),
                "interactive_path": interactive, "circuit_upload_sec": upload_sec, "verifier": verify,
            }
            if res["fold_launches"]:
                avg_ms = res["fold_ms"] / res["fold_launches"]
                gbps = res["fold_bytes"] / (res["fold_ms"] * 1e-3) / 1e9
                traffic = None
                pmc_file = PMC_SUMMARY if a.blocks == 64 else PMC_SUMMARY.replace("_b64", "_b%d" % a.blocks)
                try:        # PMC pass of the same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs), committed summary
                    if not a.randomize:
                        traffic = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))["sumfold_avg_hbm_bytes_per_launch"]
                except Exception:
                    pass
                line["roofline"] = {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                    "frac": gbps / HBM_PEAK_GBPS, "traffic": traffic,
                                    "traffic_source": "profiles/%s (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)" % pmc_file if traffic else None,
                                    "kernel": "k_sumfold3b_multi (every launch; single-stream replay of the same proof)",
                                    "single_stream_proof_ms": res.get("serial_device_ms"),
                                    "launches": res["fold_launches"], "avg_launch_us": 1e3 * avg_ms,
                                    "algorithmic_bytes_per_launch": res["fold_bytes"] / res["fold_launches"]}
            else:
                line["roofline"] = None
            if pc is not None:
                line["polynomial_commitment"] = pc
            if shard_info is not None:
                line["sharded_proof"] = shard_info
            if shard_sim is not None:
                line["sharded_proof_simulation"] = shard_sim
            if world == 1 and not a.no_cpu_baseline:
                cb = cpu_baseline(pws, a.blocks, ref_ops)
                cb["host_cpu"] = cpu_model()
                cb["host_cores_visible"] = os.cpu_count()
                line["cpu_baseline"] = cb
            print(json.dumps(line), flush=True)
        sess.close()
        circ.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
