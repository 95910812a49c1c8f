#!/usr/bin/env python3
"""bench.py — headline benchmark: GKR prover seconds and field-ops/s on the SHA-256 64-block circuit
(BASELINE.json configs[1]: "SHA-256 64-block circuit, 1xMI355X, sumcheck rounds on GPU, Virgo PC off").

A step = one complete GKR proof (Vres + three sumchecks per layer, 691 rounds) by the device prover, with
the circuit, the witness and the verifier tape already resident in HBM.  Each rank proves its own instance
(rank r draws its witness after srandom(1 + r); rank 0's instance is the golden one), so N GPUs produce N
independent proofs per step with no data-path collective ("scaling": "weak").  Rank 0's transcript is
compared byte for byte with the real reference's golden transcript on every run.

The ONE JSON line (rank 0):
  metric / value / unit / ms_per_step ...   the contract fields; value = field-ops of all ranks' proofs / max-over-ranks time
  roofline            the kernel with the largest share of the proof's summed launch time: algorithmic bytes per launch / mean launch
                      duration, both measured live (every launch of the plan bracketed with HIP events on its own stream in a
                      single-stream replay, vp_set_profiling / vp_get_launch_stats), against 8 TB/s; `traffic` from the committed
                      rocprofv3 PMC summary of this command (profiles/r02_pmc_summary_b*.json); `measured_limiter` says what the
                      counters show (VALU issue, not bytes)
  kernels             the same table for every kernel kind of the proof (launches, us, share, algorithmic MB, GB/s, frac)
  interactive_path    the drop-in entry points (one vp_round per verifier message): prover seconds by the reference's definition,
                      split into init / round / finalize calls
  circuit_upload_sec  host flatten + vp_circuit_upload (index structures built on the device) + vp_evaluate
  verifier            the host verifier's O(|C|) loops on the host and on the device
  cpu_baseline        the real reference (oracle/_ref/ref_run) on one host core, one full proof of the same circuit
  x1024_with_pc       N = 1 default run only: BASELINE configs[2], the largest single-GPU configuration, as a nested leg with the
                      same fields (transcript against the oracle fixture, roofline of ITS dominant kernel, per-launch table,
                      cpu_baseline = the oracle port) plus `polynomial_commitment`: the complete protocol (commit_private,
                      commit_public, FRI commit phase and queries, unbroken verifyFull with the reference's challenge schedule),
                      per-kernel table and rooflines of k_leaf_hash (Keccak-f/s against the VALU issue bound of its instruction
                      mix) and the NTT family (F-multiplications/s against the chip's F-multiply issue rate)

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
# PMC summaries of this command (tools/gpu_profile.sh -> tools/pmc_summary.py: separate rocprofv3 --pmc passes for FETCH_SIZE and
# WRITE_SIZE, FETCH_SIZE doubled for gfx950 as the guide's HBM section prescribes), newest first
PMC_SUMMARIES = ("r02_pmc_summary_b%d.json", "r01_l_pmc_summary_b%d.json")
# Issue-rate ceilings of the two compute-bound kernel families of the commitment, measured with tools/micro_rates.hip on MI355X
# (profiles/r02_micro_rates.txt; 256 CUs x 4 SIMDs at 2.4 GHz, 8 waves per SIMD):
#   F_p^2 multiply (31-bit split form, 16 v_mad_u64_u32 + Mersenne folds): 6.1e11 per second for the whole chip;
#   Keccak-f[1600] as 24 rounds x 180 VALU instructions (v_bitop3_b32 / v_alignbit_b32) at the measured issue cost of those two.
FMUL_PEAK_PER_S = 7.0e11        # f_mul: 224.8 SIMD-cycles per wave-multiply
# one Keccak round = 120 v_bitop3_b32 (3.20 cycles per wave-instruction per SIMD) + 58 v_alignbit_b32 (4.43) + 2 v_xor_b32 (2.78)
KECCAK_CYCLES_PER_WAVE_PERM = 24 * (120 * 3.20 + 58 * 4.43 + 2 * 2.78)
KECCAK_PEAK_PER_S = 1024 * 64 * 2.4e9 / KECCAK_CYCLES_PER_WAVE_PERM


def pmc_traffic(blocks, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary of this command at this size, or (None, None)."""
    for pat in PMC_SUMMARIES:
        f = os.path.join(ROOT, "profiles", pat % blocks)
        try:
            for k in json.load(open(f))["kernels"]:
                if k["kernel"].replace("vp::", "") == kernel and "hbm_bytes_per_launch" in k:
                    return k["hbm_bytes_per_launch"], "profiles/" + os.path.basename(f)
        except Exception:
            continue
    return None, None


def launch_table(stats, total_us=None):
    """Per-kernel and per-launch views of a vp_get_launch_stats table: achieved algorithmic GB/s against the 8 TB/s HBM roofline,
    share of the summed launch time."""
    tot = sum(e["us"] for e in stats) or 1e-9
    kern = {}
    for e in stats:
        k = kern.setdefault(e["kernel"], {"kernel": e["kernel"], "launches": 0, "us": 0.0, "bytes": 0, "work": 0, "workgroups": 0})
        k["launches"] += 1; k["us"] += e["us"]; k["bytes"] += e["bytes"]; k["work"] += e["work"]; k["workgroups"] += e["workgroups"]
    rows = []
    for k in sorted(kern.values(), key=lambda x: -x["us"]):
        gbps = k["bytes"] / (k["us"] * 1e-6) / 1e9 if k["us"] > 0 else 0.0
        rows.append({"kernel": k["kernel"], "launches": k["launches"], "total_us": round(k["us"], 2), "time_share": round(k["us"] / tot, 4),
                     "avg_launch_us": round(k["us"] / k["launches"], 2), "algorithmic_MB_per_launch": round(k["bytes"] / k["launches"] / 1e6, 3),
                     "GBps": round(gbps, 1), "hbm_frac": round(gbps / HBM_PEAK_GBPS, 4), "work_units": k["work"]})
    per_launch = []
    for e in stats:
        gbps = e["bytes"] / (e["us"] * 1e-6) / 1e9 if e["us"] > 0 else 0.0
        per_launch.append({"step": e["step"], "kernel": e["kernel"], "jobs": e["jobs"], "workgroups": e["workgroups"],
                           "rounds": ([e["first_round"], e["first_round"] + e["rounds"] - 1] if e["rounds"] else None),
                           "MB": round(e["bytes"] / 1e6, 3), "us": round(e["us"], 2), "GBps": round(gbps, 1), "hbm_frac": round(gbps / HBM_PEAK_GBPS, 4)})
    return rows, per_launch, tot


def roofline_of(rows, blocks, serial_ms, note=None):
    """The `roofline` object for the kernel with the largest share of the (single-stream) proof time."""
    if not rows:
        return None
    d = rows[0]
    traffic, src = pmc_traffic(blocks, d["kernel"])
    limiter = None
    if "sumfold" in d["kernel"] or "light" in d["kernel"]:
        limiter = ("VALU issue: SQ_INSTS_VALU per launch / 1024 SIMDs / launch cycles = one wave-instruction per 5.9-6.5 cycles per SIMD "
                   "(profiles/r02_pmc_summary_b64.json, r02_pmc_summary_b1024.json) where this instruction mix issues at 4.3-5 cycles when nothing "
                   "stalls (tools/micro_rates.hip: v_mad_u64_u32 7.0, 64-bit add 5.2, 32-bit ops 2.8-3.2): 75-85 % of the issue slots; the bytes "
                   "moved equal the algorithmic bytes, the integer multiply-add of F_p^2 (16 v_mad_u64_u32 + Mersenne folds) sets the time")
    accounting = None
    if "sumfold" in d["kernel"] and d["work_units"]:
        # The byte figure above is the strict one: a launch reads its tables once and writes the folded tables once, whatever the number of
        # rounds it fuses (and the init-generating variant never writes or reads the mult/add tables at full length at all).  SURVEY.md
        # §8d's per-unit figure is per ROUND: 48 B x (entries in + entries out) = 144 B per pair step with three table families (96 B in the
        # Liu phase, which has no add table: counted as 144 here, so this is an upper figure).  Same launches, same time:
        survey_bytes = 144.0 * d["work_units"]
        survey_gbps = survey_bytes / (d["total_us"] * 1e-6) / 1e9
        accounting = {"note": "informative only, NOT `frac`: the same kernel time priced with SURVEY.md 8d's per-round byte formula (every round reads and "
                              "writes its tables) instead of the bytes the fused launch has to move",
                      "pair_steps": d["work_units"], "bytes_per_pair_step": 144, "GBps": round(survey_gbps, 1), "frac_of_hbm_peak": round(survey_gbps / HBM_PEAK_GBPS, 4)}
    return {"bound": "hbm", "achieved": d["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": d["hbm_frac"], "traffic": traffic,
            "per_round_accounting_survey_8d": accounting,
            "traffic_source": ("%s (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, per launch)" % src) if traffic else None,
            "kernel": d["kernel"], "kernel_time_share": d["time_share"], "launches": d["launches"], "avg_launch_us": d["avg_launch_us"],
            "algorithmic_bytes_per_launch": d["algorithmic_MB_per_launch"] * 1e6, "single_stream_proof_ms": serial_ms,
            "measured_limiter": limiter,
            "how": "every launch of the plan bracketed with HIP events in a single-stream replay of the same proof (vp_set_profiling / vp_get_launch_stats)" + (("; " + note) if note else "")}



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


def profile_gkr(sess, tr_expected, blocks):
    """Roofline pass (outside the timed region, same process, same resident state): the proof is replayed on ONE stream with HIP
    events around EVERY launch (in the timed steps the independent sumchecks overlap on several streams, which makes per-kernel
    event times meaningless there).  Returns (per-kernel rows, per-launch rows, roofline object, single-stream device ms)."""
    sess.set_profiling(1)
    tr_p, res_p = sess.prove_gkr()
    stats = sess.launch_stats()
    sess.set_profiling(0)
    assert tr_p == tr_expected, "profiled replay produced a different transcript"
    rows, per_launch, _ = launch_table(stats)
    return rows, per_launch, roofline_of(rows, blocks, res_p["gkr_device_ms"]), res_p


def pc_leg(vp, sess, circ, golden, gname, full_fixture=None):
    """BASELINE.json configs[2] flavour: the commitment's commit side on the GPU (never part of `value`).  Times commit_private,
    commit_public on the PROTOCOL's public vector (eq table of the last Liu point, src/verifier.cpp:368-379) and the FRI commit
    phase, compares everything a fixture exists for, and profiles each of the three calls launch by launch."""
    import numpy as np
    from conftest import GOLDEN
    pc = {}
    t1 = time.perf_counter()
    full, okf = sess.prove_full(batched=True)                 # commit_private + GKR + commit_public(eq(r_liu, .))
    pc["full_proof_wall_sec"] = time.perf_counter() - t1
    pc["full_transcript_verified"] = okf
    pc["commit_public_device_ms"] = sess.commit_device_ms()       # last commitment call of prove_full
    fri_ref = None
    if gname in golden and "fri" in golden[gname]:            # recorded runs of the real reference (full transcript + FRI steps)
        gg = golden[gname]
        pc["full_transcript_bit_exact"] = (full == open(os.path.join(GOLDEN, gg["transcript"]), "rb").read())
        fri = open(os.path.join(GOLDEN, gg["fri"]), "rb").read()
        st = gg["fri_steps"]
        rec = np.frombuffer(fri[:48 * st], dtype=np.uint64).reshape(st, 6)
        rr = np.ascontiguousarray(rec[:, :2])
        fri_ref = b"".join(rec[i, 2:].tobytes() for i in range(st))
        pc["fixture"] = "real reference (tests/golden/%s, %s)" % (gg["transcript"], gg["fri"])
    elif full_fixture is not None:                            # oracle run of the whole protocol at this size (make_oracle_fixture_full.py)
        fx = open(full_fixture, "rb").read()
        st = circ.layer_bitlen(0) - 6
        n_full = len(full)
        pc["full_transcript_bit_exact"] = (fx[:n_full] == full)
        rr = np.frombuffer(fx[n_full + 32 * st + 2048 * 16:n_full + 32 * st + 2048 * 16 + 16 * st], dtype=np.uint64).reshape(st, 2).copy()
        fri_ref = fx[n_full:n_full + 32 * st]
        pc["fixture"] = "oracle, whole protocol incl. the reference's FRI challenges (tests/golden/%s)" % os.path.basename(full_fixture)
    else:
        st = circ.layer_bitlen(0) - 6
        rr = np.random.default_rng(1).integers(0, (1 << 61) - 1, size=(st, 2), dtype=np.uint64)
    sess.fri_commit(rr)                                        # first call allocates nothing new, warms the kernels
    sess.prove_full(batched=True)
    t2 = time.perf_counter()
    roots, fin = sess.fri_commit(rr)
    pc["fri_commit_wall_sec"] = time.perf_counter() - t2
    pc["fri_commit_device_ms"] = sess.commit_device_ms()
    pc["fri_steps"] = int(st)
    if fri_ref is not None:
        pc["fri_roots_bit_exact"] = (roots == fri_ref)
    t3 = time.perf_counter(); _, ms_priv = sess.commit_private(); pc["commit_private_device_ms"] = ms_priv
    pc["commit_private_wall_sec"] = time.perf_counter() - t3
    pc["pc_commit_side_device_ms"] = ms_priv + pc["commit_public_device_ms"] + pc["fri_commit_device_ms"]
    pc["reference_pc_prove_sec_build_container"] = golden.get(gname, {}).get("reference_pc_prove_sec_here")
    # per-launch profile of the three calls (events on the library stream; the commitment runs on one stream anyway)
    sess.set_profiling(1)
    sess.commit_private(); st_priv = sess.launch_stats()
    sess.prove_full(batched=True); st_pub = sess.launch_stats()      # the last profiled call inside is commit_public
    sess.fri_commit(rr); st_fri = sess.launch_stats()
    sess.set_profiling(0)
    allst = st_priv + st_pub + st_fri
    rows, _, tot = launch_table(allst)
    pc["kernels"] = rows
    pc["profiled_device_ms"] = tot * 1e-3
    leaf = [e for e in allst if e["kernel"] == "k_leaf_hash"]
    ntt = [e for e in allst if e["kernel"] in ("k_ntt_lds", "k_ntt_split")]
    rl = {}
    if leaf:
        w, us = sum(e["work"] for e in leaf), sum(e["us"] for e in leaf)
        rl["k_leaf_hash"] = {"bound": "valu", "achieved": w / (us * 1e-6), "peak": KECCAK_PEAK_PER_S, "unit": "Keccak-f[1600]/s",
                             "frac": w / (us * 1e-6) / KECCAK_PEAK_PER_S, "time_share": us / tot,
                             "peak_definition": "issue bound of the kernel's own instruction mix: 1024 SIMDs x 64 lanes x 2.4 GHz / %.0f cycles per wave-permutation (24 rounds x (120 v_bitop3_b32 x 3.20 + 58 v_alignbit_b32 x 4.43 + 2 v_xor_b32 x 2.78 cycles), profiles/r02_micro_rates.txt)" % KECCAK_CYCLES_PER_WAVE_PERM}
    if ntt:
        w, us = sum(e["work"] for e in ntt), sum(e["us"] for e in ntt)
        rl["k_ntt"] = {"bound": "valu", "achieved": w / (us * 1e-6), "peak": FMUL_PEAK_PER_S, "unit": "F_p^2 multiplications/s",
                       "frac": w / (us * 1e-6) / FMUL_PEAK_PER_S, "time_share": us / tot,
                       "hbm_GBps": sum(e["bytes"] for e in ntt) / (us * 1e-6) / 1e9,
                       "peak_definition": "chip-wide F_p^2 multiply issue rate of the 31-bit split form (tools/micro_rates.hip, f_mul)"}
    pc["rooflines"] = rl
    return pc


def gkr_leg(vp, circ, sess, steps, warmup, world, shard, local):
    """Timed region of one configuration: `warmup` untimed proofs, then exactly `steps` proofs bracketed by barrier + device sync."""
    in_lib = shard == "rccl"       # communicator attached inside libvpgpu (Session.attach_comm): prove_gkr returns the assembled transcript
    for _ in range(warmup):
        tr, _ = sess.prove_gkr()
        if shard and not in_lib:
            vp.allreduce_transcript(tr, local)
    gpu_sync(local)
    barrier(world)
    t0 = time.perf_counter()
    dev_ms = 0.0
    res = None
    for _ in range(steps):
        tr, res = sess.prove_gkr()
        if shard and not in_lib:       # gloo rehearsal: the same u64 sum through torch (on the GPUs it happens inside prove_gkr, over RCCL)
            tr = vp.allreduce_transcript(tr, local)
        dev_ms += res["gkr_device_ms"]
    gpu_sync(local)
    barrier(world)
    elapsed = time.perf_counter() - t0
    return tr, res, elapsed, dev_ms


def x1024_leg(vp, pws, golden, a, local):
    """BASELINE.json configs[2]: SHA-256 1024-block circuit (102 M gates, 859 rounds, tables up to 2^26), sumcheck + Virgo commitment
    on ONE MI355X — the largest single-GPU configuration, as a nested leg of the default run.  Same contract as the headline leg:
    warmup, timed steps, transcript compared with the committed oracle fixture, per-launch table with the dominant kernel's roofline,
    CPU baseline (the oracle port's GKR proof of the same circuit, one core) timed in the same run."""
    B = 1024
    g = golden["sha256_x%d" % B]
    t_b = time.perf_counter()
    circ = vp.Circuit.from_pws(pws, B, seed=1)
    build_sec = time.perf_counter() - t_b
    t_up = time.perf_counter()
    sess = vp.Session(circ, device=local)
    upload_sec = time.perf_counter() - t_up
    sess.draw_tape()
    steps, warmup = max(3, a.steps // 2), 2
    tr, res, elapsed, dev_ms = gkr_leg(vp, circ, sess, steps, warmup, 1, False, local)
    gold = open(os.path.join(ROOT, "tests", "golden", g["transcript"]), "rb").read()[g["gkr_slice"][0]:g["gkr_slice"][1]]
    ref_ops = g["mult_counter"] + g["add_counter"]
    rows, per_launch, roof, res_p = profile_gkr(sess, tr, B)
    ok_d, sec_d = sess.check(tr, device_predicates=True)
    t_i = time.perf_counter()
    tr_i, res_i, ok_i = sess.prove_interactive()
    inter = {"prover_sec": res_i["prove_sec"], "init_calls_sec": res_i.get("init_sec"), "round_calls_sec": res_i.get("round_sec"),
             "finalize_calls_sec": res_i.get("finalize_sec"), "wall_sec_with_host_verifier": time.perf_counter() - t_i, "transcript_equals_batched": tr_i == tr, "verified": ok_i}
    sess.draw_tape()
    fx = os.path.join(ROOT, "tests", "golden", "oracle_sha256_x1024_full.bin")
    pc = pc_leg(vp, sess, circ, golden, "sha256_x%d" % B, full_fixture=fx if os.path.exists(fx) else None)
    leg = {"config": {"workload": "SHA-256 %d-block circuit (SHA256_64.pws x%d, %d gates, %d layers), GKR sumcheck + Virgo FFT/LDT commit on GPU (BASELINE configs[2])"
                                  % (B, B, circ.gates, circ.layers), "field_ops_per_proof": ref_ops},
           "value": ref_ops * steps / elapsed, "unit": "field-ops/s", "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
           "prover_sec": elapsed / steps, "prover_sec_device": 1e-3 * dev_ms / steps, "rounds": res["rounds"],
           "kernel_launches_per_proof": res["launches"], "bit_exact_vs_reference": tr == gold, "golden_origin": g.get("origin"),
           "reference_prove_sec_build_container": g.get("reference_prove_sec_here"), "reference_pc_prove_sec_build_container": g.get("reference_pc_prove_sec_here"),
           "verifier_accepts_full_check": bool(ok_d), "verify_sec_device_predicates": sec_d, "interactive_path": inter,
           "circuit_build_sec": build_sec, "circuit_upload_sec": upload_sec,
           "roofline": roof, "kernels": rows, "per_launch": per_launch, "polynomial_commitment": pc}
    sess.close(); circ.close()
    if not a.no_cpu_baseline:
        import oracle_binding as ob
        t0 = time.perf_counter()
        oc = ob.Circuit.from_pws(pws, B, seed=1)
        t1 = time.perf_counter()
        otr, st = oc.prove_gkr()
        oc.close()
        ops = st["mult_count"] + st["add_count"]
        leg["cpu_baseline"] = {"value": ops / st["prove_sec"], "unit": "field-ops/s", "cores": 1, "kind": "port",
                               "sample": "one full GKR proof of the same 1024-block circuit by the oracle port (PC off), single thread",
                               "prover_sec": st["prove_sec"], "field_ops": ops, "circuit_build_sec": t1 - t0,
                               "transcript_equals_gpu": otr == tr, "host_cpu": cpu_model(), "host_cores_visible": os.cpu_count()}
    return leg


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
    ap.add_argument("--shard-split", type=int, default=0, metavar="MIN_LOG",
                    help="with --shard-sim or --shard-chains: also split tables of at least 2^(log2 W + MIN_LOG) entries by index over the ranks (vp_set_shard_split; 11 is the smallest useful value)")
    ap.add_argument("--with-pc", action="store_true", help="also time the Virgo commitment (commit_private + commit_public + FRI commit phase)")
    ap.add_argument("--no-x1024-leg", action="store_true",
                    help="skip the nested x1024_with_pc leg (BASELINE configs[2]) that the default single-GPU run appends to the x64 headline line")
    ap.add_argument("--per-launch", action="store_true", help="include the per-launch table of the headline leg in the JSON line (the per-kernel table is always there)")
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
        # one-time cost of the process (HIP context, code objects: ~0.25 s) paid by a 3-gate circuit first, so that circuit_upload_sec is
        # what a caller sees per circuit (host flatten + vp_circuit_upload with its device-side list building + vp_evaluate)
        t_up = time.perf_counter()
        c0 = vp.Circuit.randomize(2, 1, seed=1); s0 = vp.Session(c0, device=local); s0.close(); c0.close()
        first_use_sec = time.perf_counter() - t_up
        t_up = time.perf_counter()
        sess = vp.Session(circ, device=local)            # raises without the HIP library / GPU
        upload_sec = time.perf_counter() - t_up
        sess.draw_tape()
        if shard:
            sess.set_shard(rank, world)
            import torch.distributed as dist
            if "nccl" in dist.get_backend():           # the data-path collective lives in the C ABI: RCCL on the device buffer, no torch tensor
                sess.attach_comm(rank, world)
                shard = "rccl"
            if a.shard_split:
                sess.set_shard_split(a.shard_split)
                if shard != "rccl":
                    raise SystemExit("--shard-split in the multi-rank mode needs the in-library communicator (RCCL); rehearse it with --shard-sim")
        tr, res, elapsed, dev_ms = gkr_leg(vp, circ, sess, a.steps, a.warmup, world, shard, local)
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
                          "chains_split_by_index": int((owner == -1).sum()),
                          "collective": ("one RCCL all-reduce (u64 sum) of the transcript%s per proof inside vp_prove_gkr (vp_comm_init: no torch tensor, no host bounce)"
                                         % (" + export area" if a.shard_split else "")) if shard == "rccl"
                                        else "one all-reduce (sum, int64) of the %d-byte transcript per proof through torch, backend %s" % (len(tr), dist.get_backend())}
            sess.set_shard(0, 1)                       # the roofline / verifier legs below run the whole proof on every rank
            tr_full, _ = sess.prove_gkr()
            assert tr_full == tr, "assembled sharded transcript differs from the unsharded proof"
        rows, per_launch, roof, res_p = profile_gkr(sess, tr, a.blocks if not a.randomize else 0)

        shard_sim = None
        if a.shard_sim > 1 and world == 1:
            # per-rank compute of a chain-sharded proof on W GPUs, measured shard by shard on this one GPU (no collective here)
            per = []
            parts = []
            for r in range(a.shard_sim):
                sess.set_shard(r, a.shard_sim)
                if a.shard_split:
                    sess.set_shard_split(a.shard_split)
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
            summed = vp.sum_transcripts(parts)
            t_f = time.perf_counter()
            assembled = sess.shard_finish(summed) if a.shard_split else summed
            finish_ms = 1e3 * (time.perf_counter() - t_f)
            sess.set_shard(0, 1)
            shard_sim = {"world": a.shard_sim, "per_rank": per, "max_device_ms": max(x["device_ms"] for x in per),
                         "max_wall_ms": max(x["wall_ms"] for x in per),
                         "index_split_min_log": a.shard_split or None, "chains_split_by_index": int((owner == -1).sum()),
                         "host_finish_ms": finish_ms if a.shard_split else None,
                         "cost_share_per_rank": [float(cost[owner == r].sum() / cost.sum()) for r in range(a.shard_sim)],
                         "assembled_equals_unsharded": assembled == tr,
                         "note": "each shard run alone on this GPU; a W-GPU run adds one all-reduce of the transcript (and, with the index split, of the export area) per proof"}

        interactive = None
        if rank == 0:
            # the drop-in path of the reference's own call pattern (one vp_round per verifier message), outside the timed region
            t_i = time.perf_counter()
            tr_i, res_i, ok_i = sess.prove_interactive()
            interactive = {"prover_sec": res_i["prove_sec"], "init_calls_sec": res_i.get("init_sec"), "round_calls_sec": res_i.get("round_sec"),
                           "finalize_calls_sec": res_i.get("finalize_sec"), "wall_sec_with_host_verifier": time.perf_counter() - t_i,
                           "transcript_equals_batched": tr_i == tr, "verified": ok_i,
                           "note": "reference definition of Prove Time (sum of prover-method spans) over the interactive entry points (vp_round per verifier message)"}
            sess.draw_tape()

        pc = None
        if a.with_pc and rank == 0:
            pc = pc_leg(vp, sess, circ, golden, gname)

        bit_exact = None
        ref_ops = None
        if gname in golden:
            g = golden[gname]
            ref_ops = g["mult_counter"] + g["add_counter"]
            if rank == 0:
                gold = open(os.path.join(ROOT, "tests", "golden", g["transcript"]), "rb").read()[g["gkr_slice"][0]:g["gkr_slice"][1]]
                bit_exact = (tr == gold)
        # the reported flag is the FULL replay check (per-round identities, wiring predicates and getFinalValue on the device, Liu
        # check, input check): the per-round identities alone hold by construction for the rounds whose b is derived
        ok, sec_d = sess.check(tr, device_predicates=True)
        verify = None
        if rank == 0:      # the verifier's side of the same proof (outside the timed region): O(|C|) predicate loops on host vs on the GPU
            ok_h, sec_h = sess.check(tr)
            verify = {"host_predicates_sec": sec_h, "device_predicates_sec": sec_d, "accepted": bool(ok_h and ok),
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
                "bit_exact_vs_reference_golden": bit_exact, "host_verifier_accepts": bool(ok),
                "host_verifier_check": "full replay: per-round identities, wiring predicates + getFinalValue (device loops), Liu check, input-layer check",
                "golden_origin": (golden[gname].get("origin", "the real reference binary (oracle/_ref/ref_run)") if gname in golden else None),
                "interactive_path": interactive, "circuit_upload_sec": upload_sec, "process_first_use_sec": first_use_sec, "verifier": verify,
                "roofline": roof, "kernels": rows,
            }
            if a.per_launch:
                line["per_launch"] = per_launch
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
        sess.close()
        circ.close()
        if rank == 0:
            if world == 1 and a.blocks == 64 and not a.randomize and not a.no_x1024_leg and not shard and "sha256_x1024" in golden:
                line["x1024_with_pc"] = x1024_leg(vp, pws, golden, a, local)
            print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
