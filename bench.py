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
PMC_SUMMARIES = ("r03_pmc_summary_b%d.json", "r02_pmc_summary_b%d.json", "r01_l_pmc_summary_b%d.json")
# Issue-rate ceilings of the two compute-bound kernel families of the commitment, measured with tools/micro_rates.hip on MI355X
# (profiles/r02_micro_rates.txt; 256 CUs x 4 SIMDs at 2.4 GHz, 8 waves per SIMD):
#   F_p^2 multiply (31-bit split form, 16 v_mad_u64_u32 + Mersenne folds): 6.1e11 per second for the whole chip;
#   Keccak-f[1600] as 24 rounds x 180 VALU instructions (v_bitop3_b32 / v_alignbit_b32) at the measured issue cost of those two.
FMUL_PEAK_PER_S = 7.0e11        # f_mul: 224.8 SIMD-cycles per wave-multiply
# Keccak-f[1600] on 32-bit lanes: 180 VALU instructions per round is the instruction-count FLOOR (theta parity 20 three-input xors, 10 rotations by 1,
# 50 theta-apply, rho 48 64-bit rotations = 2 v_alignbit each... counted as 48 here + the 10 above = 58 v_alignbit_b32, chi 50 v_bitop3_b32, iota 2) —
# 120 v_bitop3_b32 + 58 v_alignbit_b32 + 2 v_xor_b32.  The peak prices EVERY one of those at the fastest issue cost measured for any of them
# (v_xor_b32: 2.78 SIMD-cycles per wave-instruction, tools/micro_rates.hip) — not at the cost of the kernel's own mix, against which any
# kernel would score ~1.
KECCAK_INSTR_PER_ROUND = 180
KECCAK_BEST_ISSUE_CYCLES = 2.78
KECCAK_PEAK_PER_S = 1024 * 64 * 2.4e9 / (24 * KECCAK_INSTR_PER_ROUND * KECCAK_BEST_ISSUE_CYCLES)
KECCAK_MIX_CYCLES_PER_WAVE_PERM = 24 * (120 * 3.20 + 58 * 4.43 + 2 * 2.78)      # the kernel's own mix at its measured per-instruction costs


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


# F_p^2 multiplications per pair step of the fold kernels (one pair of entries of one table family in one round): the three folds
# x0 + r (x1 - x0) of V, mult and add, and the products of the round polynomial — dm dv and m0 v0; the third product (m1 v1) only in round 1
# of a sumcheck, afterwards b comes from the previous claim (DESIGN.md §4 "five products per pair").  4 without an add table (Liu phase).
FMUL_PER_PAIR_STEP = 5


def roofline_of(rows, blocks, serial_ms, note=None):
    """The `roofline` object for the kernel with the largest share of the (single-stream) proof time.  The fold family is bound by VALU issue,
    not by HBM (PMC: traffic = algorithmic bytes, VALU busy ~80 %): its `achieved` is F_p^2-multiply-equivalents per second against the
    chip's measured F-multiply issue rate; the HBM figure of the same launches is kept beside it (hbm_frac)."""
    if not rows:
        return None
    d = rows[0]
    traffic, src = pmc_traffic(blocks, d["kernel"])
    fold = "sumfold" in d["kernel"]
    limiter = None
    if fold or "light" in d["kernel"]:
        limiter = ("VALU issue: SQ_INSTS_VALU per launch / 1024 SIMDs / launch cycles = one wave-instruction per 5.9-6.5 cycles per SIMD "
                   "(profiles/r02_pmc_summary_b64.json, r02_pmc_summary_b1024.json) where this instruction mix issues at 4.3-5 cycles when nothing "
                   "stalls (tools/micro_rates.hip: v_mad_u64_u32 7.0, 64-bit add 5.2, 32-bit ops 2.8-3.2): 75-85 % of the issue slots; the bytes "
                   "moved equal the algorithmic bytes, the integer multiply-add of F_p^2 (16 v_mad_u64_u32 + Mersenne folds) sets the time")
    accounting = None
    out = {"kernel": d["kernel"], "kernel_time_share": d["time_share"], "launches": d["launches"], "avg_launch_us": d["avg_launch_us"],
           "algorithmic_bytes_per_launch": d["algorithmic_MB_per_launch"] * 1e6, "single_stream_proof_ms": serial_ms,
           "traffic": traffic, "traffic_source": ("%s (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, per launch)" % src) if traffic else None,
           "hbm_GBps": d["GBps"], "hbm_peak_GBps": HBM_PEAK_GBPS, "hbm_frac": d["hbm_frac"], "measured_limiter": limiter,
           "how": "every launch of the plan bracketed with HIP events in a single-stream replay of the same proof (vp_set_profiling / vp_get_launch_stats)" + (("; " + note) if note else "")}
    if fold and d["work_units"]:
        fmul = FMUL_PER_PAIR_STEP * d["work_units"] / (d["total_us"] * 1e-6)
        out.update({"bound": "valu", "achieved": fmul, "peak": FMUL_PEAK_PER_S, "unit": "F_p^2 multiply-equivalents/s", "frac": fmul / FMUL_PEAK_PER_S,
                    "achieved_definition": "%d F_p^2 multiplications per pair step (3 folds + 2 products; the additions, subtractions and lazy reductions around "
                                           "them are NOT converted into multiply-equivalents, and the contribution products of the init-generating variant are "
                                           "not counted: a lower figure) x pair steps of the launches / their summed HIP-event time" % FMUL_PER_PAIR_STEP,
                    "peak_definition": "chip-wide F_p^2 multiply issue rate of the 31-bit split form, measured (tools/micro_rates.hip, profiles/r02_micro_rates.txt: "
                                       "224.8 SIMD-cycles per wave-multiply, 1024 SIMDs, 2.4 GHz)"})
        # SURVEY.md 8d's per-ROUND byte formula on the same launches (every round reads and writes its tables): 144 B per pair step with three
        # table families — informative, above what the fused launch has to move
        survey_gbps = 144.0 * d["work_units"] / (d["total_us"] * 1e-6) / 1e9
        accounting = {"note": "informative only: the same kernel time priced with SURVEY.md 8d's per-round byte formula instead of the bytes the fused launch moves",
                      "pair_steps": d["work_units"], "bytes_per_pair_step": 144, "GBps": round(survey_gbps, 1), "frac_of_hbm_peak": round(survey_gbps / HBM_PEAK_GBPS, 4)}
    else:
        out.update({"bound": "hbm", "achieved": d["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": d["hbm_frac"]})
    out["per_round_accounting_survey_8d"] = accounting
    return out


def per_round_summary(stats, full=False):
    """north_star: "achieved HBM-bandwidth fraction reported per sumcheck round" — the interactive path, where a round IS a call: algorithmic
    bytes of the round / wall time of vp_round as the verifier sees it (vp_get_round_stats)."""
    if not stats:
        return None
    names = {0: "one launch per round", 1: "resident kernel (pinned mailbox)", 2: "round 1, computed behind the init call"}
    by = {}
    for e in stats:
        k = by.setdefault(e["how"], {"rounds": 0, "us": 0.0, "bytes": 0})
        k["rounds"] += 1; k["us"] += e["us"]; k["bytes"] += e["bytes"]
    classes = [{"served_by": names.get(h, str(h)), "rounds": v["rounds"], "total_us": round(v["us"], 1), "avg_us": round(v["us"] / v["rounds"], 2),
                "algorithmic_MB": round(v["bytes"] / 1e6, 3), "GBps": round(v["bytes"] / (v["us"] * 1e-6) / 1e9, 2) if v["us"] > 0 else None}
               for h, v in sorted(by.items())]
    def row(e):
        g = e["bytes"] / (e["us"] * 1e-6) / 1e9 if e["us"] > 0 else 0.0
        return {"layer": e["layer"], "phase": e["phase"], "round": e["round"], "tables": e["tables"], "served_by": e["how"], "MB": round(e["bytes"] / 1e6, 3),
                "us": round(e["us"], 2), "GBps": round(g, 1), "hbm_frac": round(g / HBM_PEAK_GBPS, 4)}
    big = sorted((e for e in stats if e["how"] == 0), key=lambda e: -e["bytes"])[:12]
    tot_b, tot_us = sum(e["bytes"] for e in stats), sum(e["us"] for e in stats)
    out = {"rounds": len(stats), "algorithmic_MB": round(tot_b / 1e6, 2), "total_us": round(tot_us, 1), "GBps_overall": round(tot_b / (tot_us * 1e-6) / 1e9, 1),
           "hbm_frac_overall": round(tot_b / (tot_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4), "by_path": classes, "largest_rounds": [row(e) for e in big],
           "bytes_definition": "SURVEY.md 8d: 48 B x (L_in + L_out) per table family (32 B in the Liu phase); round 1 only reads",
           "time_definition": "wall time of the vp_round call (launch or mailbox round trip + arithmetic + reply)"}
    if full:
        out["all_rounds"] = [row(e) for e in stats]
    return out


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
        import torch
        if rccl and os.environ.get("VP_BENCH_BACKEND", "") != "gloo" and torch.cuda.device_count() >= world:
            torch.cuda.set_device(local)
            dist.init_process_group(backend="cpu:gloo,cuda:nccl", rank=rank, world_size=world)
        else:       # VP_BENCH_BACKEND=gloo: rehearsal of the sharded mode on a box with fewer GPUs than ranks (RCCL needs one GPU per rank)
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    return world, rank, local


class stdout_to_stderr:
    """Gloo announces its connections on the C-level stdout ("[Gloo] Rank 0 is connected to 1 peer ranks ...") when the process group forms:
    stdout carries ONE JSON line and nothing else, so file descriptor 1 points at stderr while the group is set up."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


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
    # "Polynomial commitment: prove time" exactly as the reference defines it (src/verifier.cpp:183): poly_prover.total_time = commit_private
    # + commit_public + commit_phase (lib/virgo/src/poly_commit.h:43,121,336,345, vpd_verifier.cpp:70) + fft_gkr's prover time (:92-94), host
    # wall clock around the prover calls of ONE unbroken run of the complete protocol (33 query repetitions; answering the queries is not part
    # of the number in the reference either)
    trf, okf2, times = sess.prove_and_verify_full(reps=33)
    pc["prove_sec"] = times["pc_prove_sec"]
    pc["prove_sec_definition"] = "commit_private + commit_public + fft_gkr prover + FRI commit phase, host wall clock (reference: src/verifier.cpp:183, vpd_verifier.cpp:70,92-95)"
    pc["fft_gkr_sec"] = times["pc_fft_gkr_sec"]
    pc["query_answer_sec"] = times["pc_query_answer_sec"]
    pc["complete_protocol"] = {"accepted": bool(okf2), "transcript_equals_batched_run": trf == full, "gkr_prove_sec_interactive": times["gkr_prove_sec"],
                               "verify_sec": times["verify_sec"], "query_repetitions": 33}
    if gname in golden and os.path.exists(os.path.join(GOLDEN, "fftgkr_%s.bin" % gname)):
        pc["fft_gkr_bit_exact_vs_reference_record"] = sess.last_fft_gkr() == open(os.path.join(GOLDEN, "fftgkr_%s.bin" % gname), "rb").read()
    sess.draw_tape()
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
                             "frac_of_own_instruction_mix": w / (us * 1e-6) / (1024 * 64 * 2.4e9 / KECCAK_MIX_CYCLES_PER_WAVE_PERM),
                             "peak_definition": "instruction-count floor x best issue rate: 24 rounds x %d VALU instructions (the 32-bit minimum: 120 v_bitop3_b32 + 58 v_alignbit_b32 + 2 v_xor_b32) "
                                                "x %.2f SIMD-cycles per wave-instruction (the cheapest measured, tools/micro_rates.hip) on 1024 SIMDs x 64 lanes at 2.4 GHz; "
                                                "frac_of_own_instruction_mix prices the same instructions at their own measured costs (3.20 / 4.43 / 2.78)" % (KECCAK_INSTR_PER_ROUND, KECCAK_BEST_ISSUE_CYCLES)}
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


def two_in_flight_leg(vp, circ, sess, tr_expected, steps, warmup, local, ref_ops):
    """Throughput of a prover that serves a QUEUE of proofs of one circuit: two sessions (two contexts, each with its own tables, tape and launch
    plan) driven by two host threads, so that one proof's latency tail (the single-workgroup closing launches, the ramp of its first launches)
    is filled by the other proof's kernels.  Not the headline: `value` stays one proof at a time; this is the same work, 2 x steps proofs,
    timed as a whole.  Both sessions' transcripts are compared with the headline's."""
    import threading
    other = vp.Session(circ, device=local)
    other.draw_tape()
    pair = (sess, other)
    for s_ in pair:
        for _ in range(max(1, warmup)):
            s_.prove_gkr()
    out = [None, None]
    err = []

    def run(i):
        try:
            tr = None
            for _ in range(steps):
                tr, _ = pair[i].prove_gkr()
            out[i] = tr
        except Exception as e:                                  # a failed thread must not look like a fast one
            err.append(repr(e))

    gpu_sync(local)
    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    gpu_sync(local)
    wall = time.perf_counter() - t0
    other.close()
    ok = not err and out[0] == tr_expected and out[1] == tr_expected
    return {"proofs_in_flight": 2, "proofs": 2 * steps, "wall_sec": wall, "ms_per_proof": 1e3 * wall / (2 * steps),
            "value": (ref_ops * 2 * steps / wall) if (ref_ops and ok) else None, "unit": "field-ops/s",
            "transcripts_equal_headline": bool(ok), "errors": err or None,
            "note": "two sessions of the same circuit on one GPU, one host thread each (include/vpgpu.h, Threads: different contexts run concurrently); a proof's latency is "
                    "unchanged (prover_sec), the GPU's idle tails are filled"}


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
             "finalize_calls_sec": res_i.get("finalize_sec"), "wall_sec_with_host_verifier": time.perf_counter() - t_i, "transcript_equals_batched": tr_i == tr, "verified": ok_i,
             "per_round": per_round_summary(sess.round_stats())}
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


def spawn_ranks(n):
    """`bench.py --gpus N` (N > 1) started WITHOUT a launcher: start the N ranks here, one process per GPU, exactly as the driver's command
    would (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>), and exit with its code.  This runs
    at the very top of main(), before any HIP / torch.cuda call of this process (a process that has touched the GPU must not start or
    become another GPU program on this pool).  Never falls back to one rank: too few GPUs is an error, not an `n_gpus: 1` line."""
    import socket
    import torch
    have = torch.cuda.device_count()             # counting devices does not initialise the GPU
    if have < n and os.environ.get("VP_BENCH_BACKEND", "") != "gloo":
        sys.stderr.write("bench.py: --gpus %d but %d GPU(s) visible (set VP_BENCH_BACKEND=gloo to REHEARSE more ranks than GPUs; that line says so)\n" % (n, have))
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


class Watchdog:
    """The multi-rank sub-legs run collectives that no builder box could ever execute (RCCL over more than one GPU).  If one of them does not
    come back, the rank still prints the line it has — with the sub-leg marked as timed out — and the process ends, instead of the job
    hanging without a line."""

    def __init__(self, seconds, on_timeout):
        import threading
        self.t = threading.Timer(seconds, on_timeout)
        self.t.daemon = True
        self.t.start()

    def cancel(self):
        self.t.cancel()


def allreduce_min_flag(world, ok):
    if world == 1:
        return bool(ok)
    import torch
    import torch.distributed as dist
    t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() > 0.5)


def gather_floats(world, rank, x):
    if world == 1:
        return [float(x)]
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(x) if r == rank else 0.0 for r in range(world)], dtype=torch.float64)
    dist.all_reduce(t)
    return [float(v) for v in t]


def replicas_x1024_leg(vp, pws, golden, a, world, rank, local):
    """BASELINE.json configs[3]: SHA-256 1024-block circuit, one independent proof per GPU (witness seed 1 + rank), no data-path collective.
    Rank 0's transcript against the real reference's golden (seed 1), rank 1's against the oracle's seed-2 fixture; every rank's proof through
    the full verifier replay (device predicates)."""
    B = 1024
    g = golden["sha256_x%d" % B]
    t_b = time.perf_counter()
    circ = vp.Circuit.from_pws(pws, B, seed=1 + rank)
    build_sec = time.perf_counter() - t_b
    sess = vp.Session(circ, device=local)
    sess.draw_tape()
    steps, warmup = max(3, a.steps // 4), 2
    tr, res, elapsed, dev_ms = gkr_leg(vp, circ, sess, steps, warmup, world, False, local)
    elapsed_max, proofs = aggregate(world, elapsed, float(steps))
    ok, _ = sess.check(tr, device_predicates=True)
    exact = None
    if rank == 0:
        exact = tr == open(os.path.join(ROOT, "tests", "golden", g["transcript"]), "rb").read()[g["gkr_slice"][0]:g["gkr_slice"][1]]
    elif rank == 1:
        fx = os.path.join(ROOT, "tests", "golden", "oracle_sha256_x1024_gkr_seed2.bin")
        exact = (tr == open(fx, "rb").read()) if os.path.exists(fx) else None
    all_ok = allreduce_min_flag(world, ok and exact is not False)
    dev = gather_floats(world, rank, dev_ms / steps)
    sess.close(); circ.close()
    ref_ops = g["mult_counter"] + g["add_counter"]
    return {"config": {"workload": "SHA-256 1024-block circuit x %d independent proofs, one per GPU, witness seeds 1..%d (BASELINE configs[3]); GKR sumcheck on GPU"
                                   % (world, world), "field_ops_per_proof": ref_ops, "proofs_per_step": world},
            "value": ref_ops * proofs / elapsed_max, "unit": "field-ops/s", "scaling": "weak", "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * elapsed_max / steps, "prover_sec_device_per_rank": [1e-3 * x for x in dev], "rounds": res["rounds"],
            "rank0_bit_exact_vs_reference": exact if rank == 0 else None, "every_rank_verified_and_matching_its_fixture": all_ok,
            "circuit_build_sec_rank0": build_sec}


def sharded_leg(vp, pws, golden, a, world, rank, local, blocks):
    """One proof over the GPUs of the node (north_star: "independent sumcheck instances / FFT subtrees shard across the GPUs with a single RCCL
    reduce"): the chains of ONE proof dealt to the ranks (vp_set_shard; long chains also cut by index when the in-library communicator is
    there), ONE all-reduce of the transcript per proof; then the commitment of the same instance sharded over the ranks (vp_pc_set_shard:
    slices -> one all-to-all per oracle -> positions, one all-gather of tree nodes).  Every rank holds the same instance (seed 1).  With one
    GPU per rank the collectives are RCCL calls inside the C ABI; in a rehearsal with more ranks than GPUs (VP_BENCH_BACKEND=gloo) the
    transcript goes through torch/gloo and the commitment's collectives through the host transport (vp_shard_exchange_get / _put)."""
    import numpy as np
    import torch.distributed as dist
    name = "sha256_x%d" % blocks
    g = golden[name]
    rccl = "nccl" in dist.get_backend()
    circ = vp.Circuit.from_pws(pws, blocks, seed=1)
    sess = vp.Session(circ, device=local)
    sess.draw_tape()
    full, okf = sess.prove_full(batched=True)                     # unsharded, on every rank: the answer the shards must assemble to
    gold_full = open(os.path.join(ROOT, "tests", "golden", g["transcript"]), "rb").read()
    gold = gold_full[g["gkr_slice"][0]:g["gkr_slice"][1]]
    point = sess.last_point()
    inputs = sess.layer_values(0)
    pub = sess.eq_table(point)
    n_bits = circ.layer_bitlen(0)
    sess.set_shard(rank, world)
    mode = "gloo"
    rccl_ranks = None
    if rccl:
        sess.attach_comm(rank, world)
        rccl_ranks = sess.comm_count()
        mode = "rccl"
        if world < 16:                                              # vp_set_shard_split: at most 8 slices
            sess.set_shard_split(11)
    steps, warmup = max(3, a.steps // 2), 2
    tr, res, elapsed, dev_ms = gkr_leg(vp, circ, sess, steps, warmup, world, mode, local)
    elapsed_max, _ = aggregate(world, elapsed, 0.0)
    owner, cost = sess.shard_chains()
    dev = gather_floats(world, rank, dev_ms / steps)
    ref_ops = g["mult_counter"] + g["add_counter"]
    out = {"config": {"workload": "ONE SHA-256 %d-block proof over %d ranks: sumcheck chains dealt to the ranks%s, one all-reduce of the transcript per proof"
                                  % (blocks, world, " (long chains cut by table index)" if rccl else ""), "field_ops_per_proof": ref_ops},
           "scaling": "strong", "transport": ("RCCL inside the C ABI (vp_comm_init), ncclCommCount = %s" % rccl_ranks) if rccl else "torch.distributed/gloo on host buffers (REHEARSAL: more ranks than GPUs)",
           "rccl_ranks": rccl_ranks, "steps": steps, "value": ref_ops * steps / elapsed_max, "unit": "field-ops/s", "ms_per_step": 1e3 * elapsed_max / steps,
           "device_ms_per_rank": dev, "chains": int((cost > 0).sum()), "chains_split_by_index": int((owner == -1).sum()),
           "assembled_transcript_bit_exact_vs_reference": tr == gold, "assembled_equals_unsharded": tr == full[32:32 + len(gold)]}
    sess.close(); circ.close()
    # ---- the commitment of the same instance, sharded
    if (world & (world - 1)) == 0 and (1 << (n_bits - 6)) >= 2 * world:
        st = n_bits - 6
        fri = open(os.path.join(ROOT, "tests", "golden", g["fri"]), "rb").read()
        rr = np.frombuffer(b"".join(fri[48 * k:48 * k + 16] for k in range(st)), dtype=np.uint64).reshape(st, 2).copy()
        roots_gold = b"".join(fri[48 * k + 16:48 * k + 48] for k in range(st))
        sc = vp.ShardedCommitmentRank(inputs, n_bits, rank, world, device=local, transport="rccl" if rccl else "host")
        barrier(world)
        t0 = time.perf_counter()
        root_l = sc.commit_private(); ms_priv = sc.device_ms()
        root_h, inner, all_sum = sc.commit_public(pub); ms_pub = sc.device_ms()
        roots, fin = sc.fri_commit(rr); ms_fri = sc.device_ms()
        gpu_sync(local)
        barrier(world)
        wall = time.perf_counter() - t0
        tail = gold_full[len(gold_full) - (32 + 16 + 65 * 16):]
        ok_pc = (root_l == gold_full[:32] and root_h + inner + all_sum == tail and roots == roots_gold
                 and fin.tobytes() == fri[48 * st:48 * st + 2048 * 16])
        out["commitment"] = {"workload": "commit_private + commit_public + FRI commit phase of the same input layer (2^%d wires), 64 / %d slices per rank" % (n_bits, world),
                             "transport": "RCCL (all-to-all as grouped send/recv, all-gather) inside the entry points" if rccl else "host transport over gloo (rehearsal)",
                             "rccl_ranks": sc.comm_count() if rccl else None, "wall_sec": wall,
                             "device_ms_per_rank": {"commit_private": gather_floats(world, rank, ms_priv), "commit_public": gather_floats(world, rank, ms_pub),
                                                    "fri_commit": gather_floats(world, rank, ms_fri)},
                             "roots_input0_allsum_fri_bit_exact_vs_reference_on_every_rank": allreduce_min_flag(world, ok_pc)}
        sc.close()
    else:
        out["commitment"] = {"skipped": "vp_pc_set_shard needs a power-of-two world with at least two positions per rank"}
    return out


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
    ap.add_argument("--per-launch", action="store_true", help="include the per-launch table of the headline leg (and every interactive round) in the JSON line (the per-kernel table is always there)")
    ap.add_argument("--no-two-in-flight", action="store_true", help="N = 1: skip the `two_in_flight` sub-leg (two sessions of the circuit, two host threads)")
    ap.add_argument("--no-sharded-leg", action="store_true", help="N > 1: skip the `sharded` sub-leg (one proof + its commitment over all ranks, RCCL)")
    ap.add_argument("--subleg-timeout", type=float, default=900.0, help="N > 1: seconds the multi-rank sub-legs may take before the line is printed without them")
    a = ap.parse_args()

    # ---- one process per GPU, always.  N > 1 without a launcher: start the ranks (before anything of this process touches the GPU).
    env_world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if a.gpus < 1:
        sys.stderr.write("bench.py: --gpus must be >= 1\n"); sys.exit(2)
    if env_world == 0 and a.gpus > 1:
        sys.exit(spawn_ranks(a.gpus))
    if env_world not in (0, a.gpus):
        sys.stderr.write("bench.py: launched with WORLD_SIZE=%d but --gpus %d: refusing to report a rank count that is not the one running\n" % (env_world, a.gpus))
        sys.exit(2)

    # The native libraries are loaded BEFORE torch so that every rank count uses the same HIP runtime load order
    # (libvpgpu.so first; torch then reuses the already loaded libamdhip64).  Only local rank 0 may (re)build.
    import vp_loader
    vp = vp_loader.load()
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        vp.build()
        vp.lib_host()
    with stdout_to_stderr():
        world, rank, local = dist_setup(a.gpus, rccl=True)
        barrier(world)
    local = device_of(local)
    shard = a.shard_chains and world > 1
    seed = 1 if shard else 1 + rank          # a sharded proof: every rank holds the same instance
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

        pipelined = None
        if rank == 0 and world == 1 and not shard and not a.no_two_in_flight:
            g_ = golden.get(gname)
            pipelined = two_in_flight_leg(vp, circ, sess, tr, a.steps, a.warmup, local, (g_["mult_counter"] + g_["add_counter"]) if g_ else None)

        interactive = None
        if rank == 0:
            # the drop-in path of the reference's own call pattern (one vp_round per verifier message), outside the timed region
            t_i = time.perf_counter()
            tr_i, res_i, ok_i = sess.prove_interactive()
            interactive = {"prover_sec": res_i["prove_sec"], "init_calls_sec": res_i.get("init_sec"), "round_calls_sec": res_i.get("round_sec"),
                           "finalize_calls_sec": res_i.get("finalize_sec"), "wall_sec_with_host_verifier": time.perf_counter() - t_i,
                           "transcript_equals_batched": tr_i == tr, "verified": ok_i,
                           "per_round": per_round_summary(sess.round_stats(), full=a.per_launch),
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
                "two_in_flight": pipelined,
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
            line["rccl_ranks"] = None
        # ---- N > 1: BASELINE configs[3] (x1024, one proof per GPU) and one proof + commitment sharded over all ranks, in the SAME line.
        # Every rank takes part; a watchdog prints the line without them if a collective does not come back.
        if world > 1 and not shard and a.blocks == 64 and not a.randomize:
            state = {"leg": None}

            def on_timeout():
                if rank == 0:
                    line["multi_gpu_sublegs_error"] = "timed out after %.0f s in %s" % (a.subleg_timeout, state["leg"])
                    print(json.dumps(line), flush=True)
                os._exit(0)

            wd = Watchdog(a.subleg_timeout, on_timeout)
            sub = {}
            for name, enabled, fn in (("x1024_replicas", not a.no_x1024_leg and "sha256_x1024" in golden,
                                       lambda: replicas_x1024_leg(vp, pws, golden, a, world, rank, local)),
                                      ("sharded", not a.no_sharded_leg,
                                       lambda: sharded_leg(vp, pws, golden, a, world, rank, local, 64 if a.no_x1024_leg else 1024))):
                if not enabled:
                    continue
                state["leg"] = name
                failed = None
                try:
                    sub[name] = fn()
                except Exception as e:          # a rank that fails outside a collective: the others find out at the next flag exchange
                    failed = "%s: %s" % (type(e).__name__, e)
                    sub[name] = {"error": failed}
                if not allreduce_min_flag(world, failed is None):
                    sub.setdefault(name, {})
                    if "error" not in sub[name]:
                        sub[name] = {"error": "another rank failed in this leg", "partial": sub[name]}
            wd.cancel()
            if rank == 0:
                line.update(sub)
                if isinstance(sub.get("sharded"), dict):
                    line["rccl_ranks"] = sub["sharded"].get("rccl_ranks")
        if rank == 0:
            print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
