#!/usr/bin/env python3
"""bench.py — headline benchmark: prover seconds and field-ops/s of the SHA-256 1024-block circuit with the Virgo commitment on ONE MI355X
(BASELINE.json configs[2], the largest single-GPU configuration; N > 1: configs[3], one independent instance per GPU).

A step = the prover side of the COMPLETE protocol in one pass (vph_prove_protocol): commit_private -> GKR proof (batched: Vres + three sumchecks
per layer, 859 rounds, from the pre-drawn verifier tape) -> commit_public on eq(r_liu, .) built on the device -> fft_gkr -> FRI commit phase and
final codeword — every prover call of the reference's verifier::verify() except answering the queries, and nothing of the verifier.  The
circuit, the witness and every verifier draw are resident / known before the timed region; no field data crosses PCIe inside it except the
transcript (45 KB) coming back.  Rank r proves its own instance (witness drawn after srandom(1 + r)), no data-path collective: "scaling": "weak".
Rank 0's transcript, FRI roots and final codeword are compared byte for byte with the REAL reference's recorded run on every run.

stdout carries ONE compact JSON line (<= 4 KB, strict JSON: compact_line): the contract fields, `roofline` (the kernel with the largest share of
the step: k_leaf_hash, Keccak-f/s against its instruction-count floor, HIP events around every launch on the library stream), `cpu_baseline`
(the real reference binary on a bounded sample: the same protocol at x64 on one host core), one-number summaries of the other legs; the tables —
per kernel, per launch, per interactive round, the nested x64 GKR-only leg (BASELINE configs[1]), the oracle port's full-size GKR proof on one
core, the multi-rank `sharded` sub-leg — go to the detail file named in the line (gpurun_out/bench_detail_n<N>.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--blocks B] [--no-pc] [--no-cpu-baseline]
    python bench.py --blocks 64 --no-pc              the round-1..3 headline (BASELINE configs[1]: GKR only, commitment off)
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
# PMC summaries of this command (tools/gpu_profile.sh -> tools/pmc_summary.py: separate rocprofv3 --pmc passes for FETCH_SIZE, WRITE_SIZE and the SQ
# counters, FETCH_SIZE doubled for gfx950 as the guide's HBM section prescribes): profiles/r<NN>_*pmc_summary_<tag>.json, the NEWEST round's file wins.
# tag = b<blocks> for the SHA-256 circuits, randomize_<layers>_<log> for BASELINE configs[4].


def pmc_files(tag):
    import glob
    fs = glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*pmc_summary_%s.json" % tag)) + glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_summary_%s.json" % tag))
    return sorted(set(fs), key=lambda f: os.path.basename(f), reverse=True)


def issue_share(tag):
    """tools/isa_cycles.py's record for the workload (newest round first): per kernel, the share of a launch the SIMDs spend issuing the kernel's own VALU stream
    (static mix x measured per-class issue cost x SQ_INSTS_VALU, over the launch's single-stream duration).  ({}, None) when there is none."""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_issue_share_%s.json" % tag)), key=os.path.basename, reverse=True)
    for f in fs:
        try:
            return json.load(open(f)).get("kernels", {}), os.path.relpath(f, ROOT)
        except (OSError, ValueError):
            continue
    return {}, None


def pmc_tag(blocks, randomize=None):
    return ("randomize_%d_%d" % tuple(randomize)) if randomize else ("b%d" % blocks)


# Issue-rate ceilings of the two compute-bound kernel families of the commitment, measured with tools/micro_rates.hip on MI355X
# (profiles/r02_micro_rates.txt; 256 CUs x 4 SIMDs at 2.4 GHz, 8 waves per SIMD):
#   F_p^2 multiply (31-bit split form, 16 v_mad_u64_u32 + Mersenne folds): 6.1e11 per second for the whole chip;
#   Keccak-f[1600] as 24 rounds x 180 VALU instructions (v_bitop3_b32 / v_alignbit_b32) at the measured issue cost of those two.
FMUL_PEAK_PER_S = 7.0e11        # f_mul: 224.8 SIMD-cycles per wave-multiply
# Keccak-f[1600] on 32-bit lanes: 180 VALU instructions per round is the instruction-count FLOOR (theta parity 20 three-input xors, 10 rotations by 1,
# 50 theta-apply, rho 48 rotation halves + the 10 above = 58 v_alignbit_b32, chi 50 v_bitop3_b32, iota 2) — 120 v_bitop3_b32 + 58 v_alignbit_b32 + 2 v_xor_b32.
# PEAK (round 6): every one of the 180 at the guide's uniform issue floor of 2 cycles per wave64 instruction (MI355X_MICROARCH.md: VALU, 1024 SIMDs x 64 lanes x
# 2.4 GHz) = 360 cycles per wave-round — a bound no kernel can beat per cycle.  (Rounds 4-5 priced the 58 rotation halves at their MEASURED 4 cycles: 476 cycles
# per wave-round, which the kernel's in-kernel clocks undercut at 435.6 — a "peak" that is not a floor; rounds 2-3 priced all 180 at 2.78.  Both ratios stay in
# the object under their own names.)  The kernel's distance from the floor is three measured factors, printed with it (roofline.factors):
#   issue : 360 / cycles per wave-round the kernel really takes (s_memtime stamps of the -DVP_LEAF_STAMPS flavour: LEAF_WG_KCYCLES per 1024-thread workgroup
#           = 4 waves per SIMD x 65 permutations x 24 rounds; the rotation halves are half-rate instructions on gfx950, profiles/r04_micro_keccak_instruction_classes.txt);
#   clock : the shader clock under this load / 2.4 GHz (same stamps: d(memtime) / d(memrealtime));
#   tail  : 8 rounds of workgroups x one workgroup's time / the launch's time (clock ramp at the start of a launch, the last round's stragglers).
KECCAK_INSTR_PER_ROUND = 180
KECCAK_FLOOR_CYCLES_PER_ROUND = 180 * 2.0
KECCAK_PEAK_PER_S = 1024 * 64 * 2.4e9 / (24 * KECCAK_FLOOR_CYCLES_PER_ROUND)
KECCAK_PEAK_PER_S_R05 = 1024 * 64 * 2.4e9 / (24 * (122 * 2.0 + 58 * 4.0))
KECCAK_PEAK_PER_S_R03 = 1024 * 64 * 2.4e9 / (24 * 180 * 2.78)
# in-kernel stamps of the shipped leaf-hash kernel (tools/leaf_in_step.py with the -DVP_LEAF_STAMPS flavour); tests/test_bench_line.py checks the file says the same
LEAF_STAMPS_FILE = "profiles/r06_leaf_hash_in_step.txt"
LEAF_WG_KCYCLES = 2718.0            # shader cycles of one 1024-thread workgroup (16 waves, 4 per SIMD, 65 chained permutations each)
LEAF_CYCLES_PER_WAVE_ROUND = LEAF_WG_KCYCLES * 1e3 / (4 * 65 * 24)


def pmc_kernel(tag, kernel):
    """The newest committed PMC summary's record of `kernel` for this workload -> (record, file name), or (None, None)."""
    for f in pmc_files(tag):
        try:
            for k in json.load(open(f))["kernels"]:
                if k["kernel"].replace("vp::", "") == kernel and "hbm_bytes_per_launch" in k:
                    return k, "profiles/" + os.path.basename(f)
        except Exception:
            continue
    return None, None


def pmc_traffic(tag, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary of this command at this size, or (None, None)."""
    k, src = pmc_kernel(tag, kernel)
    return (k["hbm_bytes_per_launch"], src) if k else (None, None)


def pmc_limiter(tag, kernel, avg_launch_us):
    """`measured_limiter` of a VALU-bound kernel, derived from the counters of the file it names (nothing quoted from an older round)."""
    k, src = pmc_kernel(tag, kernel)
    if not k or "SQ_INSTS_VALU_per_launch" not in k or not k.get("SQ_WAVE_CYCLES_per_launch"):
        return None
    insts, waitf = k["SQ_INSTS_VALU_per_launch"], k.get("SQ_WAIT_ANY_per_launch", 0.0) / k["SQ_WAVE_CYCLES_per_launch"]
    s = "%s: %.3g VALU wave-instructions per launch (SQ_INSTS_VALU) = %.3g per SIMD; SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.2f" % (src, insts, insts / 1024, waitf)
    if k.get("single_stream_avg_launch_us"):
        cyc = k["single_stream_avg_launch_us"] * 1e-6 * 2.4e9
        s += "; at the profile's own %.0f us per launch (single stream) one wave-instruction per %.1f cycles (2.4 GHz) per SIMD, against the guide's 2-cycle issue floor" % (k["single_stream_avg_launch_us"], cyc / (insts / 1024))
    s += "; HBM traffic per launch %.3g B (counters) — the bytes moved are about the algorithmic bytes, the integer multiply-add of F_p^2 (16 v_mad_u64_u32 + Mersenne folds) sets the time" % k["hbm_bytes_per_launch"]
    return s


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
# of a sumcheck, afterwards b comes from the previous claim (HISTORY.md section 4, "five products per pair").  4 without an add table (Liu phase).
FMUL_PER_PAIR_STEP = 5


def roofline_of(rows, tag, serial_ms, note=None):
    """The `roofline` object for the kernel with the largest share of the (single-stream) proof time.  The fold family is bound by VALU issue,
    not by HBM (PMC: traffic = algorithmic bytes, VALU busy ~80 %): its `achieved` is F_p^2-multiply-equivalents per second against the
    chip's measured F-multiply issue rate; the HBM figure of the same launches is kept beside it (hbm_frac)."""
    if not rows:
        return None
    d = rows[0]
    traffic, src = pmc_traffic(tag, d["kernel"])
    fold = "sumfold" in d["kernel"]
    limiter = pmc_limiter(tag, d["kernel"], d["avg_launch_us"]) if (fold or "light" in d["kernel"]) else None
    accounting = None
    out = {"kernel": d["kernel"], "kernel_time_share": d["time_share"], "launches": d["launches"], "avg_launch_us": d["avg_launch_us"],
           "algorithmic_bytes_per_launch": d["algorithmic_MB_per_launch"] * 1e6, "single_stream_proof_ms": serial_ms,
           "traffic": traffic, "traffic_source": ("%s (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, per launch)" % src) if traffic else None,
           "hbm_GBps": d["GBps"], "hbm_peak_GBps": HBM_PEAK_GBPS, "hbm_frac": d["hbm_frac"], "measured_limiter": limiter,
           "how": "every launch of the plan bracketed with HIP events in a single-stream replay of the same proof (vp_set_profiling / vp_get_launch_stats)" + (("; " + note) if note else "")}
    kr, ksrc = pmc_kernel(tag, d["kernel"])
    if kr and kr.get("valu_issue_floor_frac"):
        # the guide's VALU issue floor (2 cycles per wave64 instruction, 1024 SIMDs, 2.4 GHz) for this kernel's launch mix: instructions from the PMC pass and the
        # duration from the single-stream kernel trace of the SAME profiled command (tools/pmc_summary.py --add-floor) — one plan on both sides, whatever plan
        # this run's tuner picked
        out["valu_issue_floor"] = {"frac": kr["valu_issue_floor_frac"], "valu_wave_instructions_per_launch": kr["SQ_INSTS_VALU_per_launch"],
                                   "avg_launch_us_in_the_profile": kr.get("single_stream_avg_launch_us"),
                                   "wait_share_of_wave_cycles": (kr.get("SQ_WAIT_ANY_per_launch", 0.0) / kr["SQ_WAVE_CYCLES_per_launch"]) if kr.get("SQ_WAVE_CYCLES_per_launch") else None,
                                   "source": ksrc}
    shares, ssrc = issue_share(tag)
    if d["kernel"] in shares:
        out["own_stream_issue"] = dict(shares[d["kernel"]], source=ssrc)
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


def profile_gkr(sess, tr_expected, tag):
    """Roofline pass (outside the timed region, same process, same resident state): the proof is replayed on ONE stream with HIP
    events around EVERY launch (in the timed steps the independent sumchecks overlap on several streams, which makes per-kernel
    event times meaningless there).  Returns (per-kernel rows, per-launch rows, roofline object, single-stream device ms)."""
    sess.set_profiling(1)
    tr_p, res_p = sess.prove_gkr()
    stats = sess.launch_stats()
    sess.set_profiling(0)
    assert tr_p == tr_expected, "profiled replay produced a different transcript"
    rows, per_launch, _ = launch_table(stats)
    return rows, per_launch, roofline_of(rows, tag, res_p["gkr_device_ms"]), res_p


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
                             "frac_of_round3_peak_definition": w / (us * 1e-6) / KECCAK_PEAK_PER_S_R03,
                             "peak_definition": "24 rounds x (122 logic instructions x 2 cycles + 58 rotation halves x 4 cycles): the 32-bit instruction minimum of Keccak-f[1600] (120 v_bitop3_b32 + 2 v_xor_b32; 58 v_alignbit_b32) at the two issue rates of a gfx950 SIMD "
                               "(full-rate class 2 cycles per wave-instruction, half-rate class 4: tools/micro_keccak_parts.py), 1024 SIMDs x 64 lanes, 2.4 GHz; 65 chained permutations per leaf: integer-ALU-bound "
                               "(SURVEY 8d: report hashes/s, not GB/s).  frac_of_round3_peak_definition keeps rounds 2-3's bound (all 180 instructions at 2.78 cycles), which round 4's kernel exceeds in isolation"}
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


class TermGuard:
    """A launcher that sees one rank die sends SIGTERM to the others (torch.distributed.run does).  Rank 0 may sit inside a collective of
    the C library at that moment, where a Python-level handler would never run: the signal's C handler writes to a wake-up pipe instead, and a
    helper thread that reads the pipe prints the line this rank has (with the sub-leg marked) and ends the process with code 3."""

    def __init__(self, on_term):
        import signal
        import threading
        self.r, self.w = os.pipe()
        os.set_blocking(self.w, False)
        self.old_fd = signal.set_wakeup_fd(self.w, warn_on_full_buffer=False)
        self.old_handler = signal.signal(signal.SIGTERM, lambda *_: None)
        self.armed = True

        def waiter():
            while True:
                b = os.read(self.r, 1)
                if not b or not self.armed:
                    return
                if b[0] == int(signal.SIGTERM):
                    on_term()

        threading.Thread(target=waiter, daemon=True).start()

    def disarm(self):
        import signal
        self.armed = False
        signal.set_wakeup_fd(self.old_fd)
        signal.signal(signal.SIGTERM, self.old_handler if self.old_handler is not None else signal.SIG_DFL)
        try:
            os.write(self.w, b"\0")
        except OSError:
            pass


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


def sharded_prepare(vp, pws, golden, a, world, rank, local, blocks):
    """The LOCAL half of the `sharded` sub-leg — instance, session, the unsharded proof the shards must assemble to; no collective in here, so a
    rank that fails here can still tell the others at the flag exchange that follows (main)."""
    name = "sha256_x%d" % blocks
    g = golden[name]
    # test hook (tests/test_gpu_parity.py: a rank that fails / never arrives before the first collective): "raise:<rank>" | "hang:<rank>" | "die:<rank>" (killed: the launcher then ends the other ranks with SIGTERM)
    inj = os.environ.get("VP_BENCH_INJECT", "")
    if inj and int(inj.split(":")[1]) == rank:
        if inj.startswith("raise"):
            raise RuntimeError("injected failure on rank %d before the first collective" % rank)
        if inj.startswith("die"):
            os.kill(os.getpid(), 9)
        time.sleep(10 ** 6)
    circ = vp.Circuit.from_pws(pws, blocks, seed=1)
    sess = vp.Session(circ, device=local)
    sess.draw_tape()
    full, okf = sess.prove_full(batched=True)                     # unsharded, on every rank: the answer the shards must assemble to
    point = sess.last_point()
    return {"name": name, "g": g, "circ": circ, "sess": sess, "full": full, "point": point, "inputs": sess.layer_values(0), "pub": sess.eq_table(point),
            "n_bits": circ.layer_bitlen(0)}


def sharded_leg(vp, prep, golden, a, world, rank, local, blocks):
    """One proof over the GPUs of the node (north_star: "independent sumcheck instances / FFT subtrees shard across the GPUs with a single RCCL
    reduce"): the chains of ONE proof dealt to the ranks (vp_set_shard; long chains also cut by index when the in-library communicator is
    there), ONE all-reduce of the transcript per proof; then the commitment of the same instance sharded over the ranks (vp_pc_set_shard:
    slices -> one all-to-all per oracle -> positions, one all-gather of tree nodes).  Every rank holds the same instance (seed 1).  With one
    GPU per rank the collectives are RCCL calls inside the C ABI; in a rehearsal with more ranks than GPUs (VP_BENCH_BACKEND=gloo) the
    transcript goes through torch/gloo and the commitment's collectives through the host transport (vp_shard_exchange_get / _put)."""
    import numpy as np
    import torch.distributed as dist
    g, circ, sess, full, point, inputs, pub, n_bits = (prep[k] for k in ("g", "circ", "sess", "full", "point", "inputs", "pub", "n_bits"))
    rccl = "nccl" in dist.get_backend()
    gold_full = open(os.path.join(ROOT, "tests", "golden", g["transcript"]), "rb").read()
    gold = gold_full[g["gkr_slice"][0]:g["gkr_slice"][1]]
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


def gkr_workload(vp, a, pws, golden, world, rank, local, blocks, shard_req, cpu_base=True, nested=False):
    """GKR-only workload (BASELINE configs[1] at blocks = 64, configs[4] with --randomize): a step = one batched GKR proof, commitment off.
    Returns the DETAIL dict of the leg on rank 0 (None elsewhere); every rank takes part in the timed region."""
    shard = shard_req and world > 1
    seed = 1 if shard else 1 + rank          # a sharded proof: every rank holds the same instance
    gname = "sha256_x%d" % blocks if not a.randomize else "randomize_%d_%d" % tuple(a.randomize)
    if a.randomize:
        circ = vp.Circuit.randomize(a.randomize[0], a.randomize[1], seed=seed)
    else:
        circ = vp.Circuit.from_pws(pws, blocks, seed=seed)
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
    t_f = time.perf_counter()
    sess.prove_gkr()                               # the first proof of a circuit: launch plan recorded, tuner, graph capture
    first_proof_sec = time.perf_counter() - t_f
    steps, warmup = (a.steps, a.warmup) if not nested else (max(5, a.steps), max(2, a.warmup))
    tr, res, elapsed, dev_ms = gkr_leg(vp, circ, sess, steps, warmup, world, shard, local)
    elapsed, proofs = aggregate(world, elapsed, float(steps))
    shard_info = None
    if shard:
        proofs = float(steps)                      # all ranks worked on the same proof
        import torch, torch.distributed as dist
        dm = torch.tensor([dev_ms / steps if r == rank else 0.0 for r in range(world)], dtype=torch.float64)
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
    if a.batched_only:            # profiling runs: the timed batched proofs and nothing else on the device (no profiled replay, no interactive run, no second session)
        rows, per_launch, roof, res_p = [], [], None, None
    else:
        rows, per_launch, roof, res_p = profile_gkr(sess, tr, pmc_tag(blocks, a.randomize))

    shard_sim = None
    if a.shard_sim > 1 and world == 1 and not nested:
        # per-rank compute of a chain-sharded proof on W GPUs, measured shard by shard on this one GPU (no collective here)
        per = []
        parts = []
        import numpy as np
        vu_sums = None
        if a.shard_split and not a.no_vu_exchange:
            # index split: V_u of the split phase-2 chains from the ranks' partial inner products — the exchange a W-rank run does with one small
            # all-reduce ahead of the graph, done here by hand (vp_shard_vu_partials -> u64 sum -> vp_shard_vu_set)
            vus = []
            for r in range(a.shard_sim):
                sess.set_shard(r, a.shard_sim)
                sess.set_shard_split(a.shard_split)
                vus.append(sess.shard_vu_partials())
            vu_sums = np.sum(np.stack(vus), axis=0, dtype=np.uint64)
        for r in range(a.shard_sim):
            sess.set_shard(r, a.shard_sim)
            if a.shard_split:
                sess.set_shard_split(a.shard_split)
            for _ in range(2):
                sess.prove_gkr()
            ms = []
            for _ in range(max(3, a.steps // 2)):
                if vu_sums is not None and len(vu_sums):
                    sess.shard_vu_partials()                # (its device time is part of the proof's gkr_device_ms)
                    sess.shard_vu_set(vu_sums)
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
                     "v_u_by_partial_inner_products": None if vu_sums is None else int(len(vu_sums)),
                     "host_finish_ms": finish_ms if a.shard_split else None,
                     "cost_share_per_rank": [float(cost[owner == r].sum() / cost.sum()) for r in range(a.shard_sim)],
                     "assembled_equals_unsharded": assembled == tr,
                     "note": "each shard run alone on this GPU; a W-GPU run adds one all-reduce of the transcript (and, with the index split, of the export area) per proof"}

    pipelined = None
    if rank == 0 and world == 1 and not shard and not a.no_two_in_flight and not nested and not a.batched_only:
        g_ = golden.get(gname)
        pipelined = two_in_flight_leg(vp, circ, sess, tr, a.steps, a.warmup, local, (g_["mult_counter"] + g_["add_counter"]) if g_ else None)

    interactive = None
    if rank == 0 and not a.batched_only:
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
    if a.with_pc and rank == 0 and not nested:
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
    ok, sec_d = (True, None) if a.batched_only else sess.check(tr, device_predicates=True)
    verify = None
    if rank == 0 and not a.batched_only:      # the verifier's side of the same proof (outside the timed region): O(|C|) predicate loops on host vs on the GPU
        ok_h, sec_h = sess.check(tr)
        verify = {"host_predicates_sec": sec_h, "device_predicates_sec": sec_d, "accepted": bool(ok_h and ok),
                  "note": "the verifier's O(|C|) loops on the device: wiring predicates (vp_predicates), gr of verifyLiu (vp_liu_gr), input-layer MLE (vp_layer_mle); the per-round checks stay on the host"}

    line = None
    if rank == 0:
        sec_per_proof_job = elapsed / steps                      # wall time of one step (all ranks in parallel)
        ops_total = (ref_ops or 0) * proofs
        line = {
            "metric": METRIC,
            "value": ops_total / elapsed if ref_ops else None,
            "unit": "field-ops/s",
            "n_gpus": distinct_gpus(world), "ranks": world, "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * sec_per_proof_job,
            "higher_is_better": True, "scaling": "strong" if shard else "weak", "vs_baseline": None,
            "dtype": DTYPE, "data": "synthetic",
            "config": {"workload": ("SHA-256 %d-block circuit (SHA256_64.pws x%d, %d gates, %d layers), GKR sumcheck on GPU, Virgo PC off%s"
                                    % (blocks, blocks, circ.gates, circ.layers, " (BASELINE configs[1])" if blocks == 64 else "")) if not a.randomize else
                                   ("layeredCircuit::randomize(%d, %d) (%d gates, %d layers), GKR sumcheck on GPU, Virgo PC off"
                                    % (a.randomize[0], a.randomize[1], circ.gates, circ.layers)),
                       "mode": "batched (verifier tape pre-drawn; transcript identical to the interactive run)",
                       "proofs_per_step": 1 if shard else world, "field_ops_per_proof": ref_ops},
            "prover_sec": sec_per_proof_job,
            "prover_sec_device": 1e-3 * dev_ms / steps,
            "first_proof_sec": first_proof_sec,
            "rounds": res["rounds"], "kernel_launches_per_proof": res["launches"],
            "bit_exact_vs_reference_golden": bit_exact, "host_verifier_accepts": bool(ok),
            "host_verifier_check": "full replay: per-round identities, wiring predicates + getFinalValue (device loops), Liu check, input-layer check",
            "golden_origin": (golden[gname].get("origin", "the real reference binary (oracle/_ref/ref_run)") if gname in golden else None),
            "two_in_flight": pipelined,
            "interactive_path": interactive, "circuit_upload_sec": upload_sec, "verifier": verify,
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
        if world == 1 and cpu_base and not a.no_cpu_baseline and not a.randomize:
            cb = cpu_baseline(pws, blocks, ref_ops)
            cb["host_cpu"] = cpu_model()
            cb["host_cores_visible"] = os.cpu_count()
            line["cpu_baseline"] = cb
    sess.close()
    circ.close()
    return line


METRIC = "prover sec + field-ops/sec, SHA-256 circuit, 1/2/4/8 MI355X (bit-exact)"
DTYPE = "u64 (F_p^2, p=2^61-1)"


def distinct_gpus(world):
    """Headline n_gpus = DISTINCT devices the ranks run on (a rehearsal of N ranks on one card is n_gpus 1, ranks N)."""
    import torch
    n = torch.cuda.device_count()
    return min(world, n) if n else world


def keccak_roofline(stats, tag):
    """roofline object of the leaf-hash launches (k_leaf_hash): integer-ALU-bound — Keccak-f[1600]/s against the instruction-count floor; the
    HBM side of the same launches (bytes read / time against 8 TB/s) beside it."""
    leaf = [e for e in stats if e["kernel"] == "k_leaf_hash"]
    if not leaf:
        return None
    tot = sum(e["us"] for e in stats) or 1e-9
    w, us, by = sum(e["work"] for e in leaf), sum(e["us"] for e in leaf), sum(e["bytes"] for e in leaf)
    traffic, src = pmc_traffic(tag, "k_leaf_hash")
    ach = w / (us * 1e-6)
    frac = ach / KECCAK_PEAK_PER_S
    f_issue = KECCAK_FLOOR_CYCLES_PER_ROUND / LEAF_CYCLES_PER_WAVE_ROUND
    avg_us = us / len(leaf)
    # a launch of 2048 workgroups on 256 CUs is eight rounds of one workgroup per CU: shader cycles the launch needs / cycles it got at 2.4 GHz
    units = (w / len(leaf)) / (65.0 * 1024)                       # 1024-thread workgroups per launch
    rounds_of_wgs = units / 256.0
    f_clock_tail = (rounds_of_wgs * LEAF_WG_KCYCLES * 1e3) / (avg_us * 1e-6 * 2.4e9) if avg_us > 0 else None
    return {"kernel": "k_leaf_hash", "bound": "valu", "achieved": ach, "peak": KECCAK_PEAK_PER_S, "unit": "Keccak-f[1600]/s",
            "frac": frac,
            "factors": {"issue_floor_over_measured_cycles": f_issue, "cycles_per_wave_round": {"floor": KECCAK_FLOOR_CYCLES_PER_ROUND, "measured": LEAF_CYCLES_PER_WAVE_ROUND},
                        "clock_times_tail_this_run": f_clock_tail, "product": (f_issue * f_clock_tail) if f_clock_tail else None,
                        "clock_and_tail_separately": "stamps of the flavour build: " + LEAF_STAMPS_FILE + " (effective clock / 2.4 GHz, and 8 x workgroup time / launch time)",
                        "stamps_file": LEAF_STAMPS_FILE},
            "frac_of_round5_peak_definition": ach / KECCAK_PEAK_PER_S_R05,
            "frac_of_round3_peak_definition": ach / KECCAK_PEAK_PER_S_R03,
            "launches": len(leaf), "avg_launch_us": avg_us, "algorithmic_bytes_per_launch": by / len(leaf),
            "hbm_GBps": by / (us * 1e-6) / 1e9, "hbm_frac": by / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, "hbm_peak_GBps": HBM_PEAK_GBPS,
            "traffic": traffic, "traffic_source": src, "kernel_time_share": us / tot,
            "peak_definition": "24 rounds x 180 VALU instructions (the 32-bit instruction minimum of Keccak-f[1600]: 120 v_bitop3_b32 + 2 v_xor_b32 + 58 v_alignbit_b32) x 2 cycles per wave64 instruction, "
                               "the uniform issue floor of MI355X_MICROARCH.md; 1024 SIMDs x 64 lanes, 2.4 GHz; 65 chained permutations per leaf: integer-ALU-bound (SURVEY 8d: report hashes/s, not GB/s). "
                               "factors: issue = 360 / the cycles per wave-round of the kernel's own in-kernel clocks (rotation halves issue at half rate), clock x tail = the rest, measured in this run",
            "how": "HIP events around every launch on the library stream (vp_set_profiling), commit_private + commit_public + FRI commit phase of the same session"}


def protocol_workload(vp, a, pws, golden, world, rank, local, blocks):
    """BASELINE configs[2] (N = 1) / configs[3] (N > 1, one independent instance per GPU, witness seed 1 + rank): SHA-256 x`blocks`, GKR sumcheck
    AND the Virgo commitment on the GPU.  A step = the prover side of the complete protocol, one pass (vph_prove_protocol): commit_private ->
    GKR (batched) -> commit_public on eq(r_liu, .) -> fft_gkr -> FRI commit phase; nothing of the verifier inside.  Circuit, witness and every
    verifier draw are resident / known before the timed region; the public vector is built on the device.  Returns the DETAIL dict on rank 0."""
    import numpy as np
    from conftest import GOLDEN
    gname = "sha256_x%d" % blocks
    g = golden.get(gname)
    t_b = time.perf_counter()
    circ = vp.Circuit.from_pws(pws, blocks, seed=1 + rank)
    build_sec = time.perf_counter() - t_b
    t_up = time.perf_counter()
    sess = vp.Session(circ, device=local)
    upload_sec = time.perf_counter() - t_up
    sess.draw_protocol_tape()
    t_f = time.perf_counter()
    sess.warm()                                    # vp_warm: the commitment's buffers, root tables, scratch, pinned staging — where the reference has its namespace-scope arrays
    warm_sec = time.perf_counter() - t_f
    t_f = time.perf_counter()
    sess.prove_gkr()                               # first GKR proof of the circuit: plan recorded, tuner (unless VP_PLAN_CACHE knows the shape), graph capture
    first_gkr = time.perf_counter() - t_f
    t_f = time.perf_counter()
    sess.prove_protocol()                          # first complete pass after vp_warm
    first_pass = time.perf_counter() - t_f
    # How the passes follow each other (a.pass_mode; vphost.h): "sync" (default) = every call of a pass waits for its result, as the reference's call sequence does;
    # "deferred" = the calls of a pass are queued without a host wait (vp_set_deferred) and collected at its end; "pipelined" = deferred, and each pass queues the
    # next pass's commit_private behind its own FRI folds (no idle device between two proofs; same work per step: the head of step 1 runs during the last
    # warm-up step, the head queued by the last timed step is inside the timed region).  All three give the same bytes.  Measured (round 5): at
    # x1024 the three are within 1.5 % of each other either way, box by box — the chip holds ~2.1 GHz under this load whatever the gaps — at x16 the queued forms are
    # 9 % faster; the other two are timed beside the headline's (ms_per_step_by_pass_mode).
    kw = {"pipelined": dict(queue_next=True), "deferred": dict(deferred=True), "sync": dict(deferred=False)}[a.pass_mode]
    for _ in range(max(1, a.warmup)):
        sess.prove_protocol(**kw)
    gpu_sync(local)
    barrier(world)
    acc = {}
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tr, roots, fin, sec = sess.prove_protocol(**kw)
        for k, v in sec.items():
            acc[k] = acc.get(k, 0.0) + v
    gpu_sync(local)
    barrier(world)
    elapsed = time.perf_counter() - t0
    elapsed_max, proofs = aggregate(world, elapsed, float(a.steps))
    # the other two forms, a few passes each (rank 0's own numbers, outside the headline's timed region): what the queueing is worth on this box
    modes = {a.pass_mode: 1e3 * elapsed / a.steps}
    if rank == 0 and not a.no_pass_modes:
        for mname, mkw in (("sync", dict(deferred=False)), ("deferred", dict(deferred=True)), ("pipelined", dict(queue_next=True))):
            if mname == a.pass_mode:
                continue
            for _ in range(2):
                sess.prove_protocol(**mkw)
            gpu_sync(local)
            t_m = time.perf_counter()
            for _ in range(max(4, a.steps // 3)):
                sess.prove_protocol(**mkw)
            gpu_sync(local)
            modes[mname] = 1e3 * (time.perf_counter() - t_m) / max(4, a.steps // 3)
    # ---- parity: rank 0 against the REAL reference's files; every rank through the complete protocol's host verifier
    exact = {}
    if g and rank == 0:
        gold = open(os.path.join(GOLDEN, g["transcript"]), "rb").read()
        exact["transcript"] = tr == gold
        st = g["fri_steps"]
        fri = open(os.path.join(GOLDEN, g["fri"]), "rb").read()
        exact["fri_roots"] = roots == b"".join(fri[48 * k + 16:48 * k + 48] for k in range(st))
        exact["fri_final_codeword"] = fin.tobytes() == fri[48 * st:48 * st + 2048 * 16]
    elif rank == 1 and blocks == 1024:
        fx = os.path.join(GOLDEN, "oracle_sha256_x1024_gkr_seed2.bin")
        if os.path.exists(fx):
            exact["gkr_slice_vs_oracle_seed2"] = tr[32:32 + os.path.getsize(fx)] == open(fx, "rb").read()
    if a.batched_only:
        # profiling runs (tools/gpu_profile.sh): nothing but the batched passes in this process — the line carries the contract fields and the parity of the passes
        sess.close(); circ.close()
        if rank != 0:
            return None
        ref_ops = (g["mult_counter"] + g["add_counter"]) if g else None
        per = {k: v / a.steps for k, v in acc.items()}
        return {"metric": METRIC, "value": (ref_ops * proofs / elapsed_max) if ref_ops else None, "unit": "field-ops/s", "n_gpus": distinct_gpus(world), "ranks": world,
                "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed_max / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE,
                "data": "synthetic", "config": {"workload": "SHA-256 %d-block circuit, GKR + commitment, batched passes only (--batched-only: a profiling run)" % blocks, "pass_mode": a.pass_mode},
                "prover_sec": {"step_wall": elapsed_max / a.steps, "gkr": per.get("gkr"), "commit_private": per.get("commit_private"), "commit_public": per.get("commit_public"),
                               "fft_gkr": per.get("fft_gkr"), "fri_commit": per.get("fri_commit")},
                "bit_exact": exact, "bit_exact_all_ranks": True, "roofline": None, "cpu_baseline": None,
                "first_proof_sec": {"gkr_first_call_incl_plan_tuner_and_graph_capture": first_gkr, "vp_warm_commitment_sec": warm_sec, "first_complete_pass_after_vp_warm": first_pass}}
    trf, ok_full, times = sess.prove_and_verify_full(reps=33)      # interactive GKR + commitment verification, 33 query repetitions
    exact["complete_protocol_accepted"] = bool(ok_full)
    exact["interactive_run_equals_batched"] = trf == tr
    all_ok = allreduce_min_flag(world, all(v is not False for v in exact.values()))
    steps_sec = gather_floats(world, rank, elapsed / a.steps)
    if rank != 0:
        sess.close(); circ.close()
        return None
    ref_ops = (g["mult_counter"] + g["add_counter"]) if g else None
    per = {k: v / a.steps for k, v in acc.items()}
    # ---- roofline pass (outside the timed region): HIP events around every launch of each call
    point = sess.last_point()
    _, _, rr = sess.last_fri()
    sess.draw_protocol_tape()
    sess.set_profiling(1)
    tr_g, res_g = sess.prove_gkr(); st_gkr = sess.launch_stats()
    sess.commit_private(); st_priv = sess.launch_stats()
    sess.commit_public_eq(point); st_pub = sess.launch_stats()
    sess.fri_commit(rr); st_fri = sess.launch_stats()
    sess.set_profiling(0)
    allst = st_gkr + st_priv + st_pub + st_fri
    rows, per_launch, tot_us = launch_table(allst)
    gkr_rows, gkr_per_launch, _ = launch_table(st_gkr)
    pc_rows, _, pc_us = launch_table(st_priv + st_pub + st_fri)
    roof = keccak_roofline(allst, pmc_tag(blocks))
    roof_gkr = roofline_of(gkr_rows, pmc_tag(blocks), res_g["gkr_device_ms"])
    ntt = [e for e in allst if e["kernel"] in ("k_ntt_lds", "k_ntt_split", "k_ntt8_cols", "k_ntt8_rows")]
    roof_ntt = None
    if ntt:
        w, us = sum(e["work"] for e in ntt), sum(e["us"] for e in ntt)
        roof_ntt = {"bound": "valu", "achieved": w / (us * 1e-6), "peak": FMUL_PEAK_PER_S, "unit": "F_p^2 multiplications/s", "frac": w / (us * 1e-6) / FMUL_PEAK_PER_S,
                    "total_us": us, "time_share": us / tot_us, "hbm_GBps": sum(e["bytes"] for e in ntt) / (us * 1e-6) / 1e9,
                    "hbm_frac": sum(e["bytes"] for e in ntt) / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                    "peak_definition": "chip-wide F_p^2 multiply issue rate of the 31-bit split form (tools/micro_rates.hip, f_mul)"}
        shares, ssrc = issue_share(pmc_tag(blocks))
        own = {k: v["issue_share_of_launch"] for k, v in shares.items() if "ntt8" in k}
        if own:
            # against the issue cost of the instructions the passes actually execute (not the multiplier-only peak above) they are VALU-bound
            roof_ntt["own_stream_issue_share_of_launch"] = dict(own, source=ssrc)
    # ---- the drop-in (interactive) path of the GKR part, and the verifier's side
    t_i = time.perf_counter()
    tr_i, res_i, ok_i = sess.prove_interactive()
    inter = {"prover_sec": res_i["prove_sec"], "init_calls_sec": res_i.get("init_sec"), "round_calls_sec": res_i.get("round_sec"),
             "finalize_calls_sec": res_i.get("finalize_sec"), "wall_sec_with_host_verifier": time.perf_counter() - t_i,
             "transcript_equals_batched": tr_i == tr_g, "verified": ok_i, "per_round": per_round_summary(sess.round_stats(), full=a.per_launch)}
    sess.draw_protocol_tape()
    ok_d, sec_d = sess.check(tr_g, device_predicates=True)
    detail = {
        "metric": METRIC,
        "value": (ref_ops * proofs / elapsed_max) if ref_ops else None, "unit": "field-ops/s",
        "value_definition": "the reference's field-op count of the circuit (mult + add counters of its own prover, SURVEY 8d) x proofs / wall time of the timed steps; a step is the "
                            "WHOLE prover pass of the protocol (GKR + commitment), so the commitment's time is inside the denominator although the reference's counters tick in the GKR part only; "
                            "gkr_field_ops_per_sec is the same count over the GKR part alone (the round-1..3 headline's definition)",
        "n_gpus": distinct_gpus(world), "ranks": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed_max / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": "SHA-256 %d-block circuit (SHA256_64.pws x%d, %d gates, %d layers, 2^%d input wires), GKR sumcheck + Virgo FFT/LDT commit on GPU (BASELINE configs[%d]%s)"
                               % (blocks, blocks, circ.gates, circ.layers, circ.layer_bitlen(0), 2 if world == 1 else 3,
                                  "" if world == 1 else ": %d independent proofs, one per GPU, witness seeds 1..%d" % (world, world)),
                   "mode": "one prover pass per step: commit_private -> GKR (batched, tape pre-drawn) -> commit_public(eq(r_liu,.) built on device) -> fft_gkr -> FRI commit phase; "
                           + {"pipelined": "calls queued without host waits, each pass queues the next pass's commit_private behind its FRI folds (same work per step, no idle device between proofs)",
                              "deferred": "calls queued without host waits inside a pass", "sync": "every call waits for its result"}[a.pass_mode],
                   "pass_mode": a.pass_mode,
                   "proofs_per_step": world, "field_ops_per_proof": ref_ops},
        "prover_sec": {"step_wall": elapsed_max / a.steps, "gkr": per.get("gkr"), "commit_private": per.get("commit_private"), "commit_public": per.get("commit_public"),
                       "fft_gkr": per.get("fft_gkr"), "fri_commit": per.get("fri_commit"),
                       "pc_prove_reference_definition": per.get("commit_private", 0) + per.get("commit_public", 0) + per.get("fft_gkr", 0) + per.get("fri_commit", 0),
                       "note": ("device time per call (events around each call's launches; fft_gkr: host time of its begin + end, its launches run beside the other calls)"
                                if a.pass_mode != "sync" else "host wall clock per call") + ", mean over the timed steps (rank 0); the reference prints `Prove Time` (GKR) and "
                               "`Polynomial commitment: prove time` (the other four)",
                       "ms_per_step_by_pass_mode": modes,
                       "step_wall_per_rank": steps_sec},
        "gkr_field_ops_per_sec": (ref_ops / per["gkr"]) if ref_ops and per.get("gkr") else None,
        "first_proof_sec": {"gkr_first_call_incl_plan_tuner_and_graph_capture": first_gkr, "vp_warm_commitment_sec": warm_sec, "first_complete_pass_after_vp_warm": first_pass},
        "bit_exact": exact, "bit_exact_all_ranks": all_ok,
        "golden_origin": g.get("origin") if g else None,
        "reference_prove_sec_build_container": g.get("reference_prove_sec_here") if g else None,
        "reference_pc_prove_sec_build_container": g.get("reference_pc_prove_sec_here") if g else None,
        "rounds": res_g["rounds"], "kernel_launches_per_gkr_proof": res_g["launches"],
        "verifier": {"gkr_replay_device_predicates_sec": sec_d, "gkr_replay_accepted": bool(ok_d),
                     "complete_protocol": {"accepted": bool(ok_full), "gkr_prove_sec_interactive": times["gkr_prove_sec"], "pc_prove_sec": times["pc_prove_sec"],
                                           "fft_gkr_sec": times["pc_fft_gkr_sec"], "query_answer_sec": times["pc_query_answer_sec"], "verify_sec": times["verify_sec"],
                                           "query_repetitions": 33}},
        "interactive_path": inter,
        "circuit_build_sec": build_sec, "circuit_upload_sec": upload_sec,
        "roofline": roof, "roofline_gkr_dominant": roof_gkr, "roofline_ntt": roof_ntt,
        "profiled_device_ms": {"all": tot_us * 1e-3, "gkr_single_stream": res_g["gkr_device_ms"], "commitment": pc_us * 1e-3},
        "kernels": rows, "per_launch": per_launch if a.per_launch else gkr_per_launch,
    }
    sess.close(); circ.close()
    return detail


_REF_PROCS = []          # every reference child ever started: main() ends them in a finally, whatever happens to the GPU legs


def reserve_cores(n):
    """Take the last `n` CPUs of this process's affinity set away from it (and from every thread it starts later) and return them: each reference run gets a
    core of its own that nothing of the GPU legs — host threads of the library, the plan tuner, the circuit builder — is ever scheduled on."""
    try:
        avail = sorted(os.sched_getaffinity(0))
        if len(avail) < n + 2:
            return [None] * n
        mine, theirs = avail[:-n], avail[-n:]
        os.sched_setaffinity(0, mine)
        return theirs
    except (AttributeError, OSError):
        return [None] * n


def reference_start(pws, blocks, pc, cpu=None):
    """The REAL reference (oracle/_ref/ref_run: /root/reference compiled in place) on one host core, as a background process pinned to `cpu` (a core
    reserved with reserve_cores: the bench's own threads never run there); collected at the end of the run."""
    ref_run = os.path.join(ROOT, "oracle", "_ref", "ref_run")
    if not os.path.exists(ref_run):
        return None

    def pin():
        if cpu is not None:
            try:
                os.sched_setaffinity(0, {cpu})
            except OSError:
                pass
    try:
        h = {"t0": time.perf_counter(), "blocks": blocks, "pc": pc, "cpu": cpu,
             "proc": subprocess.Popen([ref_run, "--pws", pws, "--blocks", str(blocks), "--pc", "1" if pc else "0"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                                      preexec_fn=pin)}
        _REF_PROCS.append(h["proc"])
        return h
    except Exception:
        return None


def reference_kill_all():
    for p in _REF_PROCS:
        try:
            if p.poll() is None:
                p.kill()
                p.wait(timeout=10)
        except Exception:
            pass


def reference_collect(h, timeout=900):
    """-> dict of the reference's own printed numbers, or None."""
    if not h:
        return None
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    try:
        out, _ = h["proc"].communicate(timeout=timeout)
    except Exception:
        try:
            h["proc"].kill()
        except Exception:
            pass
        return None
    wall = time.perf_counter() - h["t0"]
    ru1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    cpu_sec = (ru1.ru_utime + ru1.ru_stime) - (ru0.ru_utime + ru0.ru_stime)       # this child's own CPU time (children are reaped one at a time)
    m = re.search(r"Prove Time ([0-9.]+)", out)
    pm = re.search(r"Polynomial commitment: prove time ([0-9.]+)", out)
    c = re.search(r"mult counter (-?\d+), add counter (-?\d+)", out)
    if h["proc"].returncode != 0 or not (m and c) or (h["pc"] and not pm):
        return None
    gkr, pcs = float(m.group(1)), float(pm.group(1)) if (pm and h["pc"]) else 0.0
    ops = int(c.group(1)) + int(c.group(2))
    return {"blocks": h["blocks"], "gkr_prove_sec": gkr, "pc_prove_sec": pcs, "field_ops": ops, "process_wall_sec_incl_waiting_to_be_collected": wall,
            "process_cpu_sec": cpu_sec, "pinned_to_cpu": h.get("cpu")}


def cpu_protocol_baseline(r, blocks_headline=1024):
    """cpu_baseline of the configs[2] headline from a collected run of the REAL reference on the same protocol — GKR + Virgo commitment — on a BOUNDED sample
    (round 5: the x256 circuit, a quarter of the headline's blocks, ~16 GB, ~75 s on one core; the full x1024 run takes 63 GB and 13 minutes,
    tests/golden/golden.json holds it).  value = the reference's own field-op count / (its Prove Time + its commitment prove time), the headline's unit."""
    if not r or not r.get("pc_prove_sec"):
        return None
    gkr, pcs, ops = r["gkr_prove_sec"], r["pc_prove_sec"], r["field_ops"]
    return {"value": ops / (gkr + pcs), "unit": "field-ops/s", "cores": 1, "kind": "reference",
            "sample": "the real reference binary, SHA-256 x%d (1/%d of the headline's blocks), complete protocol (GKR + commitment, verifier::verify), single thread, "
                      "%s" % (r["blocks"], max(1, blocks_headline // r["blocks"]),
                              ("pinned to CPU %d, which the bench's own threads are kept off (sched_setaffinity), while the GPU legs run" % r["pinned_to_cpu"])
                              if r.get("pinned_to_cpu") is not None else "run beside the GPU legs on another core"),
            "prover_sec": gkr + pcs, "gkr_prove_sec": gkr, "pc_prove_sec": pcs, "field_ops": ops, "gkr_field_ops_per_sec": ops / gkr,
            "process_wall_sec": r["process_wall_sec_incl_waiting_to_be_collected"], "process_cpu_sec": r.get("process_cpu_sec"), "pinned_to_cpu": r.get("pinned_to_cpu"),
            "host_cpu": cpu_model(), "host_cores_visible": os.cpu_count()}


def reference_full_size_record():
    """The REAL reference at the HEADLINE size on the host of a box of this pool, recorded once (tools/ref_x1024_on_box.sh -> profiles/…): 5.5 minutes and 64 GB, too long
    for a leg of this script; it rides beside cpu_baseline (the x256 sample measured in THIS run) as a labelled record, never as `value`."""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_reference_x1024_on_gpu_box_host.txt")), reverse=True)
    if not fs:
        return None
    try:
        txt = open(fs[0]).read()
        g = float(re.search(r"^Prove Time ([0-9.]+)", txt, re.M).group(1)); pc = float(re.search(r"Polynomial commitment: prove time ([0-9.]+)", txt).group(1))
        ok = "TRANSCRIPT_EQUAL" in txt and "FRI_EQUAL" in txt
        return {"gkr_prove_sec": g, "pc_prove_sec": pc, "prover_sec": g + pc, "field_ops_per_sec": 8299342697 / (g + pc), "transcript_and_fri_equal_golden": ok,
                "source": "profiles/" + os.path.basename(fs[0]), "note": "recorded run on a box of this pool (same CPU model), not measured in this run"}
    except Exception:
        return None


def cpu_port_x1024(pws, tr_gpu):
    """The oracle port's GKR proof of the FULL headline circuit on one host core (the real reference needs 63 GB and 13 minutes for it:
    tests/golden/golden.json holds that run); compared byte for byte with the GPU's GKR slice."""
    import oracle_binding as ob
    t0 = time.perf_counter()
    oc = ob.Circuit.from_pws(pws, 1024, seed=1)
    t1 = time.perf_counter()
    otr, st = oc.prove_gkr()
    oc.close()
    ops = st["mult_count"] + st["add_count"]
    return {"kind": "port", "cores": 1, "gkr_prove_sec": st["prove_sec"], "field_ops": ops, "gkr_field_ops_per_sec": ops / st["prove_sec"],
            "circuit_build_sec": t1 - t0, "transcript_equals_gpu": (otr == tr_gpu) if tr_gpu is not None else None,
            "sample": "one full GKR proof of the 1024-block circuit by the oracle port (commitment off), single thread"}


def first_proof_child(blocks):
    """`bench.py --first-proof-child B`: a FRESH process with VP_PLAN_CACHE pointing at the file the parent's tuner has written — what the first proof of a
    circuit costs a process that starts with a tuning record (one plan recording + one graph capture instead of ~14 candidates), and the first complete pass
    behind vp_warm.  Prints one JSON object."""
    import vp_loader
    vp = vp_loader.load()
    vp.lib_host()
    with tempfile.TemporaryDirectory() as tmp:
        pws = unpack_pws(tmp)
        circ = vp.Circuit.from_pws(pws, blocks, seed=1)
    c0 = vp.Circuit.randomize(2, 1, seed=1); s0 = vp.Session(c0, device=0); s0.close(); c0.close()      # process first use (HIP context, code objects)
    sess = vp.Session(circ, device=0)
    sess.draw_protocol_tape()
    t = time.perf_counter(); sess.warm(); warm = time.perf_counter() - t
    t = time.perf_counter(); tr, _ = sess.prove_gkr(); first = time.perf_counter() - t
    t = time.perf_counter(); _, _, _, sec1 = sess.prove_protocol(); first_pass = time.perf_counter() - t
    t = time.perf_counter(); _, _, _, sec2 = sess.prove_protocol(); second_pass = time.perf_counter() - t
    import hashlib
    print(json.dumps({"gkr_first_call_with_plan_cache": first, "vp_warm_commitment_sec": warm, "first_complete_pass_after_vp_warm": first_pass, "second_complete_pass": second_pass,
                      "first_pass_by_call": sec1, "second_pass_by_call": sec2,
                      "plan_cache": os.environ.get("VP_PLAN_CACHE"), "transcript_sha256_16": hashlib.sha256(tr).hexdigest()[:16]}), flush=True)
    sess.close(); circ.close()


def first_proof_in_fresh_process(blocks, timeout=180):
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--first-proof-child", str(blocks)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        return json.loads(line[-1]) if (r.returncode == 0 and line) else {"error": (r.stderr or "")[-300:]}
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def _finite(x):
    """strict JSON: NaN / inf become null, numpy scalars become Python numbers."""
    import math
    if isinstance(x, dict):
        return {str(k): _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return x if math.isfinite(x) else None
    try:
        f = float(x)
        return f if math.isfinite(f) else None
    except Exception:
        return str(x)


def _r(x, nd=4):
    """round to `nd` significant digits (numbers in the headline line are for reading; the detail file keeps everything)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x == 0 or x != x or x in (float("inf"), float("-inf")):
            return x
        from math import floor, log10
        return round(x, max(0, nd - 1 - int(floor(log10(abs(x))))))
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


LINE_TARGET_BYTES = 4096
LINE_HARD_CAP_BYTES = 8192


def compact_line(d, detail_file=None):
    """The ONE stdout line from a leg's detail dict: the contract fields whole, `roofline` and `cpu_baseline` as compact objects, one-number summaries
    of everything else; the tables (per kernel, per launch, per round, sub-legs) stay in the detail file.  Strict JSON, <= LINE_TARGET_BYTES when
    it can be (optional keys are dropped from the end of `optional` until it fits), never above LINE_HARD_CAP_BYTES."""
    def pick(obj, keys):
        return {k: obj[k] for k in keys if isinstance(obj, dict) and k in obj and obj[k] is not None} if isinstance(obj, dict) else None
    line = {k: d.get(k) for k in ("metric", "value", "unit", "n_gpus", "ranks", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                  "dtype", "data")}
    cfg = d.get("config") or {}
    line["config"] = {k: cfg[k] for k in ("workload", "mode", "pass_mode", "proofs_per_step", "field_ops_per_proof") if k in cfg}
    roof = d.get("roofline")
    if isinstance(roof, dict):
        line["roofline"] = pick(roof, ("kernel", "bound", "achieved", "peak", "unit", "frac", "hbm_frac", "hbm_GBps", "traffic", "algorithmic_bytes_per_launch",
                                       "avg_launch_us", "launches", "kernel_time_share", "frac_of_round5_peak_definition"))
        line["roofline"].setdefault("traffic", None)
        if roof.get("traffic_source"):
            line["roofline"]["traffic_source"] = str(roof["traffic_source"]).split(" ")[0]
        fc = roof.get("factors")
        if isinstance(fc, dict):
            line["roofline"]["factors"] = {"issue": fc.get("issue_floor_over_measured_cycles"), "clock_x_tail": fc.get("clock_times_tail_this_run"),
                                           "cycles_per_wave_round": (fc.get("cycles_per_wave_round") or {}).get("measured"), "floor_cycles": (fc.get("cycles_per_wave_round") or {}).get("floor")}
    else:
        line["roofline"] = None
    cb = d.get("cpu_baseline")
    line["cpu_baseline"] = pick(cb, ("value", "unit", "cores", "kind", "sample", "prover_sec", "gkr_prove_sec", "pc_prove_sec", "host_cpu",
                                     "reference_over_port_ratio_x64_same_box")) if isinstance(cb, dict) else None
    if isinstance(cb, dict) and isinstance(cb.get("reference_x1024_recorded_on_a_box_of_this_pool"), dict):
        fr = cb["reference_x1024_recorded_on_a_box_of_this_pool"]
        line["cpu_baseline"]["x1024_recorded"] = {"prover_sec": fr.get("prover_sec"), "field_ops_per_sec": fr.get("field_ops_per_sec"), "source": fr.get("source")}
    line["rccl_ranks"] = d.get("rccl_ranks")
    if d.get("bench_wall_sec") is not None:
        line["bench_wall_sec"] = d["bench_wall_sec"]
    be = d.get("bit_exact")
    if isinstance(be, dict):
        line["bit_exact"] = all(v is not False for v in be.values()) and bool(d.get("bit_exact_all_ranks", True))
        line["bit_exact_checks"] = be
    else:
        line["bit_exact"] = d.get("bit_exact_vs_reference_golden")
    optional = []
    ps = d.get("prover_sec")
    if isinstance(ps, dict):
        optional.append(("prover_sec", pick(ps, ("step_wall", "gkr", "commit_private", "commit_public", "fft_gkr", "fri_commit", "pc_prove_reference_definition", "ms_per_step_by_pass_mode"))))
        optional.append(("gkr_field_ops_per_sec", d.get("gkr_field_ops_per_sec")))
    else:
        optional.append(("prover_sec", ps))
        optional.append(("prover_sec_device", d.get("prover_sec_device")))
    optional.append(("host_verifier_accepts", d.get("host_verifier_accepts", ((d.get("verifier") or {}).get("complete_protocol") or {}).get("accepted"))))
    x64 = d.get("x64_gkr")
    if isinstance(x64, dict):
        r64 = x64.get("roofline") or {}
        i64 = x64.get("interactive_path") or {}
        optional.append(("x64_gkr", {"workload": "configs[1]: SHA-256 x64, GKR only", "value": x64.get("value"), "ms_per_step": x64.get("ms_per_step"),
                                     "prover_sec_device": x64.get("prover_sec_device"), "steps": x64.get("steps"), "bit_exact": x64.get("bit_exact_vs_reference_golden"),
                                     "roofline_kernel": r64.get("kernel"), "roofline_frac": r64.get("frac"), "roofline_hbm_frac": r64.get("hbm_frac"),
                                     "roofline_traffic": r64.get("traffic"), "traffic_source": (str(r64.get("traffic_source")).split(" ")[0] if r64.get("traffic_source") else None),
                                     "interactive_prover_sec": i64.get("prover_sec"), "first_proof_sec": x64.get("first_proof_sec"),
                                     "cpu_reference_prover_sec": (x64.get("cpu_baseline") or {}).get("prover_sec"),
                                     "two_in_flight_ms_per_proof": (x64.get("two_in_flight") or {}).get("ms_per_proof")}))
    rz = d.get("randomize_16_20")
    if isinstance(rz, dict):
        rr = rz.get("roofline") or {}
        optional.append(("randomize_16_20", {"workload": "configs[4]: randomize(16,20), 2^24 gates, GKR only", "value": rz.get("value"), "ms_per_step": rz.get("ms_per_step"),
                                             "prover_sec_device": rz.get("prover_sec_device"), "steps": rz.get("steps"), "bit_exact": rz.get("bit_exact_vs_reference_golden"),
                                             "roofline_kernel": rr.get("kernel"), "roofline_frac": rr.get("frac"), "hbm_frac": rr.get("hbm_frac"),
                                             "roofline_traffic": rr.get("traffic"), "roofline_algorithmic_bytes_per_launch": rr.get("algorithmic_bytes_per_launch"),
                                             "traffic_source": (str(rr.get("traffic_source")).split(" ")[0] if rr.get("traffic_source") else None),
                                             "interactive_prover_sec": (rz.get("interactive_path") or {}).get("prover_sec")}))
    fp = d.get("first_proof_sec")
    if isinstance(fp, dict):
        fr = fp.get("fresh_process_with_plan_cache_file") or {}
        fp = {k: v for k, v in fp.items() if not isinstance(v, dict)}
        fp.update({"fresh_process_gkr_first_call_with_plan_cache": fr.get("gkr_first_call_with_plan_cache"),
                   "fresh_process_first_complete_pass_after_vp_warm": fr.get("first_complete_pass_after_vp_warm"), "fresh_process_error": fr.get("error")})
        fp = {k: v for k, v in fp.items() if v is not None}
    optional.append(("first_proof_sec", fp))
    ip = d.get("interactive_path")
    if isinstance(ip, dict):
        pr = ip.get("per_round") or {}
        optional.append(("interactive_path", {"prover_sec": ip.get("prover_sec"), "init_calls_sec": ip.get("init_calls_sec"), "round_calls_sec": ip.get("round_calls_sec"),
                                              "transcript_equals_batched": ip.get("transcript_equals_batched"), "rounds": pr.get("rounds"),
                                              "hbm_frac_overall_per_round": pr.get("hbm_frac_overall")}))
    for name in ("roofline_gkr_dominant", "roofline_ntt"):
        r = d.get(name)
        if isinstance(r, dict):
            o_ = pick(r, ("kernel", "bound", "frac", "hbm_frac", "avg_launch_us", "kernel_time_share", "time_share", "total_us"))
            if isinstance(r.get("valu_issue_floor"), dict):
                o_["valu_floor_frac"] = r["valu_issue_floor"].get("frac"); o_["wait_share"] = r["valu_issue_floor"].get("wait_share_of_wave_cycles")
            optional.append((name, o_))
    ks = d.get("kernels")
    if isinstance(ks, list):
        optional.append(("kernel_time_share", {k["kernel"]: k["time_share"] for k in ks[:8]}))
    cp = d.get("cpu_port_x1024_gkr")
    if isinstance(cp, dict):
        optional.append(("cpu_port_x1024_gkr", pick(cp, ("kind", "gkr_prove_sec", "gkr_field_ops_per_sec", "transcript_equals_gpu"))))
    sh = d.get("sharded")
    if isinstance(sh, dict):
        cm = sh.get("commitment") or {}
        optional.append(("sharded", {"workload": (sh.get("config") or {}).get("workload"), "ms_per_step": sh.get("ms_per_step"), "value": sh.get("value"), "scaling": sh.get("scaling"),
                                     "rccl_ranks": sh.get("rccl_ranks"), "transport": sh.get("transport"), "bit_exact": sh.get("assembled_transcript_bit_exact_vs_reference"),
                                     # the rehearsal's commitment moves 4.4 GB per oracle through host memory and gloo: its wall time says nothing about a device
                                     ("commitment_wall_sec" if sh.get("rccl_ranks") else "commitment_wall_sec_HOST_TRANSPORT_rehearsal_not_a_device_number"): cm.get("wall_sec"),
                                     "commitment_bit_exact": cm.get("roots_input0_allsum_fri_bit_exact_vs_reference_on_every_rank"), "error": sh.get("error")}))
    for k in ("sharded_proof", "sharded_proof_simulation"):
        if isinstance(d.get(k), dict):
            v = d[k]
            optional.append((k, pick(v, ("world", "max_device_ms", "device_ms_per_rank", "chains", "chains_split_by_index", "assembled_equals_unsharded"))))
    if d.get("multi_gpu_sublegs_error"):
        line["multi_gpu_sublegs_error"] = d["multi_gpu_sublegs_error"]
    if d.get("value_definition"):
        optional.append(("value_definition", d["value_definition"][:400]))
    line["detail_file"] = detail_file
    line = _r(_finite(line))
    opt = [(k, _r(_finite(v))) for k, v in optional if v is not None]
    def render(n):
        o = dict(line)
        for k, v in opt[:n]:
            o[k] = v
        return json.dumps(o, allow_nan=False, separators=(",", ":"))
    n = len(opt)
    s = render(n)
    while n > 0 and len(s.encode()) > LINE_TARGET_BYTES:
        n -= 1
        s = render(n)
    if len(s.encode()) > LINE_HARD_CAP_BYTES:           # cannot happen with the fields above; keep the contract fields whatever happens
        keep = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
        keep["config"] = {"workload": str(line["config"].get("workload"))[:300]}
        keep["detail_file"] = detail_file
        s = json.dumps(keep, allow_nan=False, separators=(",", ":"))
    return s


def write_detail(d, n_gpus, path=None):
    """Everything the line leaves out.  Default: gpurun_out/bench_detail_n<N>.json under the repo (merged back by gpurun), else the temp dir."""
    cands = [path] if path else [os.path.join(ROOT, "gpurun_out", "bench_detail_n%d.json" % n_gpus), os.path.join(tempfile.gettempdir(), "bench_detail_n%d.json" % n_gpus)]
    for p in cands:
        try:
            os.makedirs(os.path.dirname(p), exist_ok=True)
            with open(p, "w") as f:
                json.dump(_finite(d), f, allow_nan=False)
            return os.path.relpath(p, ROOT) if p.startswith(ROOT) else p
        except OSError:
            continue
    return None


class OneLine:
    """Exactly one JSON line on stdout, whoever gets there first (the normal end of main() or the watchdog)."""

    def __init__(self):
        import threading
        self.lock = threading.Lock()
        self.done = False

    def emit(self, detail, n_gpus, path=None):
        with self.lock:
            if self.done:
                return False
            self.done = True
            f = write_detail(detail, n_gpus, path)
            print(compact_line(detail, f), flush=True)
            return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=1024, help="SHA-256 blocks of the headline circuit (default 1024: BASELINE configs[2] / configs[3])")
    ap.add_argument("--no-pc", action="store_true", help="headline = GKR only, commitment off (--blocks 64 --no-pc is BASELINE configs[1], the round-1..3 headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-port-x1024", action="store_true", help="skip the oracle port's full-size GKR proof on one host core (~55 s) beside cpu_baseline")
    ap.add_argument("--randomize", type=int, nargs=2, metavar=("LAYERS", "LOG_SIZE"), default=None,
                    help="BASELINE configs[4] flavour: layeredCircuit::randomize(LAYERS, LOG_SIZE) instead of the SHA-256 circuit, GKR only (no CPU baseline)")
    ap.add_argument("--shard-chains", action="store_true",
                    help="strong scaling, GKR only: ONE proof per step; its independent sumcheck chains are dealt out to the ranks (vp_set_shard) and the "
                         "transcript is assembled by one RCCL all-reduce per proof (launch under torch.distributed.run)")
    ap.add_argument("--no-vu-exchange", action="store_true", help="with --shard-sim --shard-split: every rank of a split phase 2 adds up the whole layer for V_u (round 4) instead of exchanging partial inner products")
    ap.add_argument("--shard-sim", type=int, default=0, metavar="W",
                    help="single GPU, GKR only: run the W shards of a chain-sharded proof one after the other and report each shard's device time")
    ap.add_argument("--shard-split", type=int, default=0, metavar="MIN_LOG",
                    help="with --shard-sim or --shard-chains: also split tables of at least 2^(log2 W + MIN_LOG) entries by index over the ranks (vp_set_shard_split; 11 is the smallest useful value)")
    ap.add_argument("--with-pc", action="store_true", help="GKR-only headline: also time the Virgo commitment calls one by one (detail file)")
    ap.add_argument("--no-x64-leg", action="store_true", help="N = 1 default run: skip the nested x64 GKR-only leg (BASELINE configs[1])")
    ap.add_argument("--no-randomize-leg", action="store_true", help="N = 1 default run: skip the nested randomize(16, 20) GKR-only leg (BASELINE configs[4])")
    ap.add_argument("--pass-mode", choices=("pipelined", "deferred", "sync"), default="sync",
                    help="how the protocol passes of the headline are queued (see protocol_workload); all three give the same bytes")
    ap.add_argument("--no-pass-modes", action="store_true", help="do not time the other two pass modes beside the headline's")
    ap.add_argument("--cpu-sample-blocks", type=int, default=256, help="cpu_baseline: SHA-256 blocks of the circuit the REAL reference proves on one host core (256: ~16 GB, ~75 s, "
                    "started in the background at the beginning of the run; 64 = round 4's sample)")
    ap.add_argument("--per-launch", action="store_true", help="detail file: per-launch table of every call and every interactive round")
    ap.add_argument("--no-two-in-flight", action="store_true", help="GKR-only legs: skip the `two_in_flight` sub-leg (two sessions of the circuit, two host threads)")
    ap.add_argument("--no-sharded-leg", action="store_true", help="N > 1: skip the `sharded` sub-leg (one proof + its commitment over all ranks, RCCL)")
    ap.add_argument("--subleg-timeout", type=float, default=300.0, help="N > 1: seconds the multi-rank sub-leg may take before the line is printed without it and the ranks exit 3")
    ap.add_argument("--detail-file", default=None, help="where everything the line leaves out goes (default gpurun_out/bench_detail_n<N>.json)")
    ap.add_argument("--first-proof-child", type=int, default=0, metavar="BLOCKS", help="internal: the fresh-process leg of first_proof_sec (see first_proof_child)")
    ap.add_argument("--no-fresh-process-leg", action="store_true", help="N = 1 headline: skip the fresh-process first proof with the plan cache file")
    ap.add_argument("--batched-only", action="store_true", help="profiling runs: the headline's setup, warm-up and timed batched passes and NOTHING else in the process (no profiled "
                    "replay, no interactive run, no verification, no nested legs) — per-kernel counters of such a run are the headline's launches alone")
    a = ap.parse_args()
    if a.first_proof_child:
        first_proof_child(a.first_proof_child)
        return
    if a.batched_only:
        a.no_cpu_baseline = a.no_x64_leg = a.no_randomize_leg = a.no_pass_modes = a.no_fresh_process_leg = True

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
    vp.lib_host()
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
    gkr_only = a.no_pc or a.randomize or a.shard_chains or a.shard_sim
    if a.randomize:
        a.no_cpu_baseline = True
    out = OneLine()
    failed_subleg = False
    t_bench0 = time.perf_counter()
    with tempfile.TemporaryDirectory() as tmp:
        pws = unpack_pws(tmp)
        # the plan tuner's choices of this run go to a file (VP_PLAN_CACHE, read by vp_create): the fresh-process leg starts from it
        if "VP_PLAN_CACHE" not in os.environ:
            os.environ["VP_PLAN_CACHE"] = os.path.join(tmp, "plan_cache_rank%d.txt" % int(os.environ.get("RANK", "0")))
        # CPU legs exist at N = 1 only (no rank of a multi-GPU run spends a second on them); the two runs of the real reference start now, on cores of their own
        ref_big = ref_64 = None
        if a.gpus == 1 and not gkr_only and not a.no_cpu_baseline and a.blocks == 1024:
            cores = reserve_cores(2)
            ref_big = reference_start(pws, a.cpu_sample_blocks, True, cores[0])
            if not a.no_x64_leg:
                ref_64 = reference_start(pws, 64, False, cores[1])
        # one-time cost of the process (HIP context, code objects: ~0.25 s) paid by a 3-gate circuit first, so that circuit_upload_sec is
        # what a caller sees per circuit (host flatten + vp_circuit_upload with its device-side list building + vp_evaluate)
        t_up = time.perf_counter()
        c0 = vp.Circuit.randomize(2, 1, seed=1); s0 = vp.Session(c0, device=local); s0.close(); c0.close()
        first_use_sec = time.perf_counter() - t_up
        if gkr_only:
            detail = gkr_workload(vp, a, pws, golden, world, rank, local, a.blocks, a.shard_chains)
        else:
            detail = protocol_workload(vp, a, pws, golden, world, rank, local, a.blocks)
        if rank == 0:
            detail["process_first_use_sec"] = first_use_sec
            detail["rccl_ranks"] = None
            r64 = None
        if world == 1 and not gkr_only and not a.no_fresh_process_leg and isinstance(detail.get("first_proof_sec"), dict):
            detail["first_proof_sec"]["fresh_process_with_plan_cache_file"] = first_proof_in_fresh_process(a.blocks)
        # ---- N = 1 default run: BASELINE configs[1] (x64, GKR only) as a nested leg, with the real reference's GKR proof of it on one host core
        if world == 1 and not gkr_only and a.blocks == 1024 and not a.no_x64_leg:
            x64 = gkr_workload(vp, a, pws, golden, 1, 0, local, 64, False, cpu_base=False, nested=True)
            detail["x64_gkr"] = x64
        # ---- N = 1 default run: BASELINE configs[4] ("synthetic random unlayered circuit, 2^24 gates, HBM-roofline run") as a nested GKR-only leg
        if world == 1 and not gkr_only and a.blocks == 1024 and not a.no_randomize_leg:
            import copy
            a_r = copy.copy(a)
            a_r.randomize = [16, 20]
            rz = gkr_workload(vp, a_r, pws, golden, 1, 0, local, 0, False, cpu_base=False, nested=True)
            g_r = golden.get("randomize_16_20") or {}
            rz["cpu_reference_prover_sec_build_container"] = g_r.get("reference_prove_sec_here")
            detail["randomize_16_20"] = rz
        # ---- the CPU side (N = 1): the real reference's runs started at the beginning are collected here, then the port's full-size GKR proof
        if world == 1 and not gkr_only and a.blocks == 1024 and not a.no_cpu_baseline:
            r64 = reference_collect(ref_64)
            detail["cpu_baseline"] = cpu_protocol_baseline(reference_collect(ref_big))
            cb = detail.get("cpu_baseline")
            if isinstance(cb, dict):
                cb["reference_x1024_recorded_on_a_box_of_this_pool"] = reference_full_size_record()
            x64 = detail.get("x64_gkr")
            if isinstance(r64, dict) and isinstance(x64, dict):      # the reference's x64 GKR proof (PC off), same box
                x64["cpu_baseline"] = {"kind": "reference", "cores": 1, "prover_sec": r64["gkr_prove_sec"], "value": r64["field_ops"] / r64["gkr_prove_sec"], "unit": "field-ops/s"}
            if not a.no_cpu_port_x1024:
                import oracle_binding as ob
                gold = open(os.path.join(ROOT, "tests", "golden", golden["sha256_x1024"]["transcript"]), "rb").read()
                gs = golden["sha256_x1024"]["gkr_slice"]
                port = cpu_port_x1024(pws, gold[gs[0]:gs[1]] if detail["bit_exact"].get("transcript") else None)
                oc = ob.Circuit.from_pws(pws, 64, seed=1)
                _, st64 = oc.prove_gkr()
                oc.close()
                port["port_x64_gkr_prove_sec_same_box"] = st64["prove_sec"]
                if isinstance(r64, dict):
                    port["reference_over_port_ratio_x64_same_box"] = r64["gkr_prove_sec"] / st64["prove_sec"]
                    if isinstance(cb, dict):
                        cb["reference_over_port_ratio_x64_same_box"] = port["reference_over_port_ratio_x64_same_box"]
                        cb["reference_x64_gkr_prove_sec_same_box"] = r64["gkr_prove_sec"]
                detail["cpu_port_x1024_gkr"] = port
        # ---- N > 1: ONE proof + its commitment sharded over all ranks (RCCL inside the C ABI), in the SAME line.  Every rank takes part; a
        # watchdog prints the line without it and ends the ranks with a non-zero code if a collective does not come back.
        if world > 1 and not gkr_only and not a.no_sharded_leg:
            state = {"leg": "sharded"}

            def on_timeout():
                # never restart anything from here: print what there is (rank 0) and leave with an error code, so that the launcher reports the hang
                if rank == 0:
                    detail["multi_gpu_sublegs_error"] = "timed out after %.0f s in %s" % (a.subleg_timeout, state["leg"])
                    detail["bench_wall_sec"] = time.perf_counter() - t_bench0
                    out.emit(detail, world, a.detail_file)
                    sys.stdout.flush()
                else:
                    time.sleep(5.0)             # rank 0 prints first: a rank that leaves takes the launcher (and with it rank 0) down
                os._exit(3)

            wd = Watchdog(a.subleg_timeout, on_timeout)

            def on_term():
                if rank == 0:
                    detail["multi_gpu_sublegs_error"] = "terminated by the launcher in %s (a peer rank died)" % state["leg"]
                    out.emit(detail, world, a.detail_file)
                    sys.stdout.flush()
                os._exit(3)

            tg = TermGuard(on_term)
            # rank 0 alone ran the profiled / interactive / verifier legs of the headline: nobody enters the sub-leg before its detail is complete,
            # so whatever happens to a peer from here on, the line exists
            try:
                barrier(world)
            except Exception as e:
                sys.stderr.write("bench.py rank %d: barrier before the sharded sub-leg failed: %s: %s\n" % (rank, type(e).__name__, e))
                on_term()
            failed = None
            sub = prep = None
            try:                                # local work only: a failure here is reported at the flag exchange below, nobody is left in a collective
                prep = sharded_prepare(vp, pws, golden, a, world, rank, local, a.blocks)
            except Exception as e:
                failed = "%s: %s" % (type(e).__name__, e)
            try:
                all_ready = allreduce_min_flag(world, failed is None)
            except Exception as e:              # a peer went away while this rank waited for it
                sys.stderr.write("bench.py rank %d: flag exchange failed: %s: %s\n" % (rank, type(e).__name__, e))
                if rank == 0:
                    detail["multi_gpu_sublegs_error"] = "a peer rank went away before the first collective (%s)" % type(e).__name__
                    out.emit(detail, world, a.detail_file)
                    sys.stdout.flush()
                os._exit(3)
            if not all_ready:
                sub = {"error": failed or "another rank failed before the first collective of this leg"}
            else:
                try:
                    sub = sharded_leg(vp, prep, golden, a, world, rank, local, a.blocks)
                except Exception as e:
                    # inside the collective part there is nothing graceful left: the peers sit in a collective this rank will never join.  Say so
                    # (rank 0 prints the line it has) and leave with an error code; the peers' collectives fail or their watchdogs fire.
                    sys.stderr.write("bench.py rank %d: sharded sub-leg failed inside its collective part: %s: %s\n" % (rank, type(e).__name__, e))
                    if rank == 0:
                        detail["multi_gpu_sublegs_error"] = "%s: %s" % (type(e).__name__, e)
                        out.emit(detail, world, a.detail_file)
                        sys.stdout.flush()
                    os._exit(3)
            wd.cancel()
            tg.disarm()
            if rank == 0:
                detail["sharded"] = sub
                detail["rccl_ranks"] = sub.get("rccl_ranks") if isinstance(sub, dict) else None
                if isinstance(sub, dict) and "error" in sub:
                    detail["multi_gpu_sublegs_error"] = sub["error"]
        if rank == 0:
            detail["bench_wall_sec"] = time.perf_counter() - t_bench0
            out.emit(detail, world, a.detail_file)
        failed_subleg = allreduce_min_flag(world, not (rank == 0 and detail.get("multi_gpu_sublegs_error"))) is False
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    if failed_subleg:               # the line is out (with multi_gpu_sublegs_error); the launcher must still see that the run was not whole
        sys.exit(3)


if __name__ == "__main__":
    try:
        main()
    finally:
        reference_kill_all()             # a GPU leg that raised (or sys.exit) must not leave a 16 GB reference run behind
