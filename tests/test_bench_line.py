"""CPU: bench.py's ONE stdout line.  The driver keeps only so much of stdout: round 3's 24 KB line did not parse.  The line builder
(bench.compact_line) must turn ANY detail dict — the tables of a real run included — into strict JSON of at most 4 KB that carries the
contract fields, `roofline` and `cpu_baseline`; everything else lives in the detail file."""
import glob
import json
import math
import os

import pytest

from conftest import ROOT

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def canned_detail(n_kernels=40, n_launches=400):
    kernels = [{"kernel": "k_kernel_%d" % i, "launches": 3 + i, "total_us": 1234.5678 / (i + 1), "time_share": 1.0 / (i + 2), "avg_launch_us": 12.3456789,
                "algorithmic_MB_per_launch": 4362.065, "GBps": 298.6, "hbm_frac": 0.0373, "work_units": 408943600} for i in range(n_kernels)]
    per_launch = [{"step": i, "kernel": "k_sumfold3b_multi", "jobs": 27, "workgroups": 11328, "rounds": [1, 6], "MB": 1286.766, "us": 297.16, "GBps": 4330.2, "hbm_frac": 0.5413}
                  for i in range(n_launches)]
    return {
        "metric": "prover sec + field-ops/sec, SHA-256 circuit, 1/2/4/8 MI355X (bit-exact)", "value": 1.0123456789e11, "unit": "field-ops/s",
        "value_definition": "x" * 900,
        "n_gpus": 1, "ranks": 1, "steps": 20, "warmup": 5, "ms_per_step": 82.123456789, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64 (F_p^2, p=2^61-1)", "data": "synthetic",
        "config": {"workload": "SHA-256 1024-block circuit (SHA256_64.pws x1024, 102347776 gates, 15 layers, 2^23 input wires), GKR sumcheck + Virgo FFT/LDT commit on GPU (BASELINE configs[2])",
                   "mode": "one prover pass per step", "proofs_per_step": 1, "field_ops_per_proof": 8299342697},
        "prover_sec": {"step_wall": 0.0821, "gkr": 0.0064, "commit_private": 0.0219, "commit_public": 0.0251, "fft_gkr": 0.0044, "fri_commit": 0.0243,
                       "pc_prove_reference_definition": 0.0757, "note": "n" * 300, "step_wall_per_rank": [0.0821]},
        "gkr_field_ops_per_sec": 1.29e12,
        "first_proof_sec": {"gkr_first_call_incl_plan_tuner_and_graph_capture": 1.9, "first_complete_pass_incl_commitment_buffers": 0.4},
        "bit_exact": {"transcript": True, "fri_roots": True, "fri_final_codeword": True, "complete_protocol_accepted": True, "interactive_run_equals_batched": True},
        "bit_exact_all_ranks": True,
        "roofline": {"kernel": "k_leaf_hash", "bound": "valu", "achieved": 9.33e9, "peak": 1.3096722621e10, "unit": "Keccak-f[1600]/s", "frac": 0.7125274069844166,
                     "frac_of_round3_peak_definition": 0.92, "launches": 3, "avg_launch_us": 14607.57, "algorithmic_bytes_per_launch": 4362065000.0, "hbm_GBps": 298.6,
                     "hbm_frac": 0.0373, "traffic": None, "traffic_source": None, "kernel_time_share": 0.55, "peak_definition": "p" * 700, "how": "h" * 300},
        "roofline_gkr_dominant": {"kernel": "k_sumfold3b_gen_multi", "bound": "valu", "frac": 0.38, "hbm_frac": 0.28, "avg_launch_us": 1616.0, "kernel_time_share": 0.5,
                                  "measured_limiter": "m" * 600},
        "roofline_ntt": {"bound": "valu", "frac": 0.29, "hbm_frac": 0.23, "total_us": 19300.0, "time_share": 0.24},
        "cpu_baseline": {"value": 1.2e7, "unit": "field-ops/s", "cores": 1, "kind": "reference", "sample": "the real reference binary, SHA-256 x64, complete protocol, single thread",
                         "prover_sec": 43.1, "gkr_prove_sec": 12.0, "pc_prove_sec": 31.0, "field_ops": 518817605, "host_cpu": "AMD EPYC 9575F 64-Core Processor",
                         "host_cores_visible": 256, "reference_over_port_ratio_x64_same_box": 1.5},
        "interactive_path": {"prover_sec": 0.029, "init_calls_sec": 0.008, "round_calls_sec": 0.02, "transcript_equals_batched": True,
                             "per_round": {"rounds": 859, "hbm_frac_overall": 0.2, "largest_rounds": [{"layer": 2, "us": 36.0}] * 12, "all_rounds": [{"layer": 2, "us": 36.0}] * 859}},
        "verifier": {"complete_protocol": {"accepted": True}},
        "kernels": kernels, "per_launch": per_launch,
        "x64_gkr": {"value": 8.5e11, "ms_per_step": 0.612, "prover_sec_device": 0.000575, "steps": 20, "bit_exact_vs_reference_golden": True,
                    "roofline": {"kernel": "k_sumfold3b_multi", "frac": 0.29, "hbm_frac": 0.28}, "interactive_path": {"prover_sec": 0.0113}, "first_proof_sec": 0.9,
                    "kernels": kernels, "per_launch": per_launch, "cpu_baseline": {"prover_sec": 4.6}, "two_in_flight": {"ms_per_proof": 0.58}},
        "cpu_port_x1024_gkr": {"kind": "port", "gkr_prove_sec": 42.5, "gkr_field_ops_per_sec": 1.95e8, "transcript_equals_gpu": True},
        "rccl_ranks": None,
    }


def check_line(s, detail_file="gpurun_out/bench_detail_n1.json"):
    import bench
    assert "\n" not in s
    assert len(s.encode()) <= bench.LINE_TARGET_BYTES, len(s.encode())
    def no_const(x):
        raise AssertionError("non-finite constant %s in the line" % x)
    o = json.loads(s, parse_constant=no_const)
    for k in CONTRACT:
        assert k in o, k
    assert isinstance(o["config"], dict) and "workload" in o["config"] and "model" not in o["config"]
    r = o["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3
    assert o["detail_file"] == detail_file
    return o


def test_line_is_compact_strict_json_and_keeps_roofline_and_cpu_baseline():
    import bench
    d = canned_detail()
    o = check_line(bench.compact_line(d, "gpurun_out/bench_detail_n1.json"))
    cb = o["cpu_baseline"]
    assert cb["kind"] == "reference" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb
    assert o["bit_exact"] is True and o["rccl_ranks"] is None
    assert o["x64_gkr"]["value"] == pytest.approx(8.5e11)
    assert "kernels" not in o and "per_launch" not in o          # tables live in the detail file


def test_line_survives_nan_inf_numpy_and_huge_tables():
    import numpy as np
    import bench
    d = canned_detail(n_kernels=500, n_launches=5000)
    d["roofline"]["traffic"] = float("nan")
    d["gkr_field_ops_per_sec"] = float("inf")
    d["ms_per_step"] = np.float64(82.5)
    d["steps"] = np.int64(20)
    d["interactive_path"]["prover_sec"] = np.float32(0.029)
    o = check_line(bench.compact_line(d, "gpurun_out/bench_detail_n1.json"))
    assert o["roofline"]["traffic"] is None and o["steps"] == 20 and o["ms_per_step"] == 82.5
    assert o.get("gkr_field_ops_per_sec") is None


def test_line_of_a_failed_multi_rank_subleg_and_of_the_gkr_only_workload():
    import bench
    d = canned_detail()
    d.update({"n_gpus": 8, "ranks": 8, "rccl_ranks": 8, "multi_gpu_sublegs_error": "timed out after 900 s in sharded",
              "sharded": {"error": "timed out", "config": {"workload": "w"}}})
    o = check_line(bench.compact_line(d, "gpurun_out/bench_detail_n8.json"), "gpurun_out/bench_detail_n8.json")
    assert o["n_gpus"] == 8 and o["rccl_ranks"] == 8 and "timed out" in o["multi_gpu_sublegs_error"]
    # the GKR-only workload's detail (--no-pc): scalar prover_sec, bit_exact_vs_reference_golden, no bit_exact dict
    g = {k: d[k] for k in CONTRACT}
    g.update({"prover_sec": 0.000612, "prover_sec_device": 0.000575, "bit_exact_vs_reference_golden": True, "host_verifier_accepts": True,
              "roofline": {"kernel": "k_sumfold3b_multi", "bound": "valu", "achieved": 2.0e11, "peak": 7.0e11, "unit": "F_p^2 multiply-equivalents/s", "frac": 2.0 / 7.0,
                           "hbm_frac": 0.28, "traffic": 72.1e6, "avg_launch_us": 30.9, "algorithmic_bytes_per_launch": 66.9e6},
              "cpu_baseline": {"value": 1.1e8, "unit": "field-ops/s", "cores": 1, "kind": "reference", "sample": "one full GKR proof", "prover_sec": 4.6},
              "kernels": d["kernels"], "rccl_ranks": None})
    o = check_line(bench.compact_line(g, None), None)
    assert o["bit_exact"] is True and o["prover_sec"] == pytest.approx(0.000612) and o["roofline"]["traffic"] == pytest.approx(72.1e6, rel=1e-3)


def test_detail_file_is_strict_json(tmp_path):
    import bench
    d = canned_detail()
    d["roofline"]["traffic"] = float("nan")
    p = bench.write_detail(d, 1, str(tmp_path / "detail.json"))
    o = json.load(open(tmp_path / "detail.json"), parse_constant=lambda x: (_ for _ in ()).throw(AssertionError(x)))
    assert len(o["per_launch"]) == 400 and o["roofline"]["traffic"] is None and p


def test_committed_detail_files_rebuild_into_a_valid_line():
    """Every detail file of a real run committed under profiles/ (r04 on) must go through the line builder as it is."""
    import bench
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[4-9]_bench_detail*.json")))
    for f in files:
        d = json.load(open(f))
        o = check_line(bench.compact_line(d, "x"), "x")
        assert o["value"] is None or math.isfinite(o["value"])


def test_roofline_reads_the_newest_pmc_summary_per_config_and_a_floor_peak():
    """the line's `traffic` comes from the NEWEST committed PMC summary of the same workload (round 5's line still read round 4's file), per
    config — the SHA-256 sizes AND BASELINE configs[4]; the Keccak peak is the guide's uniform 2-cycle floor (a kernel cannot beat it per cycle) and the stamps
    file the `factors` quote says what bench.py says."""
    import re
    import bench
    newest = {}
    for tag in ("b1024", "b64", "randomize_16_20"):
        files = bench.pmc_files(tag)
        assert files, tag
        rounds = [int(os.path.basename(f)[1:3]) for f in files]
        assert rounds == sorted(rounds, reverse=True) and rounds[0] >= 6, (tag, rounds)
        newest[tag] = os.path.basename(files[0])
    t, src = bench.pmc_traffic("b1024", "k_leaf_hash")
    assert src == "profiles/" + newest["b1024"] and 4.2e9 < t < 4.6e9
    t, src = bench.pmc_traffic("randomize_16_20", "k_sumfold3b_gen_multi")
    assert src == "profiles/" + newest["randomize_16_20"] and t > 1e8
    lim = bench.pmc_limiter("b1024", "k_sumfold3b_gen_multi", 1450.0)
    assert newest["b1024"] in lim and "SQ_WAIT_ANY" in lim and "r02" not in lim
    # the floor: 24 rounds x 180 instructions x 2 cycles; the kernel's own cycles per wave-round (stamps) are ABOVE it
    assert bench.KECCAK_FLOOR_CYCLES_PER_ROUND == 360.0 and bench.LEAF_CYCLES_PER_WAVE_ROUND > bench.KECCAK_FLOOR_CYCLES_PER_ROUND
    assert abs(bench.KECCAK_PEAK_PER_S - 1024 * 64 * 2.4e9 / (24 * 360.0)) < 1.0
    txt = open(os.path.join(ROOT, bench.LEAF_STAMPS_FILE)).read()
    kc = [int(x) for x in re.findall(r"workgroup\s+(\d+) kcycles", txt)]
    assert kc and abs(sorted(kc)[len(kc) // 2] - bench.LEAF_WG_KCYCLES) <= 2
    # and the line carries source and factors
    d = canned_detail()
    d["roofline"].update({"peak": bench.KECCAK_PEAK_PER_S, "achieved": 0.68 * bench.KECCAK_PEAK_PER_S, "frac": 0.68, "traffic": t, "traffic_source": src + " (FETCH_SIZE x2)",
                          "factors": {"issue_floor_over_measured_cycles": 0.826, "clock_times_tail_this_run": 0.82, "cycles_per_wave_round": {"floor": 360.0, "measured": 435.6}}})
    o = check_line(bench.compact_line(d, "gpurun_out/bench_detail_n1.json"))
    assert o["roofline"]["traffic_source"] == src and o["roofline"]["factors"]["issue"] == pytest.approx(0.826)


def test_issue_share_record_is_read_and_is_a_fraction():
    """tools/isa_cycles.py's record (profiles/r<NN>_issue_share_<tag>.json): bench.py attaches, per kernel, the share of a launch the SIMDs spend issuing the kernel's
    own VALU stream.  The committed record names the PMC summary it was computed from and covers the two transform passes and the dominant fold launch."""
    import bench
    shares, src = bench.issue_share("b1024")
    assert src and src.startswith("profiles/r") and os.path.exists(os.path.join(ROOT, src))
    for k in ("k_ntt8_colsx<8>", "k_ntt8_rows<false>", "k_sumfold3b_gen_multi"):
        v = shares[k]
        assert 0.3 < v["issue_share_of_launch"] < 1.2 and 3.0 < v["issue_cycles_per_instruction"] < 6.0 and 0.1 < v["multiplier_share_of_issue"] < 0.5
    rec = json.load(open(os.path.join(ROOT, src)))
    assert os.path.exists(os.path.join(ROOT, rec["pmc_summary"]))
    assert bench.issue_share("no_such_workload") == ({}, None)
