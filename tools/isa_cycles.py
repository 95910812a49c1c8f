#!/usr/bin/env python3
"""VALU issue time of a kernel from its OWN instruction stream:  tools/isa_cycles.py vpgpu.s PMC_SUMMARY.json KERNEL[=pmc name] ...
  static instruction mix of the kernel in a gfx950 listing (hipcc -S --cuda-device-only; tools/isa_mix.py's parser)
  x measured issue cost per instruction class (profiles/r02_micro_rates.txt, r04_micro_keccak_instruction_classes.txt: SIMD cycles per wave-instruction
    at the nominal 2.4 GHz with 8 waves per SIMD; classes not measured one by one take the cost of their nearest measured neighbour, listed below)
  -> mean cycles per VALU instruction of THIS stream;
  x the launch's VALU wave-instructions (SQ_INSTS_VALU, PMC summary) / 1024 SIMDs / 2.4 GHz -> the time the SIMDs need just to issue the stream,
  set against the launch's measured duration (single-stream kernel trace in the same summary): the fraction of the launch the VALUs are issuing.
The static mix stands for the dynamic one: exact for the straight-line transform kernels (the coset / row loop is > 85 % of the listing), approximate for the
fold kernels (three init modes and seven gate types in one listing)."""
import collections
import json
import re
import sys

# SIMD cycles per wave-instruction (nominal clock), measured: r02_micro_rates.txt unless noted
COST = {
    "v_mad_u64_u32": 7.04, "v_lshl_add_u64": 5.16, "v_mul_lo_u32": 5.02, "v_mul_hi_u32": 4.63, "v_fma_f64": 4.73, "v_add_u32": 2.93, "v_mad_u32_u24": 4.70,
    "v_lshlrev_b64": 4.44, "v_bitop3_b32": 3.20, "v_alignbit_b32": 4.43, "v_xor_b32": 2.78,
    # neighbours (same encoding class / same measured family):
    "v_lshrrev_b64": 4.44, "v_ashrrev_i64": 4.44,                              # 64-bit shifts
    "v_add_co_u32": 4.72, "v_addc_co_u32": 4.72, "v_sub_co_u32": 4.72, "v_subb_co_u32": 4.72, "v_subrev_co_u32": 4.72, "v_subbrev_co_u32": 4.72,   # add_co + addc: 9.43 for the two
    "v_and_b32": 2.78, "v_or_b32": 2.78, "v_mov_b32": 2.78, "v_not_b32": 2.78, "v_sub_u32": 2.93, "v_subrev_u32": 2.93, "v_cndmask_b32": 2.93,
    "v_lshrrev_b32": 3.2, "v_lshlrev_b32": 3.2, "v_ashrrev_i32": 3.2, "v_bfe_i32": 4.4, "v_bfe_u32": 4.4,      # r04: v_lshlrev_b32 3.2; VOP3-only integer 4.2-4.6
    "v_mov_b32_dpp": 4.4, "v_lshl_or_b32": 4.4, "v_add3_u32": 4.4, "v_lshl_add_u32": 4.4, "v_and_or_b32": 4.4, "v_or3_b32": 4.4, "v_perm_b32": 4.4,
    "v_cmp_lt_u64": 4.4, "v_cmp_gt_u64": 4.4, "v_cmp_eq_u64": 4.4, "v_cmp_ne_u64": 4.4, "v_cmp_ge_u64": 4.4, "v_cmp_le_u64": 4.4, "v_mov_b64": 4.4,
    "v_readlane_b32": 4.4, "v_readfirstlane_b32": 4.4, "v_writelane_b32": 4.4,
}
DEFAULT = 3.2


def base(op):
    op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    return op


def mixes(listing, want):
    cur = None
    out = {}
    for ln in open(listing):
        ln = ln.rstrip("\n")
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = None
            for w in want:
                if w in m.group(1):
                    cur = m.group(1); out[cur] = collections.Counter()
            continue
        if cur is None or ln.startswith("\t.") or (ln.startswith(".L") and ln.endswith(":")):
            continue
        t = ln.strip().split()
        if not t or not t[0].startswith("v_"):
            continue
        op = t[0]
        if op.endswith("_dpp") or "dpp" in ln and "row_" in ln or "quad_perm" in ln:
            op = "v_mov_b32_dpp" if op.startswith("v_mov_b32") else op
        out[cur][base(op) if op != "v_mov_b32_dpp" else op] += 1
    return out


def main():
    json_out = None
    if "--json" in sys.argv:
        k = sys.argv.index("--json"); json_out = sys.argv[k + 1]; del sys.argv[k:k + 2]
    listing, pmc = sys.argv[1], json.load(open(sys.argv[2]))
    record = {"how": "tools/isa_cycles.py: static VALU mix of the compiled kernel x measured issue cost per instruction class (profiles/r02_micro_rates.txt, "
                     "r04_micro_keccak_instruction_classes.txt; nominal-clock SIMD cycles) x SQ_INSTS_VALU of the launch / 1024 SIMDs / 2.4 GHz, over the launch's "
                     "single-stream duration in the same PMC summary", "pmc_summary": sys.argv[2], "kernels": {}}
    kern = {k["kernel"].replace("vp::", "").replace("void ", ""): k for k in pmc["kernels"]}
    want = [a.split("=")[0] for a in sys.argv[3:]]
    pmc_name = {a.split("=")[0]: (a.split("=")[1] if "=" in a else None) for a in sys.argv[3:]}
    for name, c in mixes(listing, want).items():
        tot = sum(c.values())
        cyc = sum(v * COST.get(o, DEFAULT) for o, v in c.items())
        by = collections.Counter()
        for o, v in c.items():
            by[o] += v * COST.get(o, DEFAULT)
        key = next(w for w in want if w in name)
        print("%s\n  VALU instructions (static) %d, issue cost of the stream %.0f cycles = %.2f per instruction" % (name[:110], tot, cyc, cyc / tot))
        print("  by class (share of the issue cost): " + "  ".join("%s %d x %.2f = %.0f%%" % (o, c[o], COST.get(o, DEFAULT), 100 * v / cyc) for o, v in by.most_common(8)))
        unknown = sorted((o for o in c if o not in COST), key=lambda o: -c[o])[:8]
        if unknown:
            print("  classes at the default %.1f: %s" % (DEFAULT, " ".join("%s(%d)" % (o, c[o]) for o in unknown)))
        pk = pmc_name[key]
        rec = kern.get(pk) if pk else None
        if rec and rec.get("single_stream_avg_launch_us"):
            # run totals on both sides, per launch of the trace: the counter pass and the single-stream trace may replay plans with different launch counts
            n = rec["SQ_INSTS_VALU_per_launch"] * rec["launches"] / rec.get("single_stream_launches", rec["launches"])
            us = rec["single_stream_avg_launch_us"]
            issue_us = n / 1024.0 * (cyc / tot) / 2400.0
            print("  launch: %.3e VALU wave-instructions (SQ_INSTS_VALU), %.1f us (single-stream trace) -> the SIMDs need %.1f us to issue them: %.2f of the launch"
                  % (n, us, issue_us, issue_us / us))
            record["kernels"][pk] = {"static_valu_instructions": tot, "issue_cycles_per_instruction": round(cyc / tot, 3), "valu_wave_instructions_per_launch": n,
                                     "launch_us": us, "issue_us": round(issue_us, 1), "issue_share_of_launch": round(issue_us / us, 3),
                                     "multiplier_share_of_issue": round(by.get("v_mad_u64_u32", 0.0) / cyc, 3)}
    if json_out:
        json.dump(record, open(json_out, "w"), indent=1)


if __name__ == "__main__":
    main()
