#!/usr/bin/env python3
"""Goldens of lib/virgo's commitment with NON-ZERO masks, from the REAL reference (oracle/_ref/ref_run --pc-masked: poly_commit_prover::commit_private_array /
commit_public_array / commit_phase called directly — the reference's own prover never passes a mask, src/prover.cpp:526).  Build container only.

    python tests/golden/make_pc_masked.py     -> tests/golden/pc_masked_<case>.bin (root_l | root_h | all_sum[65] | openings), pc_masked_fri_<case>.bin, pc_masked.json
"""
import hashlib
import json
import os
import struct
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pc_masked_inputs as pmi

REF_RUN = os.path.join(ROOT, "oracle", "_ref", "ref_run")


def main():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True, stdout=subprocess.DEVNULL)
    meta = {}
    for name in pmi.CASES:
        x = pmi.inputs(name)
        with tempfile.TemporaryDirectory() as tmp:
            inp = os.path.join(tmp, "in.bin")
            pmi.write_case_file(name, inp)
            rec, fri = os.path.join(HERE, "pc_masked_%s.bin" % name), os.path.join(HERE, "pc_masked_fri_%s.bin" % name)
            r = subprocess.run([REF_RUN, "--pc-masked", inp, "--dump", rec, "--dump-fri", fri], stdout=subprocess.PIPE, text=True, check=True)
        gap = int(r.stdout.split("mask_position_gap")[1].split()[0]); steps = int(r.stdout.split("steps")[1].split()[0])
        meta[name] = {"n": x["n"], "m": x["m"], "seed": pmi.CASES[name][2], "mask_position_gap": gap, "fri_steps": steps,
                      "record": os.path.basename(rec), "fri": os.path.basename(fri), "record_sha256": hashlib.sha256(open(rec, "rb").read()).hexdigest(),
                      "fri_sha256": hashlib.sha256(open(fri, "rb").read()).hexdigest(),
                      "origin": "real reference: oracle/_ref/ref_run --pc-masked (commit_private_array / commit_public_array / commit_phase with the masks of tests/pc_masked_inputs.py)"}
        print(name, r.stdout.strip(), os.path.getsize(rec), os.path.getsize(fri))
    json.dump(meta, open(os.path.join(HERE, "pc_masked.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
