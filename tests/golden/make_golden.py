#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REAL reference.

Runs only in the build container (needs /root/reference).  It builds
oracle/_ref/ref_run (the unmodified reference sources compiled in place by
oracle/Makefile, driven through oracle/ref_hook.hpp) and records, for each
case, the transcript of prover messages (layout: SURVEY.md §8c) together with
what the reference printed (prove time, field-op counters, rounds, circuit
structure hash).  The fixtures are data only: transcripts, counters, hashes,
and a gzip copy of the reference's own circuit data file data/SHA256_64.pws.

    python tests/golden/make_golden.py            # x1, randomize(8,12), x16
    python tests/golden/make_golden.py --with-x64 # also the 64-block case (~1 min)
    python tests/golden/make_golden.py --with-big # also BASELINE configs[2] / [4]: SHA-256 x1024 with the commitment (round 3: 131 s circuit
                                                  # build + 809 s verify(), peak RSS 63 GB — needs the whole 62 GiB container, run nothing
                                                  # beside it) and randomize(16, 20) with the commitment (~2 min, 4 GB)
"""
import argparse
import gzip
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("VP_REFERENCE", "/root/reference")
REF_RUN = os.path.join(ROOT, "oracle", "_ref", "ref_run")
PWS = os.path.join(REF, "data", "SHA256_64.pws")

# SHA-256 of the full transcripts as recorded in SURVEY.md §8c / §8d (instrumented reference build
# of the survey).  make_golden refuses to write a fixture that disagrees.
sys.path.insert(0, os.path.join(ROOT, "tests"))

# custom circuits (all gate types + assert gates; tests/custom_circuits.py): name -> (seed, layer sizes).  Input layers of
# > 512 gates: the commitment needs bit length >= 7 (vpd_verifier.cpp:115), and the reference's own FFT only works for
# transforms of >= 8 points (RS_polynomial.cpp:104-133: for order 4 the `dep == 1` loop runs zero times and stale scratch is
# read), i.e. slices of >= 8 elements = input bit length >= 9.  Below that its Merkle roots are not a function of the input.
CUSTOM = {"custom_a": (101, [600, 180, 150, 300, 64, 9]), "custom_b": (102, [1500, 2100, 900, 4100, 700])}

SURVEY_SHA256 = {
    "sha256_x1": "7d56df550455f8e32dcda3ea158e2606b23f4e8bac761ca6a081b8caeee65047",
    "sha256_x16": "d9c442312561023d237a0c8ea1a40f26273c8d5a967028ab3f1bfeafe22d179f",
    "sha256_x64": "69974a97b58f46102549d723b24f5cd6677f7c1102347f979aa4d26483274682",
    "randomize_8_12": "6caa064a89e026000f352b1919b88e0735b67e7c760f752b9e4c23828c6919c0",
}


def run_case(name, args):
    out_bin = os.path.join(HERE, f"transcript_{name}.bin")
    fri_bin = os.path.join(HERE, f"fri_{name}.bin")
    cmd = [REF_RUN] + args + ["--pc", "1", "--dump", out_bin, "--dump-fri", fri_bin]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, check=True)
    txt = res.stdout
    data = open(out_bin, "rb").read()
    digest = hashlib.sha256(data).hexdigest()
    if name in SURVEY_SHA256 and SURVEY_SHA256[name] != digest:
        raise SystemExit(f"{name}: transcript digest {digest} != SURVEY.md {SURVEY_SHA256[name]}")
    m = re.search(r"circuit layers (\d+) gates (\d+) hash ([0-9a-f]{32})", txt)
    c = re.search(r"mult counter (-?\d+), add counter (-?\d+)", txt)      # `int` counters: they wrap at x1024
    r = re.search(r"rounds (\d+)", txt)
    pt = re.search(r"Prove Time ([0-9.]+)", txt)
    pc = re.search(r"Polynomial commitment: prove time ([0-9.]+)", txt)
    ps = re.search(r"proof size = ([0-9.]+) kb", txt)
    fri = open(fri_bin, "rb").read()
    n_steps = (len(fri) - (16 * 128 + 32) * 16) // 48
    return {
        "fri": os.path.basename(fri_bin),           # n_steps x (challenge[16] | root[32]) | final codeword[2048 F] | mask codeword[32 F]
        "fri_steps": n_steps,
        "fri_sha256": hashlib.sha256(fri).hexdigest(),
        "transcript": os.path.basename(out_bin),
        "bytes": len(data),
        "sha256": digest,
        "gkr_slice": [32, len(data) - 32 - 16 - 65 * 16],   # [start, end) of the GKR messages
        "layers": int(m.group(1)),
        "gates": int(m.group(2)),
        "circuit_hash": m.group(3),
        "mult_counter": int(c.group(1)),
        "add_counter": int(c.group(2)),
        "rounds": int(r.group(1)),
        "proof_kb": float(ps.group(1)),
        "reference_prove_sec_here": float(pt.group(1)),
        "reference_pc_prove_sec_here": float(pc.group(1)),
        "args": args,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--with-x64", action="store_true")
    ap.add_argument("--with-big", action="store_true")
    a = ap.parse_args()
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True, stdout=subprocess.DEVNULL)
    pws_gz = os.path.join(HERE, "SHA256_64.pws.gz")
    with open(PWS, "rb") as f, gzip.GzipFile(pws_gz, "wb", mtime=0) as g:
        shutil.copyfileobj(f, g)
    tmp_pws = os.path.join(HERE, "_SHA256_64.pws")
    shutil.copyfile(PWS, tmp_pws)
    meta_path = os.path.join(HERE, "golden.json")
    meta = json.load(open(meta_path)) if os.path.exists(meta_path) else {}
    try:
        meta["sha256_x1"] = run_case("sha256_x1", ["--pws", tmp_pws, "--blocks", "1"])
        # the reference's own regex parser must give the same transcript as the replicating reader
        chk = os.path.join(HERE, "_chk.bin")
        subprocess.run([REF_RUN, "--pws", tmp_pws, "--ref-parser", "--dump", chk], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        assert hashlib.sha256(open(chk, "rb").read()).hexdigest() == meta["sha256_x1"]["sha256"]
        os.remove(chk)
        meta["randomize_8_12"] = run_case("randomize_8_12", ["--randomize", "8", "12"])
        meta["sha256_x16"] = run_case("sha256_x16", ["--pws", tmp_pws, "--blocks", "16"])
        import struct
        import custom_circuits as cc
        for cname, (seed, sizes) in CUSTOM.items():
            szs, ty, l, u, v, c, asr = cc.make(seed, sizes)
            path = os.path.join(HERE, "_%s.circ" % cname)
            with open(path, "wb") as f:
                f.write(struct.pack("<i", len(szs)))
                f.write(szs.tobytes())
                for g in range(len(ty)):
                    f.write(struct.pack("<iiQQQQB", int(ty[g]), int(l[g]), int(u[g]), int(v[g]), int(c[g][0]), int(c[g][1]), int(asr[g])))
            meta[cname] = run_case(cname, ["--custom", path])
            meta[cname]["custom"] = {"seed": seed, "sizes": sizes}
            os.remove(path)
        if a.with_x64:
            meta["sha256_x64"] = run_case("sha256_x64", ["--pws", tmp_pws, "--blocks", "64"])
        if a.with_big:
            # the reference's int counters wrap at these sizes: keep what it printed beside the 64-bit counts of the oracle fixture
            # (congruent mod 2^32, checked by tests/test_oracle_golden.py)
            for name, args in (("randomize_16_20", ["--randomize", "16", "20"]), ("sha256_x1024", ["--pws", tmp_pws, "--blocks", "1024"])):
                old = meta.get(name, {})
                new = run_case(name, args)
                new["mult_counter_printed_int32"], new["add_counter_printed_int32"] = new["mult_counter"], new["add_counter"]
                for k in ("mult_counter", "add_counter", "pairs", "oracle_fixture"):
                    if k in old:
                        new[k] = old[k]
                new["origin"] = "real reference (oracle/_ref/ref_run %s --pc 1 --dump --dump-fri; tests/golden/make_golden.py --with-big)" % " ".join(args[:1] + args[-2:])
                meta[name] = new
    finally:
        os.remove(tmp_pws)
    for k in meta:
        meta[k]["args"] = [os.path.basename(x) if x.startswith(HERE) else x for x in meta[k]["args"]]
    json.dump(meta, open(meta_path, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: (v["sha256"], v["mult_counter"], v["add_counter"]) for k, v in meta.items()}, indent=1))


if __name__ == "__main__":
    sys.exit(main())
