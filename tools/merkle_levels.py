#!/usr/bin/env python3
"""Per-launch device time of the Merkle-tree kernels of one commit_private / FRI commit phase (vp_set_profiling): python tools/merkle_levels.py [BLOCKS [all]]"""
import gzip, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vp_loader
vp = vp_loader.load(); vp.lib_host()
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
with tempfile.TemporaryDirectory() as tmp:
    pws = os.path.join(tmp, "s.pws")
    with gzip.open(os.path.join(ROOT, "tests", "golden", "SHA256_64.pws.gz"), "rb") as f, open(pws, "wb") as o:
        o.write(f.read())
    c = vp.Circuit.from_pws(pws, blocks, seed=1)
s = vp.Session(c); s.draw_tape()
full, ok = s.prove_full(batched=True)
pub = s.eq_table(s.last_point())
st = c.layer_bitlen(0) - 6
rr = np.random.default_rng(1).integers(0, (1 << 61) - 1, size=(st, 2), dtype=np.uint64)
for _ in range(2):
    s.commit_private(); s.commit_public(pub); s.fri_commit(rr)
s.set_profiling(1)
for name, call in (("commit_private", lambda: s.commit_private()), ("commit_public", lambda: s.commit_public(pub)), ("fri_commit", lambda: s.fri_commit(rr))):
    call()
    ls = s.launch_stats()
    tot = sum(e["us"] for e in ls)
    print("%s: %d launches, %.3f ms" % (name, len(ls), tot / 1e3))
    for e in ls:
        if len(sys.argv) > 2 or e["kernel"] in ("k_merkle", "k_fri_fold", "k_pc_pointwise"):
            print("    %-16s grid %7d jobs %3d  %9.1f us  %s" % (e["kernel"], e.get("grid", 0), e.get("jobs", 0), e["us"], {k: v for k, v in e.items() if k not in ("kernel", "grid", "jobs", "us")}))
s.set_profiling(0)
