#!/usr/bin/env python3
"""Does k_leaf_hash's duration depend on which transform kernels ran before it?  One process, the x BLOCKS circuit, two sessions that differ in
vp_options.ntt_r8 only, commit_private profiled launch by launch, sessions alternating:  python tools/leaf_probe.py [BLOCKS]"""
import gzip
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vp_loader

vp = vp_loader.load()
vp.lib_host()
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
with tempfile.TemporaryDirectory() as tmp:
    pws = os.path.join(tmp, "s.pws")
    with gzip.open(os.path.join(ROOT, "tests", "golden", "SHA256_64.pws.gz"), "rb") as f, open(pws, "wb") as o:
        o.write(f.read())
    c = vp.Circuit.from_pws(pws, blocks, seed=1)
sess = {v: vp.Session(c, options=vp.Options(ntt_r8=v)) for v in (0, 1)}
for v in (0, 1):
    sess[v].commit_private()
for rep in range(3):
    for v in (0, 1):
        s = sess[v]
        s.set_profiling(1)
        root, ms = s.commit_private()
        st = s.launch_stats()
        s.set_profiling(0)
        print("ntt_r8=%d  total %.2f ms | " % (v, ms) + "  ".join("%s %.2f" % (e["kernel"].replace("k_", ""), e["us"] / 1e3) for e in st if e["us"] > 300), flush=True)
