#!/usr/bin/env python3
"""One interactive proof (vp_round per verifier message) of SHA-256 x BLOCKS for rocprofv3:  rocprofv3 --kernel-trace --stats -- python3 tools/prof_interactive.py 1024"""
import gzip, os, sys, tempfile
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
s = vp.Session(c)
for _ in range(2):
    tr, res, ok = s.prove_interactive()
print("interactive prover_sec %.4f init %.4f rounds %.4f ok %s" % (res["prove_sec"], res["init_sec"], res["round_sec"], ok))
