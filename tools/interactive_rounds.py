#!/usr/bin/env python3
"""Development aid: per-message latency of the interactive path (VP_DEBUG_ROUNDS=1 makes host/prover.cpp print one line per vp_round).
    VP_DEBUG_ROUNDS=1 python tools/interactive_rounds.py [blocks] 2> rounds.txt"""
import gzip, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vp_loader
vp = vp_loader.load()
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 64
with tempfile.TemporaryDirectory() as tmp:
    p = os.path.join(tmp, "SHA256_64.pws")
    with gzip.open(os.path.join(ROOT, "tests", "golden", "SHA256_64.pws.gz"), "rb") as f, open(p, "wb") as g:
        g.write(f.read())
    c = vp.Circuit.from_pws(p, blocks, seed=1)
s = vp.Session(c)
s.prove_interactive()
tr, res, ok = s.prove_interactive()
print({k: res[k] for k in ("prove_sec", "init_sec", "round_sec", "finalize_sec")}, ok)
