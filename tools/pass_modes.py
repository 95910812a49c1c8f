#!/usr/bin/env python3
"""The three forms of the protocol pass (vphost.h: synchronous calls | deferred completion | deferred + the next pass's head queued behind the folds):
same bytes, wall time per pass.   python tools/pass_modes.py [BLOCKS] [PASSES]"""
import gzip, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vp_loader
vp = vp_loader.load(); vp.lib_host()
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
with tempfile.TemporaryDirectory() as tmp:
    pws = os.path.join(tmp, "s.pws")
    with gzip.open(os.path.join(ROOT, "tests", "golden", "SHA256_64.pws.gz"), "rb") as f, open(pws, "wb") as o:
        o.write(f.read())
    c = vp.Circuit.from_pws(pws, blocks, seed=1)
s = vp.Session(c)
s.draw_protocol_tape()
ref = s.prove_protocol(deferred=False)
for _ in range(3): s.prove_protocol(deferred=False)
def same(a, b): return a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2])
def run(tag, **kw):
    import torch
    for _ in range(3): r = s.prove_protocol(**kw)
    torch.cuda.synchronize()
    ok = True; parts = {}
    t = time.time()
    for i in range(K):
        r = s.prove_protocol(**kw)
        ok &= same(r, ref)
        for k, v in r[3].items(): parts[k] = parts.get(k, 0) + v / K
    torch.cuda.synchronize()
    dt = (time.time() - t) / K
    print("%-34s %8.3f ms per pass  identical %s   %s" % (tag, dt * 1e3, ok, "  ".join("%s %.2f" % (k, v * 1e3) for k, v in parts.items())), flush=True)
for rep in range(2):
    run("synchronous calls", deferred=False)
    run("deferred completion", deferred=True)
    run("deferred + next head queued", queue_next=True)
# a pass without the flag after pipelined ones (its head is there), then the interactive path and an opening still work
r = s.prove_protocol(deferred=True)
print("pass after the pipelined ones identical:", same(r, ref))
tr, res, okv = s.prove_interactive()
print("interactive proof after it: verified", okv, " GKR slice equal:", tr == ref[0][32:32 + len(tr)])
