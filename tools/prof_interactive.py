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
# round classes: wall time of the vp_round calls by how the round ran and by the size of its tables
import collections, math
cls = collections.OrderedDict()
for e in s.round_stats():
    k = (e["how"], int(math.log2(max(1, e["bytes"]))))
    c = cls.setdefault(k, [0, 0.0, 0])
    c[0] += 1; c[1] += e["us"]; c[2] += e["bytes"]
print("how  log2(bytes)  rounds   total us   us/round   GB/s")
for (how, lb), (n, us, by) in sorted(cls.items()):
    print("%3d  %10d  %6d  %9.1f  %9.2f  %7.1f" % (how, lb, n, us, us / n, by / us / 1e3))
print("sum of round calls %.3f ms over %d rounds" % (sum(c[1] for c in cls.values()) / 1e3, sum(c[0] for c in cls.values())))
