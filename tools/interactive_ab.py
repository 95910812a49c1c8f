#!/usr/bin/env python3
"""Drop-in (interactive) path: init and round 1 as separate launches (VP_FUSE_R1=0) against the init pass that also sums round 1 (k_round1_gen), same process,
alternating sessions:  python tools/interactive_ab.py [BLOCKS ...]
(the fused pass was measured neutral and removed again: commit eff9a98 holds it, profiles/r05_ab_interactive_init_with_round1_x16_x64_x1024.txt the numbers)"""
import gzip, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vp_loader
vp = vp_loader.load(); vp.lib_host()
with tempfile.TemporaryDirectory() as tmp:
    pws = os.path.join(tmp, "s.pws")
    with gzip.open(os.path.join(ROOT, "tests", "golden", "SHA256_64.pws.gz"), "rb") as f, open(pws, "wb") as o:
        o.write(f.read())
    for blocks in [int(x) for x in sys.argv[1:]] or [64, 1024]:
        c = vp.Circuit.from_pws(pws, blocks, seed=1)
        sess = {}
        for v in ("0", "1"):
            os.environ["VP_FUSE_R1"] = v
            sess[v] = vp.Session(c)
        os.environ.pop("VP_FUSE_R1")
        sess["1"].draw_tape()
        ref, _ = sess["1"].prove_gkr()
        for rep in range(3):
            for v in ("0", "1"):
                s = sess[v]
                tr, res, ok = s.prove_interactive()
                cls = {}
                for e in s.round_stats():
                    k = cls.setdefault(e["how"], [0, 0.0]); k[0] += 1; k[1] += e["us"]
                print("x%d VP_FUSE_R1=%s  prover_sec %.3f ms (init calls %.3f, rounds %.3f)  identical to the batched proof %s, verified %s | %s" % (
                    blocks, v, res["prove_sec"] * 1e3, res.get("init_sec", 0) * 1e3, res.get("round_sec", 0) * 1e3, tr == ref, ok,
                    "  ".join("how %d: %d rounds %.2f ms" % (h, n, us / 1e3) for h, (n, us) in sorted(cls.items()))), flush=True)
        for s in sess.values(): s.close()
        c.close()
