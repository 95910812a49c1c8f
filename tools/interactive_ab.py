#!/usr/bin/env python3
"""Drop-in (interactive) path, same process, alternating sessions that differ in ONE environment switch read at vp_create:
    python tools/interactive_ab.py [--env VP_ARM_ROUNDS] [BLOCKS ...]
Both switches it was written for are experiments that were measured and removed again (the tool stays for the next one):
VP_ARM_ROUNDS — the next round's launch queued ahead, waiting on the device for its challenge: slower, profiles/r05_ab_interactive_armed_rounds.txt;
VP_FUSE_R1 — the init pass that also sums round 1: neutral, commit eff9a98, profiles/r05_ab_interactive_init_with_round1_*.txt."""
import gzip, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vp_loader
vp = vp_loader.load(); vp.lib_host()
with tempfile.TemporaryDirectory() as tmp:
    pws = os.path.join(tmp, "s.pws")
    with gzip.open(os.path.join(ROOT, "tests", "golden", "SHA256_64.pws.gz"), "rb") as f, open(pws, "wb") as o:
        o.write(f.read())
    argv = sys.argv[1:]
    ENV = "VP_ARM_ROUNDS"
    if argv and argv[0] == "--env":
        ENV = argv[1]; argv = argv[2:]
    for blocks in [int(x) for x in argv] or [64, 1024]:
        c = vp.Circuit.from_pws(pws, blocks, seed=1)
        sess = {}
        for v in ("0", "1"):
            os.environ[ENV] = v
            sess[v] = vp.Session(c)
        os.environ.pop(ENV)
        sess["1"].draw_tape()
        ref, _ = sess["1"].prove_gkr()
        for rep in range(3):
            for v in ("0", "1"):
                s = sess[v]
                tr, res, ok = s.prove_interactive()
                cls = {}
                for e in s.round_stats():
                    k = cls.setdefault(e["how"], [0, 0.0]); k[0] += 1; k[1] += e["us"]
                print(("x%d " + ENV + "=%s  prover_sec %.3f ms (init calls %.3f, rounds %.3f)  identical to the batched proof %s, verified %s | %s") % (
                    blocks, v, res["prove_sec"] * 1e3, res.get("init_sec", 0) * 1e3, res.get("round_sec", 0) * 1e3, tr == ref, ok,
                    "  ".join("how %d: %d rounds %.2f ms" % (h, n, us / 1e3) for h, (n, us) in sorted(cls.items()))), flush=True)
        for s in sess.values(): s.close()
        c.close()
