#!/usr/bin/env python3
"""What does the chip give back when the proof's fold launches (latency / HBM bound) run BESIDE a commitment's leaf hash (VALU-issue bound)?
    python tools/overlap_probe.py BLOCKS
Two contexts of one process on one GPU (each has its own stream): a Session proves the GKR part, a stand-alone commitment context commits the same
input layer; each alone (best of 5 walls), then both started together from two host threads (ctypes drops the GIL inside a call), the proof also
started 1 / 3 / 6 ms late so that it falls into the leaf hash rather than the transform.  Prints one JSON line; the proof's bytes and the root are
compared with the solo runs."""
import gzip
import json
import os
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    blocks = int(sys.argv[1])
    import numpy as np
    import vp_loader
    vp = vp_loader.load()
    vp.lib_host()
    with tempfile.TemporaryDirectory() as tmp:
        pws = os.path.join(tmp, "s.pws")
        with gzip.open(os.path.join(ROOT, "tests", "golden", "SHA256_64.pws.gz"), "rb") as f, open(pws, "wb") as o:
            o.write(f.read())
        c = vp.Circuit.from_pws(pws, blocks, seed=1)
    s = vp.Session(c)
    s.draw_tape()
    s.prove_full(batched=True)                 # leaves the proof's last point (the public vector of commit_public is its eq table)
    pub = s.eq_table(s.last_point())
    s.draw_tape()
    tr0, _ = s.prove_gkr()
    inputs, n_bits = s.layer_values(0), c.layer_bitlen(0)
    sc = vp.ShardedCommitment(inputs, n_bits, 1)
    root0 = sc.commit_private()
    rh0 = sc.commit_public(pub)

    def gkr():                                  # the tape stays attached: the same bytes every time
        tr, _ = s.prove_gkr()
        assert tr == tr0

    def cpriv():
        assert sc.commit_private() == root0

    def cpub():
        assert sc.commit_public(pub) == rh0

    def wall(fn, reps=5):
        best = 1e9
        for _ in range(reps):
            t = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t)
        return best * 1e3

    def both(a, b, delay_b_ms, reps=5):
        best = None
        for _ in range(reps):
            t_a, t_b = [0.0], [0.0]
            go = threading.Barrier(3)

            def ra():
                go.wait(); t = time.perf_counter(); a(); t_a[0] = time.perf_counter() - t

            def rb():
                go.wait(); t = time.perf_counter()
                if delay_b_ms:
                    time.sleep(delay_b_ms / 1e3)
                b(); t_b[0] = time.perf_counter() - t
            ta, tb = threading.Thread(target=ra), threading.Thread(target=rb)
            ta.start(); tb.start()
            go.wait(); t = time.perf_counter()
            ta.join(); tb.join()
            w = time.perf_counter() - t
            if best is None or w < best[0]:
                best = (w, t_a[0], t_b[0])
        return {"wall_ms": round(best[0] * 1e3, 3), "a_ms": round(best[1] * 1e3, 3), "b_ms_incl_delay": round(best[2] * 1e3, 3)}

    out = {"blocks": blocks, "alone_ms": {"prove_gkr": round(wall(gkr), 3), "commit_private": round(wall(cpriv), 3), "commit_public": round(wall(cpub), 3)}}
    out["commit_private_beside_prove_gkr"] = {"delay_%g_ms" % d: both(cpriv, gkr, d) for d in (0, 1, 3, 6)}
    out["commit_public_beside_prove_gkr"] = {"delay_%g_ms" % d: both(cpub, gkr, d) for d in (0, 4, 7)}
    print(json.dumps(out))
    sc.close(); s.close(); c.close()


if __name__ == "__main__":
    main()
