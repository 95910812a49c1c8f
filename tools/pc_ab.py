#!/usr/bin/env python3
"""Same-call A/B of builds of libvpgpu.so on the commitment:  python tools/pc_ab.py BLOCKS [lib.so | - | VP_NAME=value] ...   ("-" = the product library;
VP_NAME=value = the product library with that tuning variable set).
Each variant runs in its own process (VP_LIBGPU), commits the x BLOCKS input layer twice (private + public on the protocol's eq table + FRI)
and prints the per-kernel totals of a profiled pass and the three calls' device times; roots are compared between the variants."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(blocks):
    sys.path.insert(0, ROOT)
    import gzip, tempfile
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
    full, ok = s.prove_full(batched=True)
    pub = s.eq_table(s.last_point())
    st = c.layer_bitlen(0) - 6
    rr = np.random.default_rng(1).integers(0, (1 << 61) - 1, size=(st, 2), dtype=np.uint64)
    best = None
    for _ in range(3):
        r1, ms1 = s.commit_private()
        rh, inner, alls, ms2 = s.commit_public(pub)
        roots, fin = s.fri_commit(rr); ms3 = s.commit_device_ms()
        t = (ms1 + ms2 + ms3, ms1, ms2, ms3)
        best = t if best is None or t[0] < best[0] else best
    s.set_profiling(1)
    s.commit_private(); a = s.launch_stats()
    s.commit_public(pub); b = s.launch_stats()
    s.fri_commit(rr); d = s.launch_stats()
    s.set_profiling(0)
    kern = {}
    for e in a + b + d:
        k = kern.setdefault(e["kernel"], [0, 0.0])
        k[0] += 1; k[1] += e["us"]
    import hashlib
    print(json.dumps({"commit_side_ms": best[0], "private": best[1], "public": best[2], "fri": best[3],
                      "kernels": {k: [v[0], round(v[1] / 1e3, 3)] for k, v in sorted(kern.items(), key=lambda x: -x[1][1])},
                      "digest": hashlib.sha256(full + r1 + rh + inner + alls + roots + fin.tobytes()).hexdigest()[:16]}))


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(int(sys.argv[2]))
    else:
        blocks = sys.argv[1]
        for lib in sys.argv[2:] * 2:
            env = dict(os.environ)
            if "=" in lib and lib.startswith("VP_"):
                env[lib.split("=", 1)[0]] = lib.split("=", 1)[1]
                env.pop("VP_LIBGPU", None)
            elif lib != "-":
                env["VP_LIBGPU"] = os.path.abspath(lib)
            else:
                env.pop("VP_LIBGPU", None)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", blocks], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print("%-40s %s" % (lib, line[0] if line else ("FAILED: " + r.stderr[-400:])), flush=True)
