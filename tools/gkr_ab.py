#!/usr/bin/env python3
"""Same-call A/B of builds of libvpgpu.so on the batched GKR proof:  python tools/gkr_ab.py BLOCKS [lib.so | - | VAR=val[,VAR=val][@lib.so]] ...   ("-" = the product library;
VAR=val = environment switches of vp_create for that variant; BLOCKS = a number of SHA-256 blocks or rDxW = layeredCircuit::randomize(D, W)).
Each variant runs in its own process (VP_LIBGPU), proves the x BLOCKS circuit 20 times after the first (tuned) proof and prints the mean device time,
the wall time per proof and a digest of the transcript."""
import hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def child(blocks):
    sys.path.insert(0, ROOT)
    import gzip, tempfile
    import vp_loader
    vp = vp_loader.load(); vp.lib_host()
    if blocks.startswith("r"):
        d, w = blocks[1:].split("x")
        c = vp.Circuit.randomize(int(d), int(w), seed=1)
    else:
      with tempfile.TemporaryDirectory() as tmp:
        pws = os.path.join(tmp, "s.pws")
        with gzip.open(os.path.join(ROOT, "tests", "golden", "SHA256_64.pws.gz"), "rb") as f, open(pws, "wb") as o:
            o.write(f.read())
        c = vp.Circuit.from_pws(pws, int(blocks), seed=1)
    s = vp.Session(c)
    s.draw_tape()
    tr, _ = s.prove_gkr()
    for _ in range(3): s.prove_gkr()
    dev = []; t = time.time()
    for _ in range(20):
        tr2, res = s.prove_gkr(); dev.append(res["gkr_device_ms"])
    wall = (time.time() - t) / 20
    s.set_profiling(1); s.prove_gkr(); st = s.launch_stats(); s.set_profiling(0)
    kern = {}
    for e in st:
        k = kern.setdefault(e["kernel"], [0, 0.0]); k[0] += 1; k[1] += e["us"]
    print(json.dumps({"device_ms": sum(dev) / len(dev), "min_ms": min(dev), "wall_ms": wall * 1e3, "digest": hashlib.sha256(tr).hexdigest()[:16], "same": tr == tr2,
                      "kernels": {k: [v[0], round(v[1] / 1e3, 3)] for k, v in sorted(kern.items(), key=lambda x: -x[1][1])[:5]}}))

if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for lib in sys.argv[2:] * 2:
            env = dict(os.environ)
            env.pop("VP_LIBGPU", None)
            sw, _, so = lib.partition("@") if "=" in lib else ("", "", lib)
            for kv in filter(None, sw.split(",")):
                k, v = kv.split("="); env[k] = v
            if so and so != "-": env["VP_LIBGPU"] = os.path.abspath(so)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", sys.argv[1]], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print("%-36s %s" % (lib, line[0] if line else ("FAILED: " + r.stderr[-400:])), flush=True)
