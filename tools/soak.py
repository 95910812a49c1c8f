#!/usr/bin/env python3
"""Soak: many protocol passes in every pass form on one session; device memory in use and step time at the start and at the end (a leak in the pinned ring, the
event pools or the plan's buffers would show as growth; a clock or thermal drift as a slower tail):  python tools/soak.py [BLOCKS] [PASSES]"""
import gzip, os, sys, tempfile, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vp_loader
vp = vp_loader.load(); vp.lib_host()
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 200
hip = ctypes.CDLL("libamdhip64.so")
def used():
    fr, to = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert hip.hipMemGetInfo(ctypes.byref(fr), ctypes.byref(to)) == 0
    return (to.value - fr.value) / 2**20
with tempfile.TemporaryDirectory() as tmp:
    pws = os.path.join(tmp, "s.pws")
    with gzip.open(os.path.join(ROOT, "tests", "golden", "SHA256_64.pws.gz"), "rb") as f, open(pws, "wb") as o:
        o.write(f.read())
    c = vp.Circuit.from_pws(pws, blocks, seed=1)
s = vp.Session(c)
s.draw_protocol_tape()
ref = s.prove_protocol()
for name, kw in (("sync", {}), ("deferred", {"deferred": True}), ("pipelined", {"queue_next": True}), ("sync again", {})):
    for _ in range(3): s.prove_protocol(**kw)
    m0 = used(); ts = []
    for i in range(passes):
        t = time.perf_counter(); out = s.prove_protocol(**kw); ts.append(time.perf_counter() - t)
    same = out[0] == ref[0] and out[1] == ref[1] and (out[2] == ref[2]).all()
    k = max(1, passes // 10)
    print("%-10s %d passes: first tenth %.3f ms, last tenth %.3f ms per pass; device memory in use %.1f -> %.1f MiB; last pass equals the first: %s"
          % (name, passes, 1e3 * sum(ts[:k]) / k, 1e3 * sum(ts[-k:]) / k, m0, used(), same), flush=True)
s.close(); c.close()
