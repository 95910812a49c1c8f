#!/usr/bin/env python3
"""First-use cost of entry points in a FRESH process: python tools/first_use_probe.py [BLOCKS]   (x1024 by default)
Times vp_fft_gkr (lg = input bit length - 6) three times behind vp_warm, then the calls of a protocol pass one by one, twice.  AMD_LOG_LEVEL etc. are the caller's."""
import ctypes, gzip, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vp_loader
vp = vp_loader.load(); vp.lib_host()
L = vp.lib_gpu()
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
with tempfile.TemporaryDirectory() as tmp:
    pws = os.path.join(tmp, "s.pws")
    with gzip.open(os.path.join(ROOT, "tests", "golden", "SHA256_64.pws.gz"), "rb") as f, open(pws, "wb") as o:
        o.write(f.read())
    c = vp.Circuit.from_pws(pws, blocks, seed=1)
s = vp.Session(c)
ctx = s.gpu_ctx()
t = time.perf_counter(); s.warm(); print("vp_warm %.1f ms" % (1e3 * (time.perf_counter() - t)), flush=True)
lg = c.layer_bitlen(0) - 6
nt, nm = ctypes.c_uint64(0), ctypes.c_uint64(0)
L.vp_fft_gkr_sizes(lg, ctypes.byref(nt), ctypes.byref(nm))
tape = np.random.default_rng(5).integers(0, (1 << 61) - 1, size=(nt.value, 2), dtype=np.uint64)
out = np.zeros((nm.value, 2), dtype=np.uint64); w = ctypes.c_uint64(0)
for i in range(0 if os.environ.get('PROBE_SKIP_FFT') else 3):
    t = time.perf_counter()
    rc = L.vp_fft_gkr(ctx, lg, tape.ctypes.data, nt.value, out.ctypes.data, nm.value, ctypes.byref(w))
    print("vp_fft_gkr(lg %d) call %d: %.2f ms rc %d" % (lg, i, 1e3 * (time.perf_counter() - t), rc), flush=True)
s.draw_protocol_tape()
for i in range(2):
    t = time.perf_counter(); s.prove_gkr(); print("prove_gkr %d: %.2f ms" % (i, 1e3 * (time.perf_counter() - t)), flush=True)
for i in range(3):
    t = time.perf_counter(); _, _, _, sec = s.prove_protocol(); print("prove_protocol %d: %.2f ms  %s" % (i, 1e3 * (time.perf_counter() - t), {k: round(1e3 * v, 2) for k, v in sec.items()}), flush=True)
s.close(); c.close()
