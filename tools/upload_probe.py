"""Development probe: where a Session's set-up time goes (flatten on the host, vp_circuit_upload, vp_evaluate), x64 and x1024.
   VP_DEBUG_UPLOAD=1 python tools/upload_probe.py [blocks ...]"""
import gzip, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vp_loader
vp = vp_loader.load()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
pws = os.path.join(tmp, "SHA256_64.pws")
with gzip.open(os.path.join(ROOT, "tests", "golden", "SHA256_64.pws.gz"), "rb") as f, open(pws, "wb") as g:
    g.write(f.read())
for blocks in [int(x) for x in sys.argv[1:]] or [64, 64, 1024]:
    t0 = time.perf_counter()
    c = vp.Circuit.from_pws(pws, blocks, seed=1)
    t1 = time.perf_counter()
    s = vp.Session(c)
    t2 = time.perf_counter()
    print("x%d: circuit build %.3f s, session (flatten + upload + evaluate) %.3f s" % (blocks, t1 - t0, t2 - t1), flush=True)
    s.close(); c.close()
