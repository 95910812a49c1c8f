"""Development tool: print the launch plan (nodes, streams, dependencies) of a SHA-256 xBLOCKS proof.  VP_DEBUG=1 python tools/plan_dump.py BLOCKS 2>&1 | grep 'plan node'"""
import gzip, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vp_loader
vp = vp_loader.load()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
pws = os.path.join(tmp, "SHA256_64.pws")
with gzip.open(os.path.join(ROOT, "tests", "golden", "SHA256_64.pws.gz"), "rb") as f, open(pws, "wb") as g:
    g.write(f.read())
c = vp.Circuit.from_pws(pws, int(sys.argv[1]) if len(sys.argv) > 1 else 64, seed=1)
s = vp.Session(c); s.draw_tape()
s.prove_gkr()
