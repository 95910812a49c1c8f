#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (fixture generator, like make_golden.py): runs the oracle; nothing in the product imports this.
One-off parity check at BASELINE.json's full size (configs[2]: SHA-256 x1024, 102 M gates), too slow for the test suite:
the oracle's CPU proof (a few minutes, tens of GB) against the GPU's batched and sharded proofs.

    python tests/golden/make_oracle_fixture_gkr.py oracle BLOCKS OUT.bin     # CPU only (run under `ulimit -v` to bound memory)
    python tests/golden/make_oracle_fixture_gkr.py gpu BLOCKS OUT.bin        # compares the GPU transcript with OUT.bin
BLOCKS may be "rLxE" for layeredCircuit::randomize(L, E) (configs[4]: r16x20 = 2^24 gates) instead of a SHA-256 block count.
"""
import gzip, json, os, resource, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def pws(tmp):
    p = os.path.join(tmp, "SHA256_64.pws")
    with gzip.open(os.path.join(ROOT, "tests", "golden", "SHA256_64.pws.gz"), "rb") as f, open(p, "wb") as g:
        g.write(f.read())
    return p


def main():
    mode, out = sys.argv[1], sys.argv[3]
    rnd = None
    if sys.argv[2].startswith("r"):
        rnd = tuple(int(x) for x in sys.argv[2][1:].split("x")); blocks = 0
    else:
        blocks = int(sys.argv[2])
    with tempfile.TemporaryDirectory() as tmp:
        path = pws(tmp)
        t0 = time.time()
        if mode == "oracle":
            import oracle_binding as ob
            oc = ob.Circuit.randomize(rnd[0], rnd[1], seed=1) if rnd else ob.Circuit.from_pws(path, blocks, seed=1)
            t1 = time.time()
            gold, st = oc.prove_gkr()
            open(out, "wb").write(gold)
            st = dict(st); st.update(circuit_hash=oc.hash(), build_sec=t1 - t0, total_sec=time.time() - t0,
                                     max_rss_gb=resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6)
            json.dump(st, open(out + ".json", "w"))
            print("oracle %s:" % sys.argv[2], st)
        else:
            import vp_loader
            vp = vp_loader.load()
            gold = open(out, "rb").read()
            st = json.load(open(out + ".json"))
            c = vp.Circuit.randomize(rnd[0], rnd[1], seed=1) if rnd else vp.Circuit.from_pws(path, blocks, seed=1)
            assert c.hash() == st["circuit_hash"], "levelised circuit differs from the oracle's"
            s = vp.Session(c)
            s.draw_tape()
            tr, res = s.prove_gkr()
            ok = tr == gold
            parts = []
            for r in range(8):
                s.set_shard(r, 8)
                parts.append(s.prove_gkr()[0])
            s.set_shard(0, 1)
            ok8 = vp.sum_transcripts(parts) == gold
            print(json.dumps({"case": sys.argv[2], "gates": c.gates, "transcript_bytes": len(tr), "gpu_equals_oracle": ok,
                              "sharded_x8_equals_oracle": ok8, "rounds": res["rounds"], "oracle_rounds": st["rounds"],
                              "oracle_prove_sec": st["prove_sec"], "gpu_device_ms": res["gkr_device_ms"],
                              "oracle_field_ops": st["mult_count"] + st["add_count"], "oracle_max_rss_gb": st["max_rss_gb"]}))
            sys.exit(0 if ok and ok8 else 1)


if __name__ == "__main__":
    main()
