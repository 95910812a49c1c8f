#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (fixture generator, like make_golden.py): runs the oracle; nothing in the product imports this.
The ORACLE's complete protocol at BASELINE.json's full size (configs[2]/[3]: SHA-256 x1024) for witness seed SEED, exactly as the
reference's verifier::verify() runs it: commit_private, the GKR part, commit_public on the PROTOCOL's public vector eq(r_liu, .)
(src/verifier.cpp:368-379), and — with --fri — the FRI commit phase on the challenges the reference's verifier would draw: the
glibc stream continued past the draws fft_gkr consumes (oracle/vp_oracle.h: orc_fft_gkr_draws, pinned against the real reference).

    python tests/golden/make_oracle_fixture_full.py BLOCKS SEED OUT.bin [--fri | --gkr-only]

OUT.bin = merkle_root_l | GKR slice | merkle_root_h | input_0 | all_sum[65]  (the golden layout of SURVEY.md §8c)
          [--fri: | (steps x 32-byte roots) | final codeword (2048 x 16 bytes) | fold challenges (steps x 16 bytes)]
--gkr-only writes the GKR slice alone (no commitment; a tenth of the time).  OUT.bin.json records sizes, counters and timings.
Minutes of single-core CPU and ~25 GB of memory at x1024."""
import ctypes, gzip, json, os, resource, sys, tempfile, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    blocks, seed, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    fri = "--fri" in sys.argv[4:]
    gkr_only = "--gkr-only" in sys.argv[4:]
    t0 = time.time()
    stop = threading.Event()
    threading.Thread(target=lambda: [print("... %d s" % (time.time() - t0), flush=True) for _ in iter(lambda: stop.wait(60), True)], daemon=True).start()
    import oracle_binding as ob
    with tempfile.TemporaryDirectory() as tmp:
        p = os.path.join(tmp, "SHA256_64.pws")
        with gzip.open(os.path.join(ROOT, "tests", "golden", "SHA256_64.pws.gz"), "rb") as f, open(p, "wb") as g:
            g.write(f.read())
        oc = ob.Circuit.from_pws(p, blocks, seed=seed)
    L = ob.lib()
    info = {"blocks": blocks, "seed": seed, "circuit_hash": oc.hash(), "build_sec": time.time() - t0}
    print("circuit built at %d s" % (time.time() - t0), flush=True)
    if gkr_only:
        tr, st = oc.prove_gkr()
        open(out, "wb").write(tr)
        info.update(st); info.update(bytes=len(tr), layout="gkr slice")
    else:
        L.orc_prove_full.restype = ctypes.c_int64
        L.orc_prove_full.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        buf = ctypes.create_string_buffer(1 << 20)
        st = ob.Stats()
        n = L.orc_prove_full(oc.h, buf, len(buf), ctypes.byref(st))
        assert n > 0
        full = buf.raw[:n]
        info.update(st.as_dict()); info.update(bytes=n, layout="root_l | gkr | root_h | input_0 | all_sum[65]", full_sec=time.time() - t0)
        print("full transcript (%d bytes) at %d s" % (n, time.time() - t0), flush=True)
        extra = b""
        if fri:
            nb = L.orc_circuit_layer_bitlen(oc.h, 0)
            steps = nb - 6
            L.orc_f_random_next.argtypes = [ctypes.c_int, ctypes.c_void_p]
            skip = L.orc_fft_gkr_draws(steps)
            nxt = np.zeros((skip + steps, 2), np.uint64)
            L.orc_f_random_next(skip + steps, nxt.ctypes.data)            # the stream continues where the GKR part left it
            r = np.ascontiguousarray(nxt[skip:])
            L.orc_last_point.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
            pt = np.zeros((nb, 2), np.uint64)
            assert L.orc_last_point(oc.h, pt.ctypes.data, nb) == 0
            one = np.array([1, 0], np.uint64)
            pub = np.zeros((1 << nb, 2), np.uint64)
            L.orc_beta_table(pt.ctypes.data, nb, one.ctypes.data, pub.ctypes.data)
            inp = np.zeros((1 << nb, 2), np.uint64)
            L.orc_circuit_inputs(oc.h, inp.ctypes.data)
            oc.close()
            roots = ctypes.create_string_buffer(32 * steps)
            fin = np.zeros((2048, 2), np.uint64)
            L.orc_fri_commit.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
            assert L.orc_fri_commit(inp.ctypes.data, pub.ctypes.data, nb, r.ctypes.data, roots, fin.ctypes.data) == 0
            extra = roots.raw + fin.tobytes() + r.tobytes()
            info.update(fri_steps=steps, fft_gkr_draws_skipped=skip, layout=info["layout"] + " | fri roots | final codeword | fold challenges")
        open(out, "wb").write(full + extra)
    stop.set()
    info.update(total_sec=time.time() - t0, max_rss_gb=resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6)
    json.dump(info, open(out + ".json", "w"), indent=1)
    print("done:", info, flush=True)


if __name__ == "__main__":
    main()
