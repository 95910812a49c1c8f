#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (fixture generator, like make_golden.py): runs the oracle; nothing in the product imports this.
One-off, companion of make_oracle_fixture_pc.py: the ORACLE's FRI commit phase at x1024 (17 fold steps over 65 codewords of 2^22
symbols) for the same witness (seed 1) and public vector (default_rng(8)), fold challenges from default_rng(9).  Output:
17 Merkle roots (32 bytes each) then the final codeword (2048 field elements) -> tests/golden/oracle_sha256_x1024_fri.bin.

    python tests/golden/make_oracle_fixture_fri.py BLOCKS OUT.bin
"""
import ctypes, gzip, os, resource, sys, tempfile, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
P = (1 << 61) - 1


def main():
    blocks, out = int(sys.argv[1]), sys.argv[2]
    t0 = time.time()
    stop = threading.Event()
    threading.Thread(target=lambda: [print("... %d s" % (time.time() - t0), flush=True) for _ in iter(lambda: stop.wait(60), True)], daemon=True).start()
    import oracle_binding as ob
    with tempfile.TemporaryDirectory() as tmp:
        p = os.path.join(tmp, "SHA256_64.pws")
        with gzip.open(os.path.join(ROOT, "tests", "golden", "SHA256_64.pws.gz"), "rb") as f, open(p, "wb") as g:
            g.write(f.read())
        oc = ob.Circuit.from_pws(p, blocks, seed=1)
    L = ob.lib()
    nb = (oc.layer_size(0) - 1).bit_length()
    st = nb - 6
    pub = np.random.default_rng(8).integers(0, P, size=(1 << nb, 2), dtype=np.uint64)
    r = np.random.default_rng(9).integers(0, P, size=(st, 2), dtype=np.uint64)
    inp = np.zeros((1 << nb, 2), dtype=np.uint64)
    L.orc_circuit_inputs(oc.h, inp.ctypes.data)
    oc.close()
    roots = ctypes.create_string_buffer(32 * st)
    fin = np.zeros((2048, 2), dtype=np.uint64)
    L.orc_fri_commit.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    print("circuit built at %d s, %d FRI steps" % (time.time() - t0, st), flush=True)
    assert L.orc_fri_commit(inp.ctypes.data, pub.ctypes.data, nb, r.ctypes.data, roots, fin.ctypes.data) == 0
    open(out, "wb").write(roots.raw + fin.tobytes())
    stop.set()
    print("done: %d s, max RSS %.1f GB" % (time.time() - t0, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6), flush=True)


if __name__ == "__main__":
    main()
