#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (fixture generator, like make_golden.py): runs the oracle; nothing in the product imports this.
One-off: the ORACLE's commitment outputs at BASELINE.json's full size (configs[2]: SHA-256 x1024, input layer 2^23, 65 slices of
2^22 code symbols) — merkle_root_l | merkle_root_h | input_0 | all_sum[65] for the witness of seed 1 and the public vector of
numpy default_rng(8).  Minutes of single-core CPU and tens of GB; the result (1 120 bytes) is committed as
tests/golden/oracle_sha256_x1024_pc.bin and compared with the GPU's in tests/test_gpu_parity.py.

    python tests/golden/make_oracle_fixture_pc.py BLOCKS OUT.bin
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
    n_bits = (oc.layer_size(0) - 1).bit_length()
    print("circuit built, input bits", n_bits, flush=True)
    L.orc_commit_private.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    root_l = ctypes.create_string_buffer(32)
    assert L.orc_commit_private(oc.h, root_l) == 0
    print("commit_private done at %d s" % (time.time() - t0), root_l.raw.hex(), flush=True)
    pub = np.random.default_rng(8).integers(0, P, size=(1 << n_bits, 2), dtype=np.uint64)
    inp = np.zeros((1 << n_bits, 2), dtype=np.uint64)
    L.orc_circuit_inputs(oc.h, inp.ctypes.data)
    e_inner = np.zeros(2, dtype=np.uint64); e_all = np.zeros((65, 2), dtype=np.uint64); root_h = ctypes.create_string_buffer(32)
    L.orc_commit_public.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p]
    assert L.orc_commit_public(inp.ctypes.data, pub.ctypes.data, n_bits, oc.layer_size(0), e_inner.ctypes.data, e_all.ctypes.data, root_h) == 0
    open(out, "wb").write(root_l.raw + root_h.raw + e_inner.tobytes() + e_all.tobytes())
    stop.set()
    print("done: %d s, max RSS %.1f GB" % (time.time() - t0, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6), flush=True)


if __name__ == "__main__":
    main()
