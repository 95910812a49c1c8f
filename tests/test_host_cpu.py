"""CPU: the product's host side (loader, subsetInit, verifier replay) and the C-ABI surface.
No compute call into libvpgpu.so is made here (there is no GPU in the build container)."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT


def test_cabi_exports_every_declared_symbol(vp):
    hdr = open(os.path.join(ROOT, "include", "vpgpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(vp_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 18
    out = subprocess.run(["nm", "-D", "--defined-only", vp.LIB_GPU], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = set(re.findall(r" T (vp_[a-z_0-9]+)", out))
    assert [s for s in declared if s not in exported] == []
    lib = vp.lib_gpu()
    for s in declared:
        getattr(lib, s)
    assert b"gfx950" in lib.vp_version()


def test_options_struct_layout_and_library_flavours(vp):
    """vp_options is twelve fields behind struct_size + abi (include/vpgpu.h); the three flavours of the library identify themselves (no GPU needed)."""
    o = vp.Options()
    assert ctypes.sizeof(vp.Options) == 4 * (2 + 12 + 4) == o.struct_size and o.abi == vp.VP_OPTIONS_ABI
    hdr = open(os.path.join(ROOT, "include", "vpgpu.h")).read()
    assert int(re.search(r"#define VP_OPTIONS_ABI (0x[0-9a-f]+)u", hdr).group(1), 16) == vp.VP_OPTIONS_ABI
    body = re.sub(r"/\*.*?\*/", "", hdr[hdr.index("typedef struct {\n    uint32_t struct_size;"):hdr.index("} vp_options;")], flags=re.S)
    fields = re.findall(r"(?<!u)int32_t\s+(\w+)\s*(?:\[\d+\])?;", body)
    assert tuple(f for f in fields if f != "reserved") == vp._PUBLIC_OPTIONS
    assert (o.use_graph, o.plan_autotune, o.real_values, o.real_pairs, o.pc_tensor_pub, o.persistent_rounds, o.poll, o.prefetch_round1, o.interactive_fast_init) == (1,) * 9
    assert o.persistent_timeout_ms == 10000 and o.split_cost_percent == 50 and o.debug == 0 and list(o.reserved) == [0, 0, 0, 0]
    with pytest.raises(TypeError):
        vp.Options(no_such_option=1)
    assert vp.Options(sf3b_grid=448, gkr_path=vp.PATH_LANES).tuning_env() == {"VP_SF3B_GRID": "448", "VP_GKR_PATH": "lanes"}
    lib = vp.lib_gpu()
    assert lib.vp_test_drivers() == 0 and lib.vp_checked_build() == 0
    for path, drv, chk in ((vp.LIB_GPU_TESTDRV, 1, 0), (vp.LIB_GPU_CHECKED, 0, 1)):
        if os.path.exists(path):
            L = ctypes.CDLL(path)
            assert (L.vp_test_drivers(), L.vp_checked_build()) == (drv, chk)


def test_library_contains_gfx950_code_object(vp):
    data = open(vp.LIB_GPU, "rb").read()
    assert b"gfx950" in data and b"k_round_main" in data


def test_no_gpu_fails_loudly(vp):
    from conftest import gpu_count
    if gpu_count() > 0:
        pytest.skip("a GPU is present")
    c = vp.Circuit.randomize(3, 4, seed=3)
    with pytest.raises(RuntimeError):
        vp.Session(c)
    c.close()


def test_product_never_touches_the_oracle():
    for d, _, files in os.walk(os.path.join(ROOT, "virgo-plus_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                assert "vp_oracle" not in txt and "oracle_binding" not in txt and "oracle/" not in txt.replace("oracle/ref_driver.cpp", ""), f
    for f in ("vp_loader.py",):
        assert "oracle" not in open(os.path.join(ROOT, f)).read()


@pytest.mark.parametrize("name,blocks", [("sha256_x1", 1), ("sha256_x16", 16)])
def test_loader_matches_reference_structure(vp, ob, golden, pws_path, name, blocks):
    c = vp.Circuit.from_pws(pws_path, blocks, seed=1)
    g = golden[name]
    assert (c.layers, c.gates) == (g["layers"], g["gates"])
    assert c.hash() == g["circuit_hash"]            # hash printed by the real reference's loader + subsetInit
    oc = ob.Circuit.from_pws(pws_path, blocks, seed=1)
    assert oc.hash() == c.hash()
    c.close(); oc.close()


@pytest.mark.parametrize("blocks", [2, 5, 16, 64])
def test_replicated_builder_equals_the_dag_route(vp, golden, pws_path, blocks, monkeypatch):
    """the block-replicated circuit built from the layered form of ONE block (linear time) must be the
    circuit the loader's own route produces from the DAG of all blocks — same gates, operands, subset ids and witness."""
    fast = vp.Circuit.from_pws(pws_path, blocks, seed=7)
    monkeypatch.setenv("VPH_BUILD", "dag")
    slow = vp.Circuit.from_pws(pws_path, blocks, seed=7)
    monkeypatch.delenv("VPH_BUILD")
    assert (fast.layers, fast.gates) == (slow.layers, slow.gates)
    assert fast.hash() == slow.hash()
    if "sha256_x%d" % blocks in golden:
        g = vp.Circuit.from_pws(pws_path, blocks, seed=1)
        assert g.hash() == golden["sha256_x%d" % blocks]["circuit_hash"]
        g.close()
    fast.close(); slow.close()


def test_randomize_matches_reference_structure(vp, golden):
    c = vp.Circuit.randomize(8, 12, seed=1)
    assert c.hash() == golden["randomize_8_12"]["circuit_hash"]
    c.close()


def test_loader_rejects_bad_input(vp, tmp_path):
    p = tmp_path / "bad.pws"
    p.write_text("P V0 = I0 E\nP V1 = V0 FOO V0 E\n")
    with pytest.raises(RuntimeError):
        vp.Circuit.from_pws(str(p))
    with pytest.raises(RuntimeError):
        vp.Circuit.from_pws(str(tmp_path / "missing.pws"))


@pytest.mark.parametrize("name,blocks", [("sha256_x1", 1), ("sha256_x16", 16)])
def test_host_verifier_replay_accepts_golden(vp, gold_gkr, pws_path, name, blocks):
    """The host verifier (tape order, sumcheck/Liu checks, wiring predicates, input check) accepts the
    real reference's messages and rejects a tampered copy."""
    c = vp.Circuit.from_pws(pws_path, blocks, seed=1)
    tr = gold_gkr(name)
    assert c.verify_transcript(tr)
    import numpy as np
    w = np.frombuffer(tr, dtype=np.uint64).copy()
    w[9] += np.uint64((1 << 61) - 1)               # non-canonical encoding of the same element: rejected, not reduced
    assert not c.verify_transcript(w.tobytes())
    bad = bytearray(tr)
    bad[len(bad) // 2] ^= 1
    assert not c.verify_transcript(bytes(bad))
    assert not c.verify_transcript(tr[:-16])
    c.close()


def test_host_verifier_replay_randomize(vp, gold_gkr):
    c = vp.Circuit.randomize(8, 12, seed=1)
    assert c.verify_transcript(gold_gkr("randomize_8_12"))
    c.close()


def test_custom_circuit_structures_agree(vp, ob):
    import custom_circuits as cc
    args = cc.make(11, [30, 20, 25, 7])
    c = vp.Circuit.custom(*args)
    oc = ob.Circuit.custom(*args)
    assert c.hash() == oc.hash()
    tr, st = oc.prove_gkr()
    assert st["verified"] == 1
    assert c.verify_transcript(tr)          # host verifier (incl. assert-gate predicate) accepts the oracle's proof
    c.close(); oc.close()


def test_host_sha3_is_fips202(vp):
    import hashlib
    import numpy as np
    rng = np.random.default_rng(12)
    n = 200
    msgs = rng.integers(0, 256, size=(n, 64), dtype=np.uint8)
    msgs[0] = 0
    out = np.zeros((n, 32), dtype=np.uint8)
    vp.lib_host().vph_test_sha3(msgs.ctypes.data, out.ctypes.data, n)
    for i in range(n):
        assert out[i].tobytes() == hashlib.sha3_256(msgs[i].tobytes()).digest()


def test_fiat_shamir_verifier_on_host(vp, golden, gold_gkr):
    """vph_verify_fs needs no GPU: the committed proof (made on the GPU box by tests/test_gpu_parity.py with VP_WRITE_FS_FIXTURE) is
    accepted, tampered / truncated / foreign proofs are rejected, and so is a transcript of the interactive protocol (its
    challenges came from random(), not from the hash chain)."""
    import os
    from conftest import GOLDEN
    c = vp.Circuit.randomize(6, 8, seed=5)
    proof = open(os.path.join(GOLDEN, "fs_proof_randomize_6_8_seed5.bin"), "rb").read()
    assert c.verify_fs(proof)
    for pos in (0, 5, len(proof) // 2, len(proof) - 1):
        bad = bytearray(proof); bad[pos] ^= 0x80
        assert not c.verify_fs(bytes(bad))
    assert not c.verify_fs(proof[:-16]) and not c.verify_fs(b"")
    # a limb shifted by p is the same field element in a non-canonical encoding: algebraically consistent, still rejected
    import numpy as np
    P = (1 << 61) - 1
    for limb in (0, 1, 7, len(proof) // 8 - 1):
        w = np.frombuffer(proof, dtype=np.uint64).copy()
        w[limb] += np.uint64(P)
        assert not c.verify_fs(w.tobytes()), "non-canonical limb %d accepted" % limb
    other = vp.Circuit.randomize(6, 8, seed=6)
    assert not other.verify_fs(proof)
    g = vp.Circuit.randomize(8, 12, seed=1)
    assert g.verify_transcript(gold_gkr("randomize_8_12"))
    assert not g.verify_fs(gold_gkr("randomize_8_12"))
    c.close(); other.close(); g.close()
