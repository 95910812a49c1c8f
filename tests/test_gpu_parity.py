"""GPU (-m gpu): the HIP path, called through the C ABI (libvpgpu.so via libvphost.so), against the
oracle on the same seeded inputs and against the golden transcripts of the real reference.
Bar: bit-exact (all arithmetic is integer mod 2^61-1)."""
import ctypes
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN as GOLDEN_DIR

pytestmark = pytest.mark.gpu
P = (1 << 61) - 1
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx(vp):
    lib = vp.lib_gpu()
    h = ctypes.c_void_p()
    rc = lib.vp_create(0, ctypes.byref(h))
    assert rc == 0, "vp_create failed: the HIP extension must run on the GPU box"
    yield h
    lib.vp_destroy(h)


def rand_f(rng, n):
    a = rng.integers(0, P, size=(n, 2), dtype=np.uint64)
    edge = np.array([[0, 0], [1, 0], [0, 1], [P - 1, P - 1], [P - 1, 0], [0, P - 1], [1 << 60, (1 << 60) + 5]], dtype=np.uint64)
    a[: len(edge)] = edge
    return a


@pytest.mark.parametrize("op,name", [(0, "orc_f_add"), (1, "orc_f_sub"), (2, "orc_f_mul")])
def test_field_ops(vp, ob, ctx, op, name):
    rng = np.random.default_rng(op + 10)
    n = 4099
    a, b = rand_f(rng, n), np.roll(rand_f(rng, n), 3, axis=0)
    out = np.zeros_like(a)
    assert vp.lib_gpu().vp_test_field(ctx, op, a.ctypes.data, b.ctypes.data, out.ctypes.data, n) == 0
    exp = np.zeros_like(a)
    f = getattr(ob.lib(), name)
    for i in range(n):
        f(a[i].ctypes.data, b[i].ctypes.data, exp[i].ctypes.data)
    assert np.array_equal(out, exp)
    assert vp.lib_gpu().vp_test_field(ctx, op, a.ctypes.data, b.ctypes.data, out.ctypes.data, 0) == 0   # empty input


def test_field_dot2_one_reduction_form(vp, ob, ctx):
    """f_dot2cc (vp_field.h): a b + c d of canonical operands as ONE sum of split products per limb — the form k_fri_fold0_vo3 computes the virtual oracle with.
    Against the oracle's f_mul / f_add on random values and on the extremes (0, 1, p - 1 in either limb: the negated imaginary parts p - x reach p itself)."""
    rng = np.random.default_rng(77)
    n = 4099
    a, b = rand_f(rng, n), np.roll(rand_f(rng, n), 3, axis=0)
    ext = np.array([[P - 1, P - 1], [P - 1, P - 1], [0, 0], [P - 1, 0], [0, P - 1], [1, 0], [0, 1], [P - 1, P - 1]], dtype=np.uint64)
    a[100:100 + len(ext)] = ext
    b[100:100 + len(ext)] = ext[::-1]
    out = np.zeros_like(a)
    assert vp.lib_gpu().vp_test_field(ctx, 3, a.ctypes.data, b.ctypes.data, out.ctypes.data, n) == 0
    L = ob.lib()
    prod = np.zeros_like(a)
    for i in range(n):
        L.orc_f_mul(a[i].ctypes.data, b[i].ctypes.data, prod[i].ctypes.data)
    exp = np.zeros_like(a)
    for i in range(n):
        L.orc_f_add(prod[i].ctypes.data, prod[(i + 1) % n].ctypes.data, exp[i].ctypes.data)
    assert np.array_equal(out, exp)
    assert int(out.max()) < P


@pytest.mark.parametrize("n", [0, 1, 2, 3, 7, 12, 17])
def test_beta_table(vp, ob, ctx, n):
    rng = np.random.default_rng(n)
    r = rng.integers(0, P, size=(max(n, 1), 2), dtype=np.uint64)
    for init in (np.array([1, 0], dtype=np.uint64), rng.integers(0, P, size=2, dtype=np.uint64)):
        out = np.zeros((1 << n, 2), dtype=np.uint64)
        exp = np.zeros_like(out)
        assert vp.lib_gpu().vp_test_beta(ctx, r.ctypes.data, n, init.ctypes.data, out.ctypes.data) == 0
        ob.lib().orc_beta_table(r.ctypes.data, n, init.ctypes.data, exp.ctypes.data)
        assert np.array_equal(out, exp)


def _both_modes(vp, c, gold):
    s = vp.Session(c)
    tr, res, ok = s.prove_interactive()
    assert ok, "host verifier rejected the GPU proof"
    assert tr == gold, "interactive transcript differs"
    s.draw_tape()
    tr2, res2 = s.prove_gkr()
    assert tr2 == gold, "batched transcript differs"
    ok2, _ = s.check(tr2)
    assert ok2
    tr3, _ = s.prove_gkr()            # idempotent: a second pass over the same resident state
    assert tr3 == gold
    s.close()
    # the alternative batched drivers (per-round launches; shuffle-fold + single-CU tail; one stream per sumcheck chain
    # instead of the batched launch plan) must agree
    import os
    for path in (("simple", "lanes") if vp.lib_gpu().vp_test_drivers() else ()):      # -DVP_TEST_DRIVERS flavour only (test_cross_check_drivers_flavour runs these tests under it)
        os.environ["VP_GKR_PATH"] = path
        try:
            s2 = vp.Session(c)
            s2.draw_tape()
            tr4, res4 = s2.prove_gkr()
            s2.close()
        finally:
            del os.environ["VP_GKR_PATH"]
        assert tr4 == gold, path + " batched transcript differs"
    return res, res2


@pytest.mark.parametrize("name,blocks", [("sha256_x1", 1), ("sha256_x16", 16)])
def test_sha256_transcript_matches_reference(vp, golden, gold_gkr, pws_path, name, blocks):
    c = vp.Circuit.from_pws(pws_path, blocks, seed=1)
    assert c.hash() == golden[name]["circuit_hash"]
    res, res2 = _both_modes(vp, c, gold_gkr(name))
    assert res["rounds"] == golden[name]["rounds"] == res2["rounds"]
    c.close()


@pytest.mark.parametrize("name,blocks", [("sha256_x16", 16), ("sha256_x64", 64)])
def test_fused_init_plan_matches_reference(vp, golden, gold_gkr, pws_path, name, blocks, monkeypatch):
    """The plan variant that runs the phase-1 / Liu init inside the first fold launch (GenP1 / GenLiu; by default only for
    tables of >= 2^23 entries) forced on for every table the fold kernel handles: same golden transcript."""
    monkeypatch.setenv("VP_FUSE_MIN_LOG", "17")
    c = vp.Circuit.from_pws(pws_path, blocks, seed=1)
    s = vp.Session(c)
    s.draw_tape()
    tr, res = s.prove_gkr()
    assert tr == gold_gkr(name)
    monkeypatch.setenv("VP_FUSE_DOT", "1")          # V_u riding on the fused launch instead of the separate pass
    s2 = vp.Session(c)
    s2.draw_tape()
    tr2, _ = s2.prove_gkr()
    assert tr2 == tr
    monkeypatch.delenv("VP_FUSE_MIN_LOG"); monkeypatch.delenv("VP_FUSE_DOT")
    monkeypatch.setenv("VP_DROP_Y", "0")            # every round sums m1 v1 + a1 itself instead of deriving b from the previous claim
    s4 = vp.Session(c)
    s4.draw_tape()
    tr4, _ = s4.prove_gkr()
    assert tr4 == tr
    monkeypatch.delenv("VP_DROP_Y")
    monkeypatch.setenv("VP_DROP_Y1", "1")           # round 1 leaves its b to the fix-up pass over the finished transcript (k_fixup)
    s6 = vp.Session(c)
    s6.draw_tape()
    tr6, res6 = s6.prove_gkr()
    assert tr6 == tr
    s.close(); s2.close(); s4.close(); s6.close(); c.close()


def _sharded_parts(vp, s, world):
    """The transcripts of all `world` ranks of a chain-sharded proof, produced one after the other on this GPU."""
    import numpy as np
    parts = []
    for r in range(world):
        s.set_shard(r, world)
        tr, res = s.prove_gkr()
        tr2, _ = s.prove_gkr()            # graph replay of the shard's plan
        assert tr2 == tr
        parts.append(tr)
    s.set_shard(0, 1)
    nz = [np.frombuffer(p, dtype=np.uint64) != 0 for p in parts]
    for a in range(world):
        for b in range(a + 1, world):
            assert not (nz[a] & nz[b]).any(), "ranks %d and %d wrote the same transcript slot" % (a, b)
    return parts


@pytest.mark.parametrize("name,blocks,worlds", [("sha256_x1", 1, (2, 8)), ("sha256_x16", 16, (2, 3, 4, 8)), ("sha256_x64", 64, (2, 8))])
def test_chain_sharded_proof_assembles_to_reference(vp, golden, gold_gkr, pws_path, name, blocks, worlds):
    """One proof over `world` GPUs (independent sumcheck instances): every rank runs the sumcheck chains dealt to it,
    the other transcript slots stay zero, and the u64 sum of the ranks' transcripts (= the one RCCL all-reduce) is the
    reference's transcript, byte for byte.  All ranks are executed in turn on this one GPU."""
    c = vp.Circuit.from_pws(pws_path, blocks, seed=1)
    s = vp.Session(c)
    s.draw_tape()
    gold = gold_gkr(name)
    for world in worlds:
        s.set_shard(0, world)
        owner, cost = s.shard_chains()
        assert len(owner) == 3 * (c.layers - 1) + 1
        assert set(owner[cost > 0]) == set(range(min(world, int((cost > 0).sum())))), "a rank has no work"
        parts = _sharded_parts(vp, s, world)
        assert vp.sum_transcripts(parts) == gold, "world %d" % world
    tr, _ = s.prove_gkr()               # back to the unsharded proof
    assert tr == gold
    s.close(); c.close()


def test_chain_sharded_proof_randomize_and_fused_init(vp, golden, gold_gkr, monkeypatch):
    """Sharding with a phase 2 that takes V_u from its own inner-product pass (phase 1 of the layer on another rank), on the
    synthetic circuit, with the init fused into the first fold launch forced on."""
    monkeypatch.setenv("VP_FUSE_MIN_LOG", "10")
    c = vp.Circuit.randomize(8, 12, seed=1)
    s = vp.Session(c)
    s.draw_tape()
    for world in (2, 5, 8):
        assert vp.sum_transcripts(_sharded_parts(vp, s, world)) == gold_gkr("randomize_8_12")
    s.close(); c.close()


@pytest.mark.parametrize("ge", [1, 2, 3])
def test_every_explicit_graph_form_on_sharded_plans(vp, golden, gold_gkr, pws_path, ge):
    """The plan tuner picks among the graph forms (vp_options.graph_explicit 0-3) on unsharded plans only; a caller may pin any of them, so each explicit
    form is instantiated and replayed here on the plans the tuner never sees: the chain shards of 3 ranks and the index-split shards of 2 (every rank in
    turn on this GPU, graph replayed twice), assembling to the real reference's transcript."""
    c = vp.Circuit.from_pws(pws_path, 16, seed=1)
    s = vp.Session(c, options=vp.Options(graph_explicit=ge, plan_autotune=0))
    s.draw_tape()
    gold = gold_gkr("sha256_x16")
    assert vp.sum_transcripts(_sharded_parts(vp, s, 3)) == gold
    assert _split_parts(vp, s, 2) == gold
    tr, _ = s.prove_gkr()
    assert tr == gold and s.options_in_effect().graph_explicit == ge
    s.close(); c.close()


def _split_parts(vp, s, world, min_log=11, exchange=True, vu_info=None):
    """The outputs (partial transcript + export area) of all ranks of an index-split proof, one after the other on this GPU.  exchange: V_u of the
    split phase-2 chains from the ranks' partial inner products (vp_shard_vu_partials -> u64 sum -> vp_shard_vu_set), as a W-rank caller with its
    own transport does it; the replay after it has no sums handed in and adds up the whole layer itself — both must give the same bytes."""
    import numpy as np
    sums = None
    if exchange:
        vus = []
        for r in range(world):
            s.set_shard(r, world)
            s.set_shard_split(min_log)
            vus.append(s.shard_vu_partials())
        assert len({v.shape for v in vus}) == 1, "the ranks disagree on the number of split phase-2 chains"
        sums = np.sum(np.stack(vus), axis=0, dtype=np.uint64)
        if vu_info is not None:
            vu_info.append((world, vus, sums))
    parts = []
    for r in range(world):
        s.set_shard(r, world)
        s.set_shard_split(min_log)
        if sums is not None and len(sums):
            s.shard_vu_set(sums)
        tr, _ = s.prove_gkr()
        tr2, _ = s.prove_gkr()            # graph replay of the rank's plan (whole inner products: nothing handed in)
        assert tr2 == tr
        parts.append(tr)
    out = s.shard_finish(vp.sum_transcripts(parts))
    s.set_shard(0, 1)
    return out


@pytest.mark.parametrize("name,blocks,worlds", [("sha256_x16", 16, (2, 3, 8)), ("sha256_x64", 64, (8,))])
def test_index_split_proof_assembles_to_reference(vp, golden, gold_gkr, pws_path, name, blocks, worlds, monkeypatch):
    """One proof over W ranks with the long tables cut by index (vp_set_shard_split): slice s of every table of >= 2^(log2 W + 11)
    entries is folded by rank s, the last log2 W rounds of those tables are finished on the host from the exported entries, short
    tables and short chains are dealt out whole.  The u64 sum of the ranks' outputs, finished, is the real reference's transcript —
    W = 3 covers a world size that is not a power of two (two slices, the third rank takes whole chains only).  Both policies: every
    chain with a long table cut (split_cost_percent = 0) and only the chains above half a rank's fair share (the default)."""
    c = vp.Circuit.from_pws(pws_path, blocks, seed=1)
    gold = gold_gkr(name)
    s = vp.Session(c)                      # default policy
    s.draw_tape()
    for w in worlds:
        assert _split_parts(vp, s, w) == gold, "default policy, W=%d" % w
    s.close()
    monkeypatch.setenv("VP_SPLIT_COST_PERCENT", "0")
    s = vp.Session(c)
    s.draw_tape()
    gold = gold_gkr(name)
    info = []
    for w in worlds:
        s.set_shard(0, w); s.set_shard_split(11)
        owner, _ = s.shard_chains()
        assert (owner == -1).any(), "nothing was split"
        assert _split_parts(vp, s, w, vu_info=info) == gold, "W=%d" % w
        assert _split_parts(vp, s, w, exchange=False) == gold, "W=%d, every rank adds up the whole layer" % w
    # the exchange is a real one: there are split phase-2 chains, and no single rank's share is the sum
    P = (1 << 61) - 1
    for w, vus, sums in info:
        assert len(sums) > 0, "no split phase-2 chain at W=%d" % w
        assert all((v % P != sums % P).any() for v in vus), "a rank's partial inner products already are V_u (W=%d)" % w
    tr, _ = s.prove_gkr()                  # back to the unsharded proof on the same context
    assert tr == gold
    s.close(); c.close()


def test_index_split_vu_exchange_refusals(vp, pws_path, monkeypatch):
    """vp_shard_vu_set takes exactly the number of split phase-2 chains; sums taken on one tape are refused by a proof on another (they would give a
    transcript that is wrong without any sign of it); an unsplit context has nothing to exchange."""
    import numpy as np
    monkeypatch.setenv("VP_SPLIT_COST_PERCENT", "0")
    c = vp.Circuit.from_pws(pws_path, 16, seed=1)
    s = vp.Session(c)
    s.draw_tape()
    assert len(s.shard_vu_partials()) == 0                    # not sharded: nothing
    s.set_shard(0, 2); s.set_shard_split(11)
    part = s.shard_vu_partials()
    assert len(part) > 0
    with pytest.raises(RuntimeError):
        s.shard_vu_set(np.zeros((len(part) + 1, 2), dtype=np.uint64))
    s.shard_vu_set(part)                                      # (not the sum: never used — the tape changes first)
    s.draw_tape()                                             # same draws again = the same tape: accepted
    s.prove_gkr()
    s.shard_vu_set(part)
    ctx = vp.lib_host().vph_session_ctx(s.h)
    n_tape = ctypes.c_uint64(0); n_tr = ctypes.c_uint64(0)
    assert vp.lib_gpu().vp_gkr_sizes(ctx, ctypes.byref(n_tape), ctypes.byref(n_tr)) == 0
    other = np.ones((n_tape.value, 2), dtype=np.uint64)
    out = ctypes.create_string_buffer(int(n_tr.value) + 4096)
    w = ctypes.c_uint64(0)
    vp.lib_gpu().vp_prove_gkr.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
    rc = vp.lib_gpu().vp_prove_gkr(ctx, other.ctypes.data, n_tape.value, out, len(out), ctypes.byref(w))
    assert rc != 0 and b"another tape" in vp.lib_gpu().vp_last_error(ctx)
    s.set_shard(0, 1)
    s.close(); c.close()


def test_index_split_multi_table_chains_and_complex_values(vp, ob, monkeypatch):
    """Phase-2 chains with several long tables, short tables riding with one rank, complex circuit values, assert gates: random circuits
    with every gate type (custom_circuits) and the reference's `randomize`, W = 2, 4, 8 against the oracle."""
    import custom_circuits as cc
    monkeypatch.setenv("VP_SPLIT_COST_PERCENT", "0")
    for what, mk in (("custom", lambda m: m.Circuit.custom(*cc.make(7, [300000, 280000, 150000, 270000, 9000]))),
                     ("randomize", lambda m: m.Circuit.randomize(6, 16, seed=3))):
        c = mk(vp); oc = mk(ob)
        assert c.hash() == oc.hash()
        gold, st = oc.prove_gkr()
        assert st["verified"] == 1
        s = vp.Session(c)
        s.draw_tape()
        for w in (2, 4, 8):
            assert _split_parts(vp, s, w) == gold, "%s W=%d" % (what, w)
        s.close(); c.close(); oc.close()


def test_set_shard_rejects_bad_arguments_and_survives_idle_ranks(vp, ob):
    c = vp.Circuit.randomize(4, 8, seed=7)
    s = vp.Session(c)
    for rank, world in ((-1, 2), (2, 2), (0, 0)):
        with pytest.raises(RuntimeError):
            s.set_shard(rank, world)
    # more ranks than chains (3 * 3 + 1 = 10): the surplus ranks prove nothing and contribute zeros
    s.draw_tape()
    gold, _ = ob.Circuit.randomize(4, 8, seed=7).prove_gkr()
    parts = _sharded_parts(vp, s, 16)
    assert sum(1 for p in parts if any(p)) <= 10
    assert vp.sum_transcripts(parts) == gold
    s.close(); c.close()


@pytest.mark.parametrize("make", ["randomize", "sha256_x1", "sha256_x16"])
def test_fiat_shamir_proof_round_trip(vp, gold_gkr, pws_path, make):
    """Fiat-Shamir mode: the challenges are SHA3-derived from the circuit hash and the prover's earlier messages, the
    prover runs through its interactive entry points, and the host verifier re-derives everything from the proof alone.  A
    separate mode: not comparable with the reference's transcripts, so the properties are accept / reject."""
    import os
    if make == "randomize":
        c = vp.Circuit.randomize(6, 8, seed=5)
        other = vp.Circuit.randomize(6, 8, seed=6)
    else:
        blocks = 1 if make == "sha256_x1" else 16
        c = vp.Circuit.from_pws(pws_path, blocks, seed=1)
        other = vp.Circuit.from_pws(pws_path, blocks, seed=2)           # same gates, another witness
    s = vp.Session(c)
    proof, res, ok = s.prove_fs()
    assert ok and res["rounds"] > 0
    assert c.verify_fs(proof)
    proof2, _, _ = s.prove_fs()
    assert proof2 == proof, "deterministic"
    tr_i, _, _ = s.prove_interactive()
    assert len(tr_i) == len(proof) and tr_i != proof, "same layout, different challenges"
    if make != "randomize":
        assert proof != gold_gkr(make)
    assert not other.verify_fs(proof), "a proof is bound to its circuit and witness"
    n = len(proof)
    for pos in (0, 16, 64, n // 3, n // 2, n - 17, n - 1):
        bad = bytearray(proof); bad[pos] ^= 1
        assert not c.verify_fs(bytes(bad)), "tampered byte %d accepted" % pos
    assert not c.verify_fs(proof[:-16]) and not c.verify_fs(proof + bytes(16)) and not c.verify_fs(b"")
    P = (1 << 61) - 1
    for limb in (0, 3, n // 16, n // 8 - 1):        # same field element, non-canonical encoding (limb + p)
        w = np.frombuffer(proof, dtype=np.uint64).copy()
        w[limb] += np.uint64(P)
        assert not c.verify_fs(w.tobytes()), "non-canonical limb %d accepted" % limb
    if make == "randomize" and os.environ.get("VP_WRITE_FS_FIXTURE"):
        open(os.environ["VP_WRITE_FS_FIXTURE"], "wb").write(proof)
    s.close(); c.close(); other.close()


@pytest.mark.parametrize("make", ["randomize_7_9", "custom", "sha256_x1", "ragged"])
def test_fiat_shamir_proof_bytes_equal_the_oracles(vp, ob, pws_path, make):
    """The Fiat-Shamir mode against an independent implementation: the oracle (own SHA3 sponge, own statement serialiser, own CPU prover,
    oracle/vp_oracle.cpp::orc_prove_fs) must produce the SAME proof bytes as host/verifier.cpp::proveFS driving the device through
    vp_round — every challenge depends on every earlier message, so equality pins the challenge derivation, the absorb order and all
    field elements at once.  Circuits: synthetic, every gate type + assert gates + complex constants, SHA-256, and a ragged one with
    single-entry tables and zero-round phases."""
    import custom_circuits as cc
    if make == "randomize_7_9":
        c, oc = vp.Circuit.randomize(7, 9, seed=11), ob.Circuit.randomize(7, 9, seed=11)
    elif make == "custom":
        args = cc.make(77, [700, 300, 129, 257, 64, 5])
        c, oc = vp.Circuit.custom(*args), ob.Circuit.custom(*args)
    elif make == "ragged":
        args = cc.make(78, [9, 1, 3, 1, 17, 2])
        c, oc = vp.Circuit.custom(*args), ob.Circuit.custom(*args)
    else:
        c, oc = vp.Circuit.from_pws(pws_path, 1, seed=3), ob.Circuit.from_pws(pws_path, 1, seed=3)
    assert c.hash() == oc.hash()
    gold, st = oc.prove_fs()
    assert st["verified"] == 1
    s = vp.Session(c)
    proof, res, ok = s.prove_fs()
    assert ok
    assert proof == gold
    assert c.verify_fs(gold)
    s.close(); c.close(); oc.close()


@pytest.mark.parametrize("lg", [1, 2, 3, 4, 5, 7, 10, 11, 12, 13, 17])
def test_fft_gkr_vs_reference_record_and_oracle(vp, ob, ctx, lg):
    """lib/virgo's fft_gkr (fft_circuit_GKR.cpp:833-849) on the device, vp_fft_gkr: the circuit's 64 outputs, every round
    polynomial of its 2 + 2 lg sumchecks and every claimed table value, bit-exact against the REAL reference's record of the same call
    (tests/golden/fftgkr_lg{7,13,17}.bin: `ref_run --fft-gkr LG --dump-fft`, taken at link time, lg = 7 / 13 / 17 are the x1 / x64 / x1024
    commitments) and against the oracle's restatement at every size, including the degenerate ones (one-round sumchecks).  The tape is the
    reference's own draw sequence after F::init()."""
    lib = vp.lib_gpu()
    lib.vp_fft_gkr_sizes.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.vp_fft_gkr.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
    nt, nm = ctypes.c_uint64(0), ctypes.c_uint64(0)
    assert lib.vp_fft_gkr_sizes(lg, ctypes.byref(nt), ctypes.byref(nm)) == 0
    assert nt.value == 2 * lg * lg + 9 * lg + 96 and nm.value == 64 + 3 * (2 * lg * lg + 2 * lg + 6) + 2 + 2 * lg
    tape = np.zeros((nt.value, 2), dtype=np.uint64)
    ob.lib().orc_f_random_seq(3396, int(nt.value), tape.ctypes.data)
    msgs = np.zeros((nm.value, 2), dtype=np.uint64)
    nw = ctypes.c_uint64(0)
    rc = lib.vp_fft_gkr(ctx, lg, tape.ctypes.data, nt.value, msgs.ctypes.data, nm.value, ctypes.byref(nw))
    assert rc == 0, lib.vp_last_error(ctx)
    assert nw.value == nm.value
    L = ob.lib()
    L.orc_fft_gkr.restype = ctypes.c_int64
    L.orc_fft_gkr.argtypes = [ctypes.c_int, ctypes.c_long, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
    buf = ctypes.create_string_buffer(16 * int(nm.value))
    ok = ctypes.c_int(0)
    n = L.orc_fft_gkr(lg, 3396, buf, len(buf), None, ctypes.byref(ok))
    assert n == 16 * nm.value and ok.value == 1
    got = msgs.tobytes()
    if got != buf.raw:
        first = next(i for i in range(int(nm.value)) if got[16 * i:16 * i + 16] != buf.raw[16 * i:16 * i + 16])
        raise AssertionError("fft_gkr(lg=%d): first differing message element %d of %d" % (lg, first, nm.value))
    if lg in (7, 13, 17):
        assert got == open(os.path.join(GOLDEN_DIR, "fftgkr_lg%d.bin" % lg), "rb").read()
    # a second call on the same context (buffers reused), another tape: still the oracle's answer
    ob.lib().orc_f_random_seq(77, int(nt.value), tape.ctypes.data)
    assert lib.vp_fft_gkr(ctx, lg, tape.ctypes.data, nt.value, msgs.ctypes.data, nm.value, None) == 0
    assert L.orc_fft_gkr(lg, 77, buf, len(buf), None, ctypes.byref(ok)) == n and ok.value == 1
    assert msgs.tobytes() == buf.raw
    # misuse: wrong sizes, non-canonical tape element
    assert lib.vp_fft_gkr(ctx, lg, tape.ctypes.data, nt.value - 1, msgs.ctypes.data, nm.value, None) == -1
    assert lib.vp_fft_gkr(ctx, lg, tape.ctypes.data, nt.value, msgs.ctypes.data, nm.value - 1, None) == -1
    tape[3, 1] = np.uint64(P)
    assert lib.vp_fft_gkr(ctx, lg, tape.ctypes.data, nt.value, msgs.ctypes.data, nm.value, None) == -1
    assert lib.vp_fft_gkr(ctx, 21, tape.ctypes.data, nt.value, msgs.ctypes.data, nm.value, None) == -5


def test_launch_stats_table_covers_every_launch(vp, gold_gkr):
    """vp_set_profiling + vp_get_launch_stats (include/vpgpu.h): the profiled replay produces the same transcript and a per-launch
    table that names every kernel kind of the plan with its algorithmic bytes, rounds and an event-measured duration; the
    commitment's calls report theirs."""
    c = vp.Circuit.randomize(8, 12, seed=1)
    s = vp.Session(c)
    s.draw_tape()
    tr, res = s.prove_gkr()
    assert tr == gold_gkr("randomize_8_12")
    s.set_profiling(1)
    tr_p, res_p = s.prove_gkr()
    st = s.launch_stats()
    s.set_profiling(0)
    assert tr_p == tr
    kinds = {e["kernel"] for e in st}
    assert {"k_beta_half_direct", "k_seg_multi", "k_emit_multi"} <= kinds and ("k_light_multi" in kinds)
    assert len(st) == res_p["launches"]
    assert all(e["us"] > 0 for e in st)
    assert all(e["bytes"] > 0 for e in st if e["kernel"] not in ("k_beta_half_direct", "k_fixup"))
    covered = set()
    for e in st:
        if e["rounds"]:
            covered |= set(range(e["first_round"], e["first_round"] + e["rounds"]))
    assert 1 in covered and max(covered) >= c.layer_bitlen(0)       # the round-covering launches span round 1 .. the longest sumcheck
    assert res_p["fold_launches"] == sum(1 for e in st if e["kernel"] == "k_sumfold3b_multi")
    # the commitment's three calls
    s.set_profiling(1)
    s.commit_private()
    cp = s.launch_stats()
    assert {"k_ntt_lds", "k_leaf_hash", "k_merkle"} <= {e["kernel"] for e in cp}
    leaf = [e for e in cp if e["kernel"] == "k_leaf_hash"][0]
    assert leaf["work"] == 65 * (1 << (c.layer_bitlen(0) - 2))      # 65 chained Keccak-f per leaf, 2^(n-2) leaves (fri.cpp:96-124)
    s.set_profiling(0)
    s.close(); c.close()


def test_randomize_transcript_matches_reference(vp, golden, gold_gkr):
    c = vp.Circuit.randomize(8, 12, seed=1)
    _both_modes(vp, c, gold_gkr("randomize_8_12"))
    c.close()


def test_sha256_x64_full_size(vp, golden, gold_gkr, pws_path):
    """BASELINE.json configs[1]: 64-block SHA-256, sumcheck on the GPU, PC off."""
    c = vp.Circuit.from_pws(pws_path, 64, seed=1)
    assert c.hash() == golden["sha256_x64"]["circuit_hash"]
    s = vp.Session(c)
    s.draw_tape()
    tr, res = s.prove_gkr()
    assert tr == gold_gkr("sha256_x64")
    assert res["rounds"] == golden["sha256_x64"]["rounds"]
    ok, _ = s.check(tr, device_predicates=True)
    assert ok
    s.close(); c.close()


@pytest.mark.parametrize("layers,log_size,seed", [(2, 0, 1), (2, 1, 2), (3, 1, 3), (3, 2, 4), (4, 3, 5), (5, 4, 6), (6, 5, 7),
                                                   (9, 2, 8), (12, 3, 9), (3, 9, 10), (4, 13, 11),
                                                   (24, 7, 12), (40, 8, 13)])          # deep: the closing launch holds 32 / 16 entries per table
def test_small_and_ragged_circuits_vs_oracle(vp, ob, layers, log_size, seed):
    """Edge cases: single-entry tables, empty subsets, tables that retire into add_term early,
    zero-round phases (SURVEY.md §7 'loader quirks')."""
    c = vp.Circuit.randomize(layers, log_size, seed=seed)
    oc = ob.Circuit.randomize(layers, log_size, seed=seed)
    assert c.hash() == oc.hash()
    gold, st = oc.prove_gkr()
    assert st["verified"] == 1
    _both_modes(vp, c, gold)
    c.close(); oc.close()


def test_evaluate_matches_oracle_outputs(vp, ob, pws_path):
    """circuitValue of the output layer feeds Vres: first 16 transcript bytes; also inputs round-trip."""
    c = vp.Circuit.from_pws(pws_path, 2, seed=5)
    oc = ob.Circuit.from_pws(pws_path, 2, seed=5)
    s = vp.Session(c)
    vals = s.layer_values(0)
    exp = np.zeros((oc.layer_size(0), 2), dtype=np.uint64)
    ob.lib().orc_circuit_inputs(oc.h, exp.ctypes.data)
    assert np.array_equal(vals, exp)
    gold, _ = oc.prove_gkr()
    s.draw_tape()
    tr, _ = s.prove_gkr()
    assert tr == gold
    s.close(); c.close(); oc.close()


# ---- Virgo polynomial commitment, commit side --------------------------------------------------------------
def test_sha3_matches_hashlib(vp, ctx):
    import hashlib
    rng = np.random.default_rng(5)
    n = 1000
    msgs = rng.integers(0, 256, size=(n, 64), dtype=np.uint8)
    msgs[0] = 0
    msgs[1] = 255
    out = np.zeros((n, 32), dtype=np.uint8)
    assert vp.lib_gpu().vp_test_sha3(ctx, msgs.ctypes.data, out.ctypes.data, n) == 0
    for i in range(n):
        assert out[i].tobytes() == hashlib.sha3_256(msgs[i].tobytes()).digest()
    # known answer recorded from the compiled reference: my_hhash(64 zero bytes)
    assert out[0].tobytes().hex() == "070fa1ab6fcc557ed14d42941f1967693048551eb9042a8d0a057afbd75e81e0"


@pytest.mark.parametrize("ln,ratio", [(0, 1), (1, 1), (3, 32), (3, 1), (7, 32), (10, 1), (12, 32), (13, 1), (13, 32), (14, 1), (14, 32), (15, 1), (15, 32), (16, 1), (16, 32), (17, 1), (17, 32)])
def test_fft_vs_oracle(vp, ob, ctx, ln, ratio):
    rng = np.random.default_rng(ln * 7 + ratio)
    n = 1 << ln
    c = rng.integers(0, P, size=(n, 2), dtype=np.uint64)
    out = np.zeros((n * ratio, 2), dtype=np.uint64)
    exp = np.zeros_like(out)
    assert vp.lib_gpu().vp_test_fft(ctx, c.ctypes.data, n, n * ratio, 0, out.ctypes.data) == 0
    ob.lib().orc_fft.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    ob.lib().orc_fft(c.ctypes.data, n, n * ratio, exp.ctypes.data)
    assert np.array_equal(out, exp)
    if ratio == 1:      # inverse round trip and parity
        back = np.zeros_like(c)
        assert vp.lib_gpu().vp_test_fft(ctx, out.ctypes.data, n, n, 1, back.ctypes.data) == 0
        assert np.array_equal(back, c)


def test_fft_known_answer(vp, ctx):
    # c_i = (i+1, 2i+3), i < 8, evaluated on the order-32 group
    c = np.array([[i + 1, 2 * i + 3] for i in range(8)], dtype=np.uint64)
    # the commitment only uses ratios 1 and 32; 8 -> 32 is ratio 4, so embed: evaluate 8 coefficients padded to... use the
    # ratio-32 shape on a 1-coefficient-per-... (not expressible) -> check through the oracle-equivalent ratio 1 on 32 padded coefs
    pad = np.zeros((32, 2), dtype=np.uint64)
    pad[:8] = c
    out = np.zeros((32, 2), dtype=np.uint64)
    assert vp.lib_gpu().vp_test_fft(ctx, pad.ctypes.data, 32, 32, 0, out.ctypes.data) == 0
    assert tuple(int(x) for x in out[0]) == (36, 80)
    assert tuple(int(x) for x in out[1]) == (617377187976651873, 1655836513184006865)
    assert tuple(int(x) for x in out[31]) == (107673715198561662, 1973828267175406033)


@pytest.mark.parametrize("name,blocks", [("sha256_x1", 1), ("sha256_x16", 16), ("sha256_x64", 64)])
def test_commit_private_root_matches_reference(vp, golden, pws_path, name, blocks):
    """merkle_root_l: first 32 bytes of the real reference's transcript."""
    from conftest import GOLDEN
    import os
    c = vp.Circuit.from_pws(pws_path, blocks, seed=1)
    s = vp.Session(c)
    root, ms = s.commit_private()
    gold = open(os.path.join(GOLDEN, golden[name]["transcript"]), "rb").read()[:32]
    assert root == gold
    root2, _ = s.commit_private()          # idempotent
    assert root2 == gold
    s.close(); c.close()


def test_commit_private_randomize(vp, golden):
    from conftest import GOLDEN
    import os
    c = vp.Circuit.randomize(8, 12, seed=1)
    s = vp.Session(c)
    root, _ = s.commit_private()
    assert root == open(os.path.join(GOLDEN, golden["randomize_8_12"]["transcript"]), "rb").read()[:32]
    s.close(); c.close()


@pytest.mark.parametrize("name,blocks", [("sha256_x1", 1), ("sha256_x16", 16), ("sha256_x64", 64)])
@pytest.mark.parametrize("batched", [False, True])
def test_full_transcript_with_commitment_matches_reference(vp, golden, pws_path, name, blocks, batched):
    """merkle_root_l | GKR | merkle_root_h | input_0 | all_sum[65]: the complete golden transcript, whose SHA-256
    equals the digest recorded in SURVEY.md §8c."""
    import hashlib, os
    from conftest import GOLDEN
    if blocks == 64 and not batched:
        pytest.skip("the interactive 64-block run spends seconds in the host verifier's predicate loops")
    c = vp.Circuit.from_pws(pws_path, blocks, seed=1)
    s = vp.Session(c)
    tr, ok = s.prove_full(batched=batched)
    gold = open(os.path.join(GOLDEN, golden[name]["transcript"]), "rb").read()
    assert ok
    assert tr == gold
    assert hashlib.sha256(tr).hexdigest() == golden[name]["sha256"]
    s.close(); c.close()


def test_full_transcript_randomize(vp, golden):
    import os
    from conftest import GOLDEN
    c = vp.Circuit.randomize(8, 12, seed=1)
    s = vp.Session(c)
    tr, ok = s.prove_full(batched=True)
    assert ok and tr == open(os.path.join(GOLDEN, golden["randomize_8_12"]["transcript"]), "rb").read()
    s.close(); c.close()


def test_commit_public_vs_oracle_random_vector(vp, ob, pws_path):
    """commit_public on an arbitrary public vector (not an eq table) against the oracle's restatement."""
    c = vp.Circuit.from_pws(pws_path, 1, seed=9)
    oc = ob.Circuit.from_pws(pws_path, 1, seed=9)
    s = vp.Session(c)
    s.commit_private()
    n_bits = c.layer_bitlen(0)
    rng = np.random.default_rng(4)
    pub = rng.integers(0, P, size=(1 << n_bits, 2), dtype=np.uint64)
    pub[5] = 0
    root_h, inner, all_sum, _ = s.commit_public(pub)
    inp = np.zeros((1 << n_bits, 2), dtype=np.uint64)
    ob.lib().orc_circuit_inputs(oc.h, inp.ctypes.data)
    e_inner = np.zeros(2, dtype=np.uint64); e_all = np.zeros((65, 2), dtype=np.uint64); e_root = ctypes.create_string_buffer(32)
    ob.lib().orc_commit_public.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p]
    assert ob.lib().orc_commit_public(inp.ctypes.data, pub.ctypes.data, n_bits, oc.layer_size(0), e_inner.ctypes.data, e_all.ctypes.data, e_root) == 0
    assert inner == e_inner.tobytes() and all_sum == e_all.tobytes() and root_h == e_root.raw
    s.close(); c.close(); oc.close()


def _fri_golden(golden, name):
    import os
    from conftest import GOLDEN
    g = golden[name]
    fri = open(os.path.join(GOLDEN, g["fri"]), "rb").read()
    st = g["fri_steps"]
    rec = np.frombuffer(fri[: 48 * st], dtype=np.uint64).reshape(st, 6)
    r = np.ascontiguousarray(rec[:, :2])
    roots = b"".join(rec[i, 2:].tobytes() for i in range(st))
    fin = np.frombuffer(fri[48 * st: 48 * st + 2048 * 16], dtype=np.uint64).reshape(2048, 2)
    return r, roots, fin


@pytest.mark.parametrize("name,blocks", [("sha256_x1", 1), ("sha256_x16", 16), ("sha256_x64", 64)])
def test_protocol_pass_matches_reference(vp, golden, pws_path, name, blocks):
    """vph_prove_protocol — the prover side of the complete protocol in one pass from a tape drawn up front (commit_private, batched GKR,
    vp_commit_public_eq on the opening point, fft_gkr, FRI commit phase; bench.py's configs[2] step): the real reference's whole transcript,
    its FRI roots and its final codeword, byte for byte; twice in a row (the second pass runs on the buffers of the first)."""
    import os
    from conftest import GOLDEN
    c = vp.Circuit.from_pws(pws_path, blocks, seed=1)
    s = vp.Session(c)
    s.draw_protocol_tape()
    gold = open(os.path.join(GOLDEN, golden[name]["transcript"]), "rb").read()
    _, roots_gold, fin_gold = _fri_golden(golden, name)
    for _ in range(2):
        tr, roots, fin, sec = s.prove_protocol()
        assert tr == gold
        assert roots == roots_gold and np.array_equal(fin, fin_gold)
        assert sec["total"] >= sec["gkr"] + sec["commit_private"] + sec["commit_public"] + sec["fft_gkr"] + sec["fri_commit"] - 1e-6
    if name == "sha256_x1":                            # fft_gkr's messages inside the protocol run: the reference's record
        assert s.last_fft_gkr() == open(os.path.join(GOLDEN, "fftgkr_sha256_x1.bin"), "rb").read()
    s.close(); c.close()


def test_leaf_hash_generated_chains_equal_the_compilers_at_every_workgroup_size(vp, pws_path):
    """The leaf-hash chains of the commitment by the generated fixed-register block (csrc/vp_keccak_asm.h: workgroups of 1024 threads from 2^18 leaves on, of
    512 at 2^17, waves in phase) against the compiler's Keccak-f in workgroups of 256 (leaf_asm = 0), on the whole prover pass of the protocol: x256 (2^19-leaf
    trees; the FRI levels of 2^18 ... 16 leaves end to end in one launch) and x64 (2^17): transcript, Merkle roots, FRI roots and final codeword are equal."""
    for blocks in (256, 64):
        c = vp.Circuit.from_pws(pws_path, blocks, seed=1)
        out = []
        for asm in (1, 0):
            s = vp.Session(c, options=vp.Options(leaf_asm=asm))
            s.draw_protocol_tape()
            tr, roots, fin, _ = s.prove_protocol()
            out.append((tr, roots, fin.copy()))
            s.close()
        assert out[0][0] == out[1][0] and out[0][1] == out[1][1] and np.array_equal(out[0][2], out[1][2])
        c.close()


@pytest.mark.parametrize("blocks", [64, 256, 512])
def test_commit_private_two_real_slices_per_transform(vp, golden, pws_path, blocks):
    """vp_commit_private of a real witness sends slices p and p + 32 through each transform as ONE complex sequence (real_pairs; the encoder's last store
    separates the two slices' values on every coset): the root equals the one-transform-per-slice form's (and, at x64, the real reference's)."""
    c = vp.Circuit.from_pws(pws_path, blocks, seed=1)
    roots = []
    for rp in (1, 0):
        s = vp.Session(c, options=vp.Options(real_pairs=rp))
        root, _ = s.commit_private()
        roots.append(root)
        s.close()
    assert roots[0] == roots[1]
    if blocks == 64:
        import os
        from conftest import GOLDEN
        assert roots[0] == open(os.path.join(GOLDEN, golden["sha256_x64"]["transcript"]), "rb").read()[:32]
    c.close()


def test_commit_public_eq_equals_commit_public_on_the_table(vp, pws_path):
    """vp_commit_public_eq(point) == vp_commit_public(eq(point, .)): root, inner product, all_sum — for a random point, a point with a zero
    coordinate and a point with a coordinate 1 (pub[0] = 0: the tensor shortcut must step aside)."""
    c = vp.Circuit.from_pws(pws_path, 2, seed=3)
    s = vp.Session(c)
    s.commit_private()
    n = c.layer_bitlen(0)
    rng = np.random.default_rng(11)
    for case in range(3):
        point = rng.integers(0, P, size=(n, 2), dtype=np.uint64)
        if case == 1:
            point[3] = 0
        if case == 2:
            point[n - 2] = (1, 0)
        want = s.commit_public(s.eq_table(point))[:3]
        got = s.commit_public_eq(point)[:3]
        assert got == want
    bad = rng.integers(0, P, size=(n, 2), dtype=np.uint64)
    bad[0, 0] = P                                       # non-canonical coordinate: refused at the door
    with pytest.raises(RuntimeError):
        s.commit_public_eq(bad)
    with pytest.raises(RuntimeError):
        s.commit_public_eq(bad[:-1])
    s.close(); c.close()


@pytest.mark.parametrize("batched", [True, False])
@pytest.mark.parametrize("name,blocks", [("sha256_x1", 1), ("sha256_x16", 16), ("sha256_x64", 64)])
def test_fri_commit_phase_matches_reference(vp, golden, pws_path, name, blocks, batched):
    """Every FRI Merkle root and the final codeword of the real reference, given its recorded fold challenges: all steps in
    one device pass (vp_fri_commit) and one vp_fri_step per challenge."""
    c = vp.Circuit.from_pws(pws_path, blocks, seed=1)
    s = vp.Session(c)
    tr, ok = s.prove_full(batched=True)             # leaves l, q (-> virtual oracle) and h in HBM
    assert ok
    r, roots_gold, fin_gold = _fri_golden(golden, name)
    roots, fin = s.fri_commit(r, batched=batched)
    assert roots == roots_gold
    assert np.array_equal(fin, fin_gold)
    with pytest.raises(RuntimeError):               # the commit phase is over
        s.fri_commit(r[:1], batched=batched)
    s.close(); c.close()


def test_fri_commit_phase_randomize(vp, golden):
    c = vp.Circuit.randomize(8, 12, seed=1)
    s = vp.Session(c)
    s.prove_full(batched=True)
    r, roots_gold, fin_gold = _fri_golden(golden, "randomize_8_12")
    roots, fin = s.fri_commit(r)
    assert roots == roots_gold and np.array_equal(fin, fin_gold)
    s.close(); c.close()


def test_commitment_with_split_transforms_vs_oracle(vp, ob, pws_path):
    """128 blocks: slices of 2^14 elements, beyond the in-LDS transform -> split path.  No golden at this size:
    compare commit_private / commit_public with the oracle's restatement on the same witness and public vector."""
    c = vp.Circuit.from_pws(pws_path, 128, seed=3)
    oc = ob.Circuit.from_pws(pws_path, 128, seed=3)
    assert c.hash() == oc.hash()
    s = vp.Session(c)
    root, _ = s.commit_private()
    L = ob.lib()
    L.orc_commit_private.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    er = ctypes.create_string_buffer(32)
    assert L.orc_commit_private(oc.h, er) == 0
    assert root == er.raw
    n_bits = c.layer_bitlen(0)
    assert n_bits - 6 == 14
    rng = np.random.default_rng(8)
    pub = rng.integers(0, P, size=(1 << n_bits, 2), dtype=np.uint64)
    root_h, inner, all_sum, _ = s.commit_public(pub)
    inp = np.zeros((1 << n_bits, 2), dtype=np.uint64)
    L.orc_circuit_inputs(oc.h, inp.ctypes.data)
    e_inner = np.zeros(2, dtype=np.uint64); e_all = np.zeros((65, 2), dtype=np.uint64); e_root = ctypes.create_string_buffer(32)
    L.orc_commit_public.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p]
    assert L.orc_commit_public(inp.ctypes.data, pub.ctypes.data, n_bits, oc.layer_size(0), e_inner.ctypes.data, e_all.ctypes.data, e_root) == 0
    assert inner == e_inner.tobytes() and all_sum == e_all.tobytes() and root_h == e_root.raw
    s.close(); c.close(); oc.close()


def test_sha256_x128_vs_oracle(vp, ob, pws_path):
    """Beyond the recorded reference runs: 128 blocks (12.8 M gates, tables up to 2^23) against the oracle's transcript."""
    c = vp.Circuit.from_pws(pws_path, 128, seed=2)
    oc = ob.Circuit.from_pws(pws_path, 128, seed=2)
    assert c.hash() == oc.hash()
    gold, st = oc.prove_gkr()
    assert st["verified"] == 1
    s = vp.Session(c)
    s.draw_tape()
    tr, res = s.prove_gkr()
    assert tr == gold
    assert res["rounds"] == st["rounds"]
    s.close(); c.close(); oc.close()


def test_sha256_x1024_full_size_vs_real_reference(vp, golden, gold_gkr, pws_path):
    """BASELINE.json configs[2] / [3] size: 1024 blocks, 102 M gates, tables up to 2^26, 859 rounds.  The expected transcript is the REAL
    REFERENCE's (round 3: oracle/_ref/ref_run --blocks 1024 --pc 1 in the build container — Prove Time 183 s, commitment 576 s, 63 GB;
    tests/golden/transcript_sha256_x1024.bin, fri_sha256_x1024.bin); the oracle's own fixtures of the same run (made on the GPU box's host
    in round 2) are kept as a second check and are byte-identical where they overlap.  Batched proof, and the same proof sharded over 8 ranks."""
    g = golden["sha256_x1024"]
    assert g["origin"].startswith("real reference")
    ref_full = open(os.path.join(GOLDEN_DIR, g["transcript"]), "rb").read()
    ref_fri = open(os.path.join(GOLDEN_DIR, g["fri"]), "rb").read()
    assert gold_gkr("sha256_x1024") == open(os.path.join(GOLDEN_DIR, g["oracle_fixture"]["transcript"]), "rb").read()
    c = vp.Circuit.from_pws(pws_path, 1024, seed=1)
    assert c.gates == 102347776 and c.hash() == g["circuit_hash"]
    s = vp.Session(c)
    s.draw_tape()
    tr, res = s.prove_gkr()
    assert tr == gold_gkr("sha256_x1024")
    assert res["rounds"] == g["rounds"] == 859
    assert vp.sum_transcripts(_sharded_parts(vp, s, 8)) == tr
    ok, _ = s.check(tr, device_predicates=True)
    assert ok
    # the commitment at the same size (input layer 2^23: 65 slices of 2^22 code symbols, transforms of 2^17, 2^21 leaves): the
    # oracle's outputs (tests/golden/make_oracle_fixture_pc.py: 273 s of CPU, 22 GB) = merkle_root_l | merkle_root_h | input_0 | all_sum[65]
    exp = open(os.path.join(GOLDEN_DIR, "oracle_sha256_x1024_pc.bin"), "rb").read()
    root, _ = s.commit_private()
    assert root == exp[:32]
    pub = np.random.default_rng(8).integers(0, P, size=(1 << c.layer_bitlen(0), 2), dtype=np.uint64)
    root_h, inner, all_sum, _ = s.commit_public(pub)
    assert root_h == exp[32:64] and inner == exp[64:80] and all_sum == exp[80:]
    # ... and the FRI commit phase (the low-degree test's prover side): 17 fold steps, every Merkle root and the final codeword
    # (tests/golden/make_oracle_fixture_fri.py, fold challenges from default_rng(9))
    fexp = open(os.path.join(GOLDEN_DIR, "oracle_sha256_x1024_fri.bin"), "rb").read()
    st = c.layer_bitlen(0) - 6
    assert st == 17
    r = np.random.default_rng(9).integers(0, P, size=(st, 2), dtype=np.uint64)
    roots, fin = s.fri_commit(r)
    assert roots == fexp[:32 * st], "FRI roots differ from the oracle's"
    assert fin.tobytes() == fexp[32 * st:], "final FRI codeword differs"
    # The COMPLETE protocol in one unbroken run at this size (verifier::verify(): commit_private, interactive GKR, commit_public
    # on the protocol's own public vector eq(r_liu, .), fft_gkr's draws, FRI commit phase, query repetitions) against the oracle's
    # run of the same (tests/golden/make_oracle_fixture_full.py 1024 1 --fri: 22 min of CPU, 34 GB): full transcript, every FRI
    # root, the final codeword and the fold challenges the reference's verifier would draw.
    fx = open(os.path.join(GOLDEN_DIR, "oracle_sha256_x1024_full.bin"), "rb").read()
    trf, okf, _ = s.prove_and_verify_full(reps=3)
    assert okf
    n = len(trf)
    assert n == 32 + len(tr) + 32 + 16 + 65 * 16 and trf == fx[:n], "full transcript differs from the oracle's"
    assert trf == ref_full, "full transcript (root_l | GKR | root_h | input_0 | all_sum) differs from the real reference's"
    roots2, fin2, r2 = s.last_fri()
    assert roots2 == fx[n:n + 32 * st] and fin2.tobytes() == fx[n + 32 * st:n + 32 * st + 2048 * 16]
    assert r2.tobytes() == fx[n + 32 * st + 2048 * 16:], "FRI fold challenges differ from the reference's draw order"
    # the real reference's FRI record: per step challenge[16] | root[32], then the final codeword and the (all-zero) mask codeword
    assert b"".join(ref_fri[48 * k:48 * k + 16] for k in range(st)) == r2.tobytes(), "FRI challenges differ from the real reference's"
    assert b"".join(ref_fri[48 * k + 16:48 * k + 48] for k in range(st)) == roots2, "FRI roots differ from the real reference's"
    assert ref_fri[48 * st:48 * st + 2048 * 16] == fin2.tobytes() and ref_fri[48 * st + 2048 * 16:] == bytes(32 * 16)
    trb, okb = s.prove_full(batched=True)
    assert okb and trb == trf
    # ... and the commitment of this size sharded over 8 ranks (8 slices per rank, 2^18 leaves per rank): same roots
    inputs = s.layer_values(0)
    pub = s.eq_table(s.last_point())
    s.close(); c.close()
    r8 = np.frombuffer(fx[n + 32 * st + 2048 * 16:], dtype=np.uint64).reshape(st, 2).copy()
    fin8 = np.frombuffer(fx[n + 32 * st:n + 32 * st + 2048 * 16], dtype=np.uint64).reshape(2048, 2)
    _sharded_commitment_case(vp, inputs, 23, pub, r8, 8, fx[:32], fx[n - (32 + 16 + 65 * 16):n], fx[n:n + 32 * st], fin8)
    # configs[3] draws witness seeds 1..8 (one proof per GPU): seed 2 against its own oracle fixture (make_oracle_fixture_full.py 1024 2 --gkr-only)
    c2 = vp.Circuit.from_pws(pws_path, 1024, seed=2)
    s2 = vp.Session(c2)
    s2.draw_tape()
    tr2, _ = s2.prove_gkr()
    assert tr2 != tr and tr2 == open(os.path.join(GOLDEN_DIR, "oracle_sha256_x1024_gkr_seed2.bin"), "rb").read()
    s2.close(); c2.close()


def _opening_ok(root, leaf, vals, path):
    """Python restatement of the verifier's opening check (lib/virgo/src/vpd_verifier.cpp:9-40, fri.cpp:96-124): the leaf is the chain
    of 65 SHA3-256 over (value pair || previous digest), then the path to the root."""
    import hashlib
    h = bytes(32)
    for k in range(65):
        h = hashlib.sha3_256(vals[2 * k].tobytes() + vals[2 * k + 1].tobytes() + h).digest()
    depth = len(path) - 1
    if h != path[depth]:
        return False
    pos = leaf
    for k in range(depth):
        h = hashlib.sha3_256((path[k] + h) if (pos & 1) else (h + path[k])).digest()
        pos >>= 1
    return h == root


def _sharded_commitment_case(vp, inputs, n, pub, r, world, root_l, tail_gold, roots_gold, fin_gold):
    sc = vp.ShardedCommitment(inputs, n, world)
    assert sc.commit_private() == root_l, "world %d: merkle_root_l" % world
    root_h, inner, all_sum = sc.commit_public(pub)
    assert root_h + inner + all_sum == tail_gold, "world %d: merkle_root_h | input_0 | all_sum" % world
    roots, fin = sc.fri_commit(r)
    assert roots == roots_gold, "world %d: FRI roots" % world
    assert np.array_equal(fin, fin_gold), "world %d: final codeword" % world
    # openings: answered by the owner of the leaf only, and they verify against the assembled roots
    st = r.shape[0]
    lw = world.bit_length() - 1
    for oracle, root, n_leaves in [(0, root_l, 1 << (n - 2)), (1, root_h, 1 << (n - 2)), (2, roots[:32], 1 << (n - 3)),
                                   (2 + st - lw - 2, roots[32 * (st - lw - 2):32 * (st - lw - 1)], 16 << (lw + 1)),        # last locally hashed level
                                   (2 + st - lw - 1, roots[32 * (st - lw - 1):32 * (st - lw)], 16 << lw),                  # first replicated level
                                   (2 + st - 1, roots[32 * (st - 1):], 16)]:
        if oracle < 2 or oracle - 2 < 0:
            pass
        for leaf in sorted({0, 33 % n_leaves, n_leaves // 2 + 5 if n_leaves > 16 else 3, n_leaves - 1}):
            got = sc.open(oracle, leaf)
            assert got is not None, "world %d oracle %d leaf %d" % (world, oracle, leaf)
            assert _opening_ok(root, leaf, *got), "world %d oracle %d leaf %d: opening does not verify" % (world, oracle, leaf)
        if oracle < 2 + st - lw - 1 and world > 1:
            assert sc.open(oracle, 0, rank=1) is None, "a rank that does not own the leaf answered"
    ms = sc.device_ms()
    sc.close()
    return ms


@pytest.mark.parametrize("name,blocks,worlds", [("sha256_x1", 1, (2, 4, 8)), ("sha256_x16", 16, (2, 8)), ("sha256_x64", 64, (8,))])
def test_commitment_sharded_over_ranks(vp, golden, pws_path, name, blocks, worlds):
    """SURVEY §8e "PC sharding": the commitment over W ranks — slices dealt to the ranks, one all-to-all per oracle to position
    ownership, local leaf hashes / five tree levels / FRI folds, level-5 nodes all-gathered, top of the tree on every rank.  W light
    contexts on this one GPU with the collectives done by vp_shard_exchange_local; every rank must return the real reference's
    merkle_root_l, merkle_root_h, input_0, all_sum[65], all FRI roots and the final codeword, and owners' openings must verify."""
    import os
    from conftest import GOLDEN
    g = golden[name]
    gold = open(os.path.join(GOLDEN, g["transcript"]), "rb").read()
    c = vp.Circuit.from_pws(pws_path, blocks, seed=1)
    s = vp.Session(c)
    full, ok = s.prove_full(batched=True)
    assert ok and full == gold
    n = c.layer_bitlen(0)
    inputs = s.layer_values(0)
    pub = s.eq_table(s.last_point())
    r, roots_gold, fin_gold = _fri_golden(golden, name)
    s.close(); c.close()
    for world in worlds:
        _sharded_commitment_case(vp, inputs, n, pub, r, world, gold[:32], gold[-(32 + 16 + 65 * 16):], roots_gold, fin_gold)


def test_rccl_transport_on_one_rank(vp, golden, pws_path):
    """The RCCL transport of include/vpgpu.h (vp_comm_unique_id / vp_comm_init, librccl resolved at run time) on the one GPU of this
    box: a one-rank communicator; vp_allreduce_u64 on a device buffer; and the sharded commitment code path with world = 1, whose
    all-to-all (ncclSend / ncclRecv group) and all-gather then run through RCCL instead of the in-process exchange."""
    import os
    from conftest import GOLDEN
    L = vp.lib_gpu()
    g = golden["sha256_x1"]
    gold = open(os.path.join(GOLDEN, g["transcript"]), "rb").read()
    c = vp.Circuit.from_pws(pws_path, 1, seed=1)
    s = vp.Session(c)
    full, ok = s.prove_full(batched=True)
    assert ok and full == gold
    inputs = s.layer_values(0)
    pub = s.eq_table(s.last_point())
    n = c.layer_bitlen(0)
    s.close(); c.close()
    uid = ctypes.create_string_buffer(128)
    assert L.vp_comm_unique_id(ctypes.cast(uid, ctypes.c_void_p)) == 0
    ctx = ctypes.c_void_p()
    assert L.vp_create(0, ctypes.byref(ctx)) == 0
    assert L.vp_comm_init(ctx, ctypes.cast(uid, ctypes.c_void_p), 0, 1) == 0, L.vp_last_error(ctx)
    # all-reduce of a device buffer (sum over one rank = identity), through hipMalloc'ed memory
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), 8 * 1000) == 0
    a = np.random.default_rng(3).integers(0, 1 << 62, size=1000, dtype=np.uint64)
    assert hip.hipMemcpy(buf, a.ctypes.data, a.nbytes, 1) == 0
    assert L.vp_allreduce_u64(ctx, buf, 1000) == 0
    assert hip.hipDeviceSynchronize() == 0
    b = np.zeros_like(a)
    assert hip.hipMemcpy(b.ctypes.data, buf, a.nbytes, 2) == 0
    assert np.array_equal(a, b)
    hip.hipFree(buf)
    # the sharded commitment, one rank, collectives over RCCL: the reference's roots
    inputs = np.ascontiguousarray(inputs, dtype=np.uint64)
    assert L.vp_pc_load_input(ctx, inputs.ctypes.data, inputs.shape[0], n) == 0
    assert L.vp_pc_set_shard(ctx, 0, 1) == 0
    root = ctypes.create_string_buffer(32)
    assert L.vp_commit_private(ctx, ctypes.cast(root, ctypes.c_void_p)) == 0, L.vp_last_error(ctx)
    assert root.raw == gold[:32]
    pub = np.ascontiguousarray(pub, dtype=np.uint64)
    inner = np.zeros(2, np.uint64); alls = np.zeros((65, 2), np.uint64); rh = ctypes.create_string_buffer(32)
    assert L.vp_commit_public(ctx, pub.ctypes.data, pub.shape[0], inner.ctypes.data, alls.ctypes.data, ctypes.cast(rh, ctypes.c_void_p)) == 0, L.vp_last_error(ctx)
    assert rh.raw + inner.tobytes() + alls.tobytes() == gold[-(32 + 16 + 65 * 16):]
    r, roots_gold, fin_gold = _fri_golden(golden, "sha256_x1")
    roots = ctypes.create_string_buffer(32 * r.shape[0])
    assert L.vp_fri_commit(ctx, r.ctypes.data, r.shape[0], ctypes.cast(roots, ctypes.c_void_p)) == 0, L.vp_last_error(ctx)
    assert roots.raw == roots_gold
    assert L.vp_comm_destroy(ctx) == 0
    L.vp_destroy(ctx)


def test_sha256_x256_size_independent_properties(vp, pws_path):
    """256 blocks (25.6 M gates): no oracle run at this size; the host verifier (sumcheck identities, Liu identity,
    final input-layer check) must accept the device transcript, the proof must be reproducible, and a different
    witness must change it."""
    c = vp.Circuit.from_pws(pws_path, 256, seed=11)
    s = vp.Session(c)
    s.draw_tape()
    tr, _ = s.prove_gkr()
    ok, _ = s.check(tr, device_predicates=True)
    assert ok
    tr2, _ = s.prove_gkr()
    assert tr2 == tr
    bad = bytearray(tr); bad[100] ^= 4
    ok_bad, _ = s.check(bytes(bad), device_predicates=True)
    assert not ok_bad
    s.close(); c.close()


def test_randomize_16_20_synthetic_config(vp, golden, gold_gkr):
    """BASELINE.json configs[4] / SURVEY §8d config 5: layeredCircuit::randomize(16, 20) = 16 layers of 2^20 random Mul/Add
    gates (2^24 gates).  The interactive and the batched device proofs must equal the REAL REFERENCE's transcript (round 3:
    oracle/_ref/ref_run --randomize 16 20 --pc 1 in the build container, Prove Time 28.3 s + commitment 60.5 s; the oracle's fixture of
    round 2 is the same bytes), the 8-way chain-sharded proof must assemble to it, the host verifier (all sumcheck / Liu identities, wiring
    predicates on the device) must accept, and the complete protocol with the commitment in one unbroken run must reproduce the
    reference's roots, input_0, all_sum, FRI challenges, FRI roots and final codeword."""
    g = golden["randomize_16_20"]
    assert g["origin"].startswith("real reference")
    assert gold_gkr("randomize_16_20") == open(os.path.join(GOLDEN_DIR, g["oracle_fixture"]["transcript"]), "rb").read()
    c = vp.Circuit.randomize(16, 20, seed=1)
    assert c.hash() == golden["randomize_16_20"]["circuit_hash"]
    s = vp.Session(c)
    tr, res, ok = s.prove_interactive()
    assert ok
    assert tr == gold_gkr("randomize_16_20")
    s.draw_tape()
    tr2, res2 = s.prove_gkr()
    assert tr2 == tr
    assert vp.sum_transcripts(_sharded_parts(vp, s, 8)) == tr
    assert res["rounds"] == res2["rounds"] == g["rounds"]
    ok2, _ = s.check(tr2, device_predicates=True)
    assert ok2
    trf, okf, _ = s.prove_and_verify_full(reps=5)
    assert okf and trf == open(os.path.join(GOLDEN_DIR, g["transcript"]), "rb").read()
    r_gold, roots_gold, fin_gold = _fri_golden(golden, "randomize_16_20")
    roots, fin, r = s.last_fri()
    assert np.array_equal(r, r_gold) and roots == roots_gold and np.array_equal(fin, fin_gold)
    s.close(); c.close()


@pytest.mark.parametrize("seed,sizes", [(1, [40, 33, 50, 17]), (2, [200, 180, 150, 300, 64, 9]), (3, [1500, 2100, 900, 4100, 700]),
                                        (4, [5, 3, 2, 1]), (5, [70000, 50000, 30000]), (6, [300000, 280000, 150000, 270000])])
def test_all_gate_types_and_assert_gates_vs_oracle(vp, ob, seed, sizes):
    """Addc / Mulc / Copy / AntiNaab / AntiSub and assert gates are in the reference's API (src/prover.cpp:49-87,
    229-272,319-360,209-212) but in none of its data sets: random circuits with all types, device vs oracle."""
    import custom_circuits as cc
    args = cc.make(seed, sizes)
    c = vp.Circuit.custom(*args)
    oc = ob.Circuit.custom(*args)
    assert c.hash() == oc.hash()
    gold, st = oc.prove_gkr()
    assert st["verified"] == 1
    _both_modes(vp, c, gold)
    c.close(); oc.close()


@pytest.mark.parametrize("seed,sizes", [(2, [200, 180, 150, 300, 64, 9]), (3, [1500, 2100, 900, 4100, 700]), (4, [5, 3, 2, 1])])
def test_device_predicates_all_gate_types(vp, seed, sizes):
    """The verifier's wiring-predicate sums computed on the device (vp_predicates; verifier.cpp:50-113) must make the
    host verifier accept exactly what it accepts with its own loops — circuits with every gate type and assert gates —
    and reject a transcript with one final claim altered (the final-value check consumes all predicate sums)."""
    import custom_circuits as cc
    c = vp.Circuit.custom(*cc.make(seed, sizes))
    s = vp.Session(c)
    s.draw_tape()
    tr, _ = s.prove_gkr()
    ok_host, _ = s.check(tr)
    ok_dev, _ = s.check(tr, device_predicates=True)
    assert ok_host and ok_dev
    bad = bytearray(tr); bad[-8] ^= 1            # last vr: only the predicate-based checks of the layers below see it
    ok_bad_host, _ = s.check(bytes(bad))
    ok_bad_dev, _ = s.check(bytes(bad), device_predicates=True)
    assert not ok_bad_host and not ok_bad_dev
    s.close(); c.close()


def test_device_predicates_sha256_x64(vp, pws_path):
    """64 blocks: the device predicate path accepts the proof; timing of both paths is printed by bench.py."""
    c = vp.Circuit.from_pws(pws_path, 64, seed=1)
    s = vp.Session(c)
    s.draw_tape()
    tr, _ = s.prove_gkr()
    ok_dev, sec_dev = s.check(tr, device_predicates=True)
    assert ok_dev
    bad = bytearray(tr); bad[len(tr) // 2] ^= 16
    assert not s.check(bytes(bad), device_predicates=True)[0]
    s.close(); c.close()


def test_limits_and_call_order_are_errors_not_crashes(vp, ob, ctx):
    """Maximum sizes and misuse (the reference has `assert`s and UB there): the deepest circuit the library supports
    (VP_MAX_TAB = 64 layers) proves and matches the oracle; one layer more is VP_ELIMIT; ABI calls out of the reference's
    state-machine order, or with missing arguments, return VP_EINVAL and leave the context usable."""
    c = vp.Circuit.randomize(64, 3, seed=5)
    oc = ob.Circuit.randomize(64, 3, seed=5)
    gold, st = oc.prove_gkr()
    assert st["verified"] == 1
    s = vp.Session(c)
    s.draw_tape()
    tr, _ = s.prove_gkr()
    assert tr == gold
    tr_i, _, ok = s.prove_interactive()
    assert ok and tr_i == gold
    s.close(); c.close(); oc.close()
    too_deep = vp.Circuit.randomize(65, 2, seed=5)
    with pytest.raises(RuntimeError, match="too many layers|-5"):
        vp.Session(too_deep)
    too_deep.close()
    lib = vp.lib_gpu()
    poly = (ctypes.c_uint64 * 6)()
    r = (ctypes.c_uint64 * 2)(1, 2)
    assert lib.vp_round(ctx, r, poly) == -1                        # no sumcheck in progress
    assert lib.vp_finalize(ctx, r, poly, 1) == -1
    assert lib.vp_phase1_init(ctx, 1, None, None) == -1            # no circuit uploaded on this context
    lib.vp_prove_gkr.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
    assert lib.vp_prove_gkr(ctx, None, 0, None, 0, None) == -1
    assert lib.vp_commit_private(ctx, None) == -1
    assert lib.vp_circuit_upload(ctx, 1, None) == -1
    out = (ctypes.c_uint64 * 2)()
    a = (ctypes.c_uint64 * 2)(3, 4)
    assert lib.vp_test_field(ctx, 2, a, a, out, 1) == 0            # the context still works
    # a challenge with a limb >= p (the ABI asks for canonical elements; bits 61-63 carry the resident kernel's sequence tag): an error
    # at the door, not a 10 s stall inside the mailbox — in the middle of a sumcheck, which then simply continues
    c = vp.Circuit.randomize(3, 6, seed=9)
    s = vp.Session(c)
    lib.vp_phase1_init.argtypes = [ctypes.c_void_p] + [ctypes.c_int] + [ctypes.c_void_p] * 2
    lib.vp_round.argtypes = [ctypes.c_void_p] * 3
    lib.vp_finalize.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int]
    h = s.gpu_ctx()
    rl = np.ones((8, 2), dtype=np.uint64)
    assert lib.vp_phase1_init(h, 2, rl.ctypes.data, rl.ctypes.data) == 0
    zero = (ctypes.c_uint64 * 2)(0, 0)
    assert lib.vp_round(h, zero, poly) == 0
    for bad in ((P, 0), (0, P), (1 << 61, 5), (7, (1 << 64) - 1)):
        rb = (ctypes.c_uint64 * 2)(*bad)
        assert lib.vp_round(h, rb, poly) == -1
        assert lib.vp_finalize(h, rb, poly, 1) == -1
    assert lib.vp_round(h, r, poly) == 0                             # the sumcheck is still alive
    s.close(); c.close()


def test_index_split_refuses_more_than_eight_slices(vp):
    """vp_set_shard_split adds the slices' partial sums (< 2^61 each) with a plain u64 all-reduce: 8 addends fit, 16 could wrap mod 2^64
    (= 8 mod p: a silently wrong transcript).  W = 16 is refused (VP_ELIMIT); W = 8 and W = 12 (8 slices) are taken; chain sharding
    alone (disjoint slices) has no such bound."""
    c = vp.Circuit.randomize(3, 12, seed=3)
    s = vp.Session(c)
    lib = vp.lib_gpu()
    lib.vp_set_shard.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    h = s.gpu_ctx()
    for world, rc in ((8, 0), (12, 0), (16, -5), (64, -5)):
        assert lib.vp_set_shard(h, 0, world) == 0
        assert lib.vp_set_shard_split(h, 10) == rc, world
    assert lib.vp_set_shard(h, 3, 32) == 0 and lib.vp_set_shard_split(h, 0) == 0     # no index split: any world
    assert lib.vp_set_shard(h, 0, 1) == 0
    s.close(); c.close()


@pytest.mark.parametrize("mode", ["launch_per_round", "resident", "round1_kernels"])
def test_interactive_path_variants(vp, golden, gold_gkr, pws_path, monkeypatch, mode):
    """The drop-in entry points (vp_round per verifier message) in their forms — one launch per round (VP_PERSIST=0), the resident mailbox
    kernel for the rounds that fit one CU (default), and the init calls through the per-sumcheck kernels of round 1 instead of the batched
    path's (VP_FAST_INIT=0): the real reference's transcript in every case, on SHA-256 x16 (tables up to 2^20, multi-table phase 2) and on a
    ragged small circuit."""
    if mode == "launch_per_round":
        monkeypatch.setenv("VP_PERSIST", "0")
    elif mode == "round1_kernels":
        monkeypatch.setenv("VP_FAST_INIT", "0")
    c = vp.Circuit.from_pws(pws_path, 16, seed=1)
    s = vp.Session(c)
    tr, res, ok = s.prove_interactive()
    assert ok and tr == gold_gkr("sha256_x16")
    tr2, _, ok2 = s.prove_interactive()                 # a second proof on the same context (mailbox sequence numbers keep counting)
    assert ok2 and tr2 == tr
    s.close(); c.close()
    c = vp.Circuit.randomize(8, 12, seed=1)
    s = vp.Session(c)
    tr, _, ok = s.prove_interactive()
    assert ok and tr == gold_gkr("randomize_8_12")
    s.close(); c.close()


def _drive_phase1(vp, h, c, disturb_at=None, disturb=None):
    """Phase 1 of the top layer of `c`, message by message through the C ABI, with fixed challenges: Vres, every round polynomial, the claim."""
    lib = vp.lib_gpu()
    lib.vp_vres.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.vp_phase1_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.vp_round.argtypes = [ctypes.c_void_p] * 3
    lib.vp_finalize.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int]
    top = c.layers - 1
    bl_top, bl_pre = c.layer_bitlen(top), c.layer_bitlen(top - 1)
    rng = np.random.default_rng(12)
    r0 = rng.integers(0, P, size=(max(bl_top, 1), 2), dtype=np.uint64)
    ar = rng.integers(0, P, size=(1, 2), dtype=np.uint64)
    ru = rng.integers(0, P, size=(bl_pre, 2), dtype=np.uint64)
    out = np.zeros((1, 2), dtype=np.uint64)
    assert lib.vp_vres(h, r0.ctypes.data, bl_top, out.ctypes.data) == 0
    msgs = [out.tobytes()]
    assert lib.vp_phase1_init(h, top, r0.ctypes.data, ar.ctypes.data) == 0
    prev = np.zeros((1, 2), dtype=np.uint64)
    poly = np.zeros((3, 2), dtype=np.uint64)
    for j in range(bl_pre):
        if disturb_at == j:
            disturb()
        rc = lib.vp_round(h, prev.ctypes.data, poly.ctypes.data)
        assert rc == 0, (j, lib.vp_last_error(h))
        msgs.append(poly.tobytes())
        prev = ru[j:j + 1].copy()
    if disturb_at == bl_pre:
        disturb()
    claim = np.zeros((1, 2), dtype=np.uint64)
    assert lib.vp_finalize(h, prev.ctypes.data, claim.ctypes.data, 1) == 0, lib.vp_last_error(h)
    msgs.append(claim.tobytes())
    return b"".join(msgs)


@pytest.mark.parametrize("where", [0, 1, 4, 9, 10])
def test_resident_kernel_yields_to_another_context_and_resumes(vp, where):
    """One process, two contexts (the shape the sharded tests and any multi-prover host have): context A is in the middle of a sumcheck, its
    resident round kernel waiting for the next challenge, when context B is created, uploads a circuit, proves and is destroyed — hipMalloc /
    hipFree, device-wide synchronising calls that used to wait behind A's kernel until its 10 s time-out, after which A's phase was lost.
    Now B's entry points suspend A's kernel (phase saved to device memory), B proceeds at once, and A's next vp_round / vp_finalize
    relaunches the kernel on the saved phase: A's messages are bit-identical to an undisturbed run, wherever the interruption falls —
    before the first vp_round (round 1 already answered behind the init call), mid-phase, before the last round, before finalize."""
    import time
    c = vp.Circuit.randomize(3, 10, seed=21)
    s = vp.Session(c)
    h = s.gpu_ctx()
    undisturbed = _drive_phase1(vp, h, c)
    took = []

    def other_context():
        t0 = time.perf_counter()
        c2 = vp.Circuit.randomize(4, 9, seed=5)
        s2 = vp.Session(c2)                      # vp_create + vp_circuit_upload (hipMalloc) + vp_evaluate on a SECOND context
        s2.draw_tape()
        tr, _ = s2.prove_gkr()
        tr_i, _, ok = s2.prove_interactive()     # B runs its own resident kernels meanwhile
        assert ok and tr_i == tr
        s2.close(); c2.close()                   # vp_destroy: hipFree
        took.append(time.perf_counter() - t0)

    disturbed = _drive_phase1(vp, h, c, disturb_at=where, disturb=other_context)
    assert disturbed == undisturbed
    assert took and took[0] < 5.0, "the other context waited %.1f s behind the resident kernel" % took[0]
    again = _drive_phase1(vp, h, c)              # and the context is as good as new
    assert again == undisturbed
    s.close(); c.close()


def test_contexts_on_different_threads(vp, gold_gkr, pws_path):
    """include/vpgpu.h, "Threads": calls on one context are serialised by the context's lock, different contexts run concurrently.  Three
    threads at once on the x16 circuit: two INTERACTIVE proofs (each phase's resident round kernel is suspended whenever the other thread's
    init call — or the third thread's session set-up — needs the device, and resumes), and a thread that creates a session, proves in
    batched mode and destroys it, over and over.  The verifier's challenges come from the process-wide random() stream, as in the reference,
    so with three threads drawing from it the transcripts are not the golden ones: every proof has to pass the full verification instead
    (per-round identities, wiring predicates, Liu and input checks), and the sessions used afterwards, alone, reproduce the golden transcript."""
    import threading
    c = vp.Circuit.from_pws(pws_path, 16, seed=1)
    gold = gold_gkr("sha256_x16")
    out, err = {}, []
    stop = threading.Event()
    ready = threading.Barrier(3)

    def interactive(name):
        try:
            s = vp.Session(c)
            ready.wait(timeout=120)                        # set-up calls take turns; the proofs below are what has to overlap
            n = 0
            while n < 3 or not stop.is_set():
                tr, _, ok = s.prove_interactive()
                assert ok and len(tr) == len(gold), name
                n += 1
                if n >= 400:
                    break
            out[name] = (s, s.tail_resumes())
        except BaseException as e:                         # noqa: B036 - the main thread reports it
            err.append((name, repr(e)))

    def churn():
        try:
            n = 0
            ready.wait(timeout=120)
            while n < 6:
                s = vp.Session(c)
                s.draw_tape()
                tr, _ = s.prove_gkr()
                ok, _ = s.check(tr, device_predicates=True)
                assert ok and len(tr) == len(gold)
                s.close()
                n += 1
            out["churn"] = n
        except BaseException as e:                         # noqa: B036
            err.append(("churn", repr(e)))
        finally:
            stop.set()

    th = [threading.Thread(target=interactive, args=("a",)), threading.Thread(target=interactive, args=("b",)), threading.Thread(target=churn)]
    for t in th:
        t.start()
    th[2].join(timeout=300)
    stop.set()
    for t in th[:2]:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in th), "a thread is stuck (lock cycle?)"
    assert not err, err
    assert out["churn"] >= 1 and out["a"][1] + out["b"][1] > 0, out    # the resident kernels really were suspended and resumed
    for name in ("a", "b"):                                            # alone again: the reference's stream, the reference's transcript
        s = out[name][0]
        s2 = vp.Session(c)
        tr, _, ok = s2.prove_interactive()
        assert ok and tr == gold
        s2.close(); s.close()
    c.close()


def test_resident_kernel_times_out_saves_its_phase_and_resumes(vp):
    """A verifier that falls silent (debugger, SIGSTOP, a loaded host) for longer than persistent_timeout_ms: the resident kernel saves its
    phase and releases the CU by itself; when the verifier comes back the phase continues — it used to be lost (VP_EHIP)."""
    import time
    c = vp.Circuit.randomize(3, 10, seed=22)
    s_ref = vp.Session(c)
    undisturbed = _drive_phase1(vp, s_ref.gpu_ctx(), c)
    s_ref.close()
    s = vp.Session(c, options=vp.Options(persistent_timeout_ms=150))
    for where in (2, 10):
        assert _drive_phase1(vp, s.gpu_ctx(), c, disturb_at=where, disturb=lambda: time.sleep(0.6)) == undisturbed
    s.close(); c.close()


def test_abandoned_sumcheck_releases_the_resident_kernel(vp, gold_gkr):
    """A caller that stops in the middle of a sumcheck (the resident round kernel is waiting for the next challenge) and calls any
    other entry point: the kernel is told to leave first, nothing hangs, and the context proves correctly afterwards."""
    c = vp.Circuit.randomize(8, 12, seed=1)
    s = vp.Session(c)
    L = vp.lib_gpu()
    ctx = vp.lib_host().vph_session_ctx(s.h)
    L.vp_vres.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int, ctypes.c_void_p]
    L.vp_phase1_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.vp_round.argtypes = [ctypes.c_void_p] * 3
    L.vp_finalize.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int]
    rng = np.random.default_rng(4)
    r = np.ascontiguousarray(rng.integers(0, P, size=(16, 2), dtype=np.uint64))
    one = np.array([1, 0], dtype=np.uint64)
    out = np.zeros((3, 2), dtype=np.uint64)
    assert L.vp_vres(ctx, r.ctypes.data, 12, out.ctypes.data) == 0
    assert L.vp_phase1_init(ctx, 7, r.ctypes.data, one.ctypes.data) == 0
    zero = np.zeros(2, dtype=np.uint64)
    assert L.vp_round(ctx, zero.ctypes.data, out.ctypes.data) == 0
    for k in range(3):                                   # the table (2^12 entries) fits one CU from round 2 on: the kernel is resident now
        assert L.vp_round(ctx, r[k].ctypes.data, out.ctypes.data) == 0
    assert L.vp_phase1_init(ctx, 7, r.ctypes.data, one.ctypes.data) == 0        # abandon: start the phase again
    assert L.vp_round(ctx, zero.ctypes.data, out.ctypes.data) == 0
    a = out.copy()
    assert L.vp_round(ctx, r[0].ctypes.data, out.ctypes.data) == 0
    tr, _, ok = s.prove_interactive()                    # abandon again, through the host API this time
    assert ok and tr == gold_gkr("randomize_8_12")
    s.draw_tape()
    assert s.prove_gkr()[0] == tr
    s.close(); c.close()


def test_real_value_products_are_a_pure_specialisation(vp, golden, gold_gkr, pws_path, monkeypatch):
    """Round 1 of every sumcheck multiplies by circuit values; when vp_evaluate finds them all real (SHA-256: always) the fold, the
    phase-1 init and the V_u inner product take the half-price real x complex products.  Same transcript with the general
    products forced (VP_REAL_V=0), and a circuit with complex constants (custom_circuits: Mulc/Addc constants have imaginary
    parts, so its values are complex and the flag must come out 0) is covered by test_all_gate_types_and_assert_gates_vs_oracle."""
    c = vp.Circuit.from_pws(pws_path, 16, seed=1)
    for rv in ("1", "0"):
        monkeypatch.setenv("VP_REAL_V", rv)
        s = vp.Session(c)
        s.draw_tape()
        tr, _ = s.prove_gkr()
        assert tr == gold_gkr("sha256_x16"), "VP_REAL_V=" + rv
        s.close()
    c.close()


def test_options_struct_selects_the_same_alternatives_as_the_test_environment(vp, gold_gkr, pws_path):
    """vp_options (include/vpgpu.h) is the ABI for the library's switches; the VP_* environment variables the other tests flip are a
    test-only override on top of it.  A few alternatives selected through the struct alone: same transcript, and the launch table
    shows that the alternative actually ran."""
    c = vp.Circuit.from_pws(pws_path, 16, seed=1)
    gold = gold_gkr("sha256_x16")
    d = vp.Options()
    assert d.struct_size == ctypes.sizeof(vp.Options) and d.abi == vp.VP_OPTIONS_ABI and d.real_values == 1 and d.persistent_rounds == 1
    drivers = bool(vp.lib_gpu().vp_test_drivers())
    launches = {}
    for name, kw in (("default", {}), ("keep_y", {"drop_y": 0}), ("complex_products", {"real_values": 0}),
                     ("lanes", {"gkr_path": vp.PATH_LANES}), ("simple", {"gkr_path": vp.PATH_SIMPLE}), ("no_graph", {"use_graph": 0, "serial": 1}),
                     ("copy_engine", {"kernel_copies": 0}), ("blocking_wait", {"poll": 0}), ("fixed_layout", {"plan_autotune": 0}),
                     ("combine_node", {"plan_autotune": 0, "fuse_combine": 0}), ("one_fold_stream", {"plan_autotune": 0, "fold_branches": 0})):
        if name in ("lanes", "simple") and not drivers:
            with pytest.raises(RuntimeError):                    # the product library ships the launch plan alone: vp_create refuses the other drivers
                vp.Session(c, options=vp.Options(**kw))
            continue
        s = vp.Session(c, options=vp.Options(**kw))
        s.draw_tape()
        tr, res = s.prove_gkr()
        assert tr == gold, name
        launches[name] = res["launches"]
        if name == "default":
            tr_i, _, ok = s.prove_interactive()
            assert ok and tr_i == gold
        s.close()
    # the alternatives really ran: one launch per round on the simple path, one stream per chain (no batched nodes) on the lanes path
    if drivers:
        assert launches["simple"] > 10 * launches["default"] and launches["lanes"] > launches["default"]
    # (the default session picks its plan layout by measurement on the first proof: compare launch counts against the fixed layout)
    assert launches["no_graph"] == launches["fixed_layout"]
    assert launches["combine_node"] > launches["fixed_layout"]
    s = vp.Session(c, options=vp.Options(persistent_rounds=0))
    tr_i, _, ok = s.prove_interactive()
    assert ok and tr_i == gold
    s.close()
    # the plan tuner (on by default) chooses only among the fields the caller left at their defaults: what was set explicitly is still in
    # effect after the first proof (vp_get_options), whatever else the tuner picked
    s = vp.Session(c, options=vp.Options(fuse_combine=1, sf3b_grid=448, fuse_min_log=21))
    s.draw_tape()
    tr, _ = s.prove_gkr()
    eff = s.options_in_effect()
    assert tr == gold and eff.plan_autotune == 1 and (eff.fuse_combine, eff.sf3b_grid, eff.fuse_min_log) == (1, 448, 21)
    s.close()
    # a struct of another layout (a caller built against another header) is refused, not read field by field at shifted offsets
    for bad in ("size", "abi"):
        old = vp.Options()
        if bad == "size":
            old.struct_size = 8
        else:
            old.abi = 0
        with pytest.raises(RuntimeError):
            vp.Session(c, options=old)
    c.close()


_DRIVER_TESTS = ("test_sha256_transcript_matches_reference or test_randomize_transcript_matches_reference or test_small_and_ragged or test_all_gate_types_and_assert_gates_vs_oracle "
                 "or test_all_gate_types_full_protocol_vs_reference or test_options_struct_selects_the_same_alternatives_as_the_test_environment")


def test_cross_check_drivers_flavour(vp):
    """The launch plan's two cross-check drivers (VP_GKR_PATH=lanes: the plan's recorder run live, one stream per sumcheck chain; =simple: one launch per
    round through the interactive kernels) are compiled into a tests flavour only (-DVP_TEST_DRIVERS, tools/_build/testdrv; the product library refuses
    VP_GKR_PATH).  The tests that compare them with the plan run here under that flavour, in a fresh process."""
    import subprocess, sys
    if vp.lib_gpu().vp_test_drivers():
        pytest.skip("already running under the drivers flavour")
    assert os.path.exists(vp.LIB_GPU_TESTDRV), "vp.build() did not produce the drivers flavour"
    env = dict(os.environ, VP_LIBGPU=vp.LIB_GPU_TESTDRV)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu", "-k", _DRIVER_TESTS],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


def test_reference_binary_drives_the_device_prover(pws_path, tmp_path, golden):
    """oracle/_ref/ref_run_vpgpu = the reference's own main() + verifier + circuit code + lib/virgo (minus fri.cpp), unmodified, linked
    with INTEGRATION.md's three forwarding files (prover_vpgpu.cpp for src/prover.cpp, fri_vpgpu.cpp for lib/virgo/src/fri.cpp,
    fft_gkr_vpgpu.cpp for fft_circuit_GKR.cpp's fft_gkr) and
    libvpgpu.so (tests/test_integration_link.py builds it where the reference tree exists).  On SHA256_64.pws every sumcheck message,
    both Merkle roots, input_0 / all_sum, every FRI root, the final codeword and all 33 x (2 + 7) openings with their Merkle paths come
    from the device; the reference's verifier (src/verifier.cpp) and lib/virgo's verify_poly_commitment (vpd_verifier.cpp:76-328) check
    them and print "Verification pass".  The messages the forwarding files handed over are ALSO the real reference's own, byte for byte:
    the dump in the golden layout equals tests/golden/transcript_sha256_x1.bin, the FRI dump equals fri_sha256_x1.bin."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_run_vpgpu")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_run_vpgpu not built (needs the reference tree at build time)")
    dump, dump_fri, dump_fft = tmp_path / "messages.bin", tmp_path / "fri.bin", tmp_path / "fft.bin"
    env = dict(os.environ, VPI_DUMP=str(dump), VPI_DUMP_FRI=str(dump_fri), VPI_DUMP_FFT=str(dump_fft), VPI_TRACE="1")
    r = subprocess.run([exe, str(pws_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Verification pass" in r.stderr and "Verification fail" not in r.stderr
    assert "Prove Time" in r.stdout and "Polynomial commitment: prove time" in r.stdout
    # who served the reference's verifier: 439 rounds, 1 + 1 commitments, n - 6 = 7 FRI steps, 33 query repetitions x (l, h) and x 7 levels
    m = re.search(r"vpgpu calls: commit_private (\d+) commit_public (\d+) fri_step (\d+) fri_final (\d+) open_init (\d+) open_step (\d+) round (\d+) "
                  r"finalize (\d+) rand_consumers (\d+) fft_gkr (\d+)", r.stderr)
    assert m, r.stderr[-2000:]
    # rand_consumers = 0: no vp_* call took draws from the process's random() generator (the library keeps the ROCm runtime's own draws
    # away from it; the verifier's challenges come from that stream, fieldElement.cpp:119-124)
    assert [int(x) for x in m.groups()] == [1, 1, 7, 1, 66, 231, 439, 42, 0, 1]
    g = golden["sha256_x1"]
    assert dump.read_bytes() == open(os.path.join(GOLDEN_DIR, g["transcript"]), "rb").read()
    assert dump_fri.read_bytes() == open(os.path.join(GOLDEN_DIR, g["fri"]), "rb").read()
    # fft_gkr (fft_gkr_vpgpu.cpp in place of lib/virgo's fft_circuit_GKR.cpp): the device's messages = the CPU reference's record of that call
    assert dump_fft.read_bytes() == open(os.path.join(GOLDEN_DIR, "fftgkr_sha256_x1.bin"), "rb").read()
    assert "fft gkr failed" not in r.stderr


@pytest.mark.parametrize("blocks", [16, 64])
def test_reference_verifier_accepts_device_commitment_at_block_counts(pws_path, tmp_path, golden, blocks):
    """The same seam at x16 and x64 (the reference's main() fixes repeat = 1; oracle/integration/blocks_main.cpp replaces main() only and
    feeds the B-fold DAG to the reference's own DAG_to_layered / subsetInit): the unmodified reference verifier — GKR and
    verify_poly_commitment with its 33 query repetitions — accepts a proof whose every message, root, opening and Merkle path comes from the
    device, and the messages handed over equal the CPU reference's transcript and FRI record of the same circuit and seed, byte for byte."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_run_vpgpu_blocks")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_run_vpgpu_blocks not built (needs the reference tree at build time)")
    dump, dump_fri = tmp_path / "messages.bin", tmp_path / "fri.bin"
    env = dict(os.environ, VPI_DUMP=str(dump), VPI_DUMP_FRI=str(dump_fri), VPI_TRACE="1")
    r = subprocess.run([exe, str(pws_path), str(blocks)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200, env=env)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    assert "Verification pass" in r.stderr and "Verification fail" not in r.stderr and "ok 1" in r.stdout
    m = re.search(r"vpgpu calls: commit_private (\d+) commit_public (\d+) fri_step (\d+) fri_final (\d+) open_init (\d+) open_step (\d+) round (\d+) "
                  r"finalize (\d+) rand_consumers (\d+) fft_gkr (\d+)", r.stderr)
    assert m, r.stderr[-2000:]
    calls = [int(x) for x in m.groups()]
    g = golden["sha256_x%d" % blocks]
    assert calls[0] == 1 and calls[1] == 1 and calls[3] == 1 and calls[4] == 66 and calls[8] == 0 and calls[9] == 1
    assert calls[5] == 33 * calls[2]                                  # one opening per query repetition and FRI level
    assert dump.read_bytes() == open(os.path.join(GOLDEN_DIR, g["transcript"]), "rb").read()
    assert dump_fri.read_bytes() == open(os.path.join(GOLDEN_DIR, g["fri"]), "rb").read()


def test_tensor_public_vector_shortcut_is_exact(vp, ob, monkeypatch):
    """vp_commit_public encodes ONE slice when the public vector is a tensor (every slice a multiple of slice 0 — the protocol's eq table
    always is) and forms q_i = c_i q_0 where it is consumed.  Same roots, input_0 and all_sum as the general path (VP_PC_TENSOR=0) on: an eq
    table; a tensor with generic factors; a tensor whose corner is zero and a near-tensor with one entry changed (both must take the
    general path: the device check is exact); FRI roots after either path."""
    rng = np.random.default_rng(31)
    c = vp.Circuit.randomize(3, 11, seed=4)
    n = c.layer_bitlen(0)
    N = 1 << (n - 6)

    def run(pub, tensor):
        monkeypatch.setenv("VP_PC_TENSOR", "1" if tensor else "0")
        s = vp.Session(c)
        root_l, _ = s.commit_private()
        root_h, inner, all_sum, _ = s.commit_public(pub)
        rr = np.random.default_rng(5).integers(0, P, size=(n - 6, 2), dtype=np.uint64)
        roots, fin = s.fri_commit(rr)
        s.close()
        return root_l, root_h, inner, all_sum, roots, fin.tobytes()

    s0 = vp.Session(c)
    eq = s0.eq_table(rng.integers(0, P, size=(n, 2), dtype=np.uint64))
    s0.close()
    lo, hi = rng.integers(0, P, size=(N, 2), dtype=np.uint64), rng.integers(0, P, size=(64, 2), dtype=np.uint64)
    L = ob.lib()
    generic = np.zeros((64 * N, 2), dtype=np.uint64)
    for i in range(64):
        for k in range(N):
            L.orc_f_mul(hi[i].ctypes.data, lo[k].ctypes.data, generic[i * N + k].ctypes.data)
    zero_corner = generic.copy(); zero_corner[0] = 0
    near = generic.copy(); near[17 * N + 5, 0] ^= np.uint64(1)
    for name, pub in (("eq table", eq), ("generic tensor", generic), ("zero corner", zero_corner), ("near-tensor", near)):
        assert run(pub, True) == run(pub, False), name
    c.close()


def _run_ranks(world, transport, blocks, timeout=600):
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_sharded_worker.py"), transport, str(blocks)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, env=env)


def test_two_processes_share_a_sharded_proof_and_commitment_host_transport(vp):
    """Two PROCESSES (torch.distributed.run, world size 2, both on this box's GPU): each holds the x16 instance, proves only the chains dealt to
    it (vp_set_shard), the transcript is assembled by one all-reduce (gloo); then the commitment with 32 slices per rank, its all-to-all
    and all-gather moved by the host transport (vp_shard_exchange_get / _put + gloo).  Everything against the real reference's golden
    data, on every rank.  The data path of a real multi-GPU run differs only in the transport (RCCL inside the C ABI: next test)."""
    r = _run_ranks(2, "host", 16)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "RANK 0 OK" in r.stdout and "RANK 1 OK" in r.stdout


def _bench_two_ranks(extra_env, timeout_s, *flags):
    """bench.py --gpus 2 as the driver starts it (torch.distributed.run, one process per rank), both ranks on this box's GPU (gloo control plane,
    host transport: the rehearsal mode)."""
    import subprocess, sys
    env = dict(os.environ, VP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29577",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--blocks", "16", "--no-cpu-baseline"] + list(flags)
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s, env=env, cwd=ROOT)


@pytest.mark.parametrize("inject", [None, "raise:1", "hang:1", "die:1"])
def test_bench_multi_rank_line_and_exit_code_when_a_rank_fails(vp, tmp_path, inject):
    """The multi-rank run of bench.py prints exactly ONE parseable line whatever happens in its `sharded` sub-leg, and its exit code tells the
    launcher when the run was not whole: a rank that raises before the first collective (the others learn at the flag exchange), a rank that never
    arrives (the watchdog prints what there is and every rank leaves with code 3), a rank that is killed (the launcher ends rank 0 with SIGTERM — or
    rank 0 sees the peer vanish at the flag exchange, whichever comes first: either way it prints its line before it leaves), and the healthy run
    (code 0, sub-leg bit-exact)."""
    import json
    detail = str(tmp_path / "detail.json")
    env = {"VP_BENCH_INJECT": inject} if inject else {}
    flags = ["--detail-file", detail] + (["--subleg-timeout", "25"] if inject and inject.startswith("hang") else [])
    r = _bench_two_ranks(env, 600, *flags)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout[-800:], r.stderr[-2500:])
    o = json.loads(lines[0])
    assert len(lines[0].encode()) <= 4096 and o["ranks"] == 2 and o["n_gpus"] == 1      # two ranks on ONE card: a rehearsal never reads as two GPUs
    assert o["bit_exact"] is True and o["config"]["proofs_per_step"] == 2
    if inject is None:
        assert r.returncode == 0, (r.returncode, r.stderr[-2500:])
        assert "multi_gpu_sublegs_error" not in o and o["sharded"]["bit_exact"] is True and o["sharded"]["commitment_bit_exact"] is True
    else:
        assert r.returncode != 0
        assert "multi_gpu_sublegs_error" in o and any(w in o["multi_gpu_sublegs_error"] for w in ("failed", "timed out", "went away", "terminated"))


_EXIT_WORKER = r"""
import ctypes, os, sys
sys.path.insert(0, %r)
order = sys.argv[1]
if order == "torch_first":
    import torch
    torch.cuda.device_count()
import vp_loader
vp = vp_loader.load()
L = vp.lib_gpu()
ctx = ctypes.c_void_p()
assert L.vp_create(0, ctypes.byref(ctx)) == 0
uid = ctypes.create_string_buffer(128)
assert L.vp_comm_unique_id(ctypes.cast(uid, ctypes.c_void_p)) == 0
assert L.vp_comm_init(ctx, ctypes.cast(uid, ctypes.c_void_p), 0, 1) == 0, L.vp_last_error(ctx)
n = ctypes.c_int(0)
assert L.vp_comm_count(ctx, ctypes.byref(n)) == 0 and n.value == 1
if order == "library_first":
    import torch
    torch.cuda.device_count()
L.vp_destroy(ctx)
print("DONE", order, flush=True)
"""


@pytest.mark.parametrize("order", ["torch_first", "library_first"])
def test_process_with_pytorch_and_the_librarys_rccl_exits_cleanly(vp, order):
    """bench.py's ranks import torch and then attach the library's RCCL communicator; a test process may do it the other way round.  Either
    way the process has to END with status 0 (a rank that aborts in a library destructor after printing its line still fails the job):
    torch first = one ROCm stack shared under the same sonames; library first = two stacks, ours outside the global symbol scope."""
    import subprocess, sys
    r = subprocess.run([sys.executable, "-c", _EXIT_WORKER % ROOT, order], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert "DONE " + order in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])


def test_two_gpus_rccl_sharded_proof_and_commitment(vp):
    """The same over RCCL / xGMI with one GPU per rank: the transcript all-reduce (also of an index-split proof's export area) and the
    commitment's collectives are RCCL calls on device buffers inside the C ABI (vp_comm_init).  Needs two GPUs: skipped on the builder's
    one-GPU box, run wherever the suite finds them."""
    from conftest import gpu_count
    if gpu_count() < 2:
        pytest.skip("needs 2 GPUs (this box has %d)" % gpu_count())
    r = _run_ranks(2, "rccl", 64)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "RANK 0 OK" in r.stdout and "RANK 1 OK" in r.stdout


def test_many_small_circuits_interactive_and_batched_vs_oracle(vp, ob):
    """A sweep over 150 random circuits of odd shapes (2-9 layers, 1-300 gates per layer, every gate type, assert gates, complex
    constants; and `randomize` circuits with layer sizes 2^0..2^7): the drop-in path (resident round kernel, round 1 queued at init)
    and the batched plan must both reproduce the oracle's transcript — single-entry tables, zero- and one-round phases, tables that
    retire in the middle of a resident phase, empty subsets."""
    import custom_circuits as cc
    rng = np.random.default_rng(2024)
    n_bad = 0
    for it in range(150):
        if it % 3 == 2:
            layers, lg = int(rng.integers(2, 8)), int(rng.integers(0, 8))
            c = vp.Circuit.randomize(layers, lg, seed=100 + it); oc = ob.Circuit.randomize(layers, lg, seed=100 + it)
            what = "randomize(%d, %d, seed %d)" % (layers, lg, 100 + it)
        else:
            sizes = [int(x) for x in rng.integers(1, 300 if it % 2 else 12, size=int(rng.integers(2, 10)))]
            args = cc.make(1000 + it, sizes)
            c = vp.Circuit.custom(*args); oc = ob.Circuit.custom(*args)
            what = "custom(seed %d, %s)" % (1000 + it, sizes)
        assert c.hash() == oc.hash(), what
        gold, st = oc.prove_gkr()
        assert st["verified"] == 1, what
        s = vp.Session(c)
        tr, _, ok = s.prove_interactive()
        s.draw_tape()
        tr2, _ = s.prove_gkr()
        if not (ok and tr == gold and tr2 == gold):
            n_bad += 1
            print("MISMATCH:", what, "interactive ok/equal", ok, tr == gold, "batched equal", tr2 == gold)
        s.close(); c.close(); oc.close()
    assert n_bad == 0


def test_violated_assert_gate_is_reported(vp):
    """The reference exits the process when an assert gate is non-zero (src/prover.cpp:18-21); the library returns
    VP_EASSERT through the host constructor instead."""
    import custom_circuits as cc
    sizes, ty, l, u, v, c, a = cc.make(7, [16, 16, 8], with_asserts=False)
    a[20] = 1                                   # some layer-1 gate with a (surely) non-zero value
    circ = vp.Circuit.custom(sizes, ty, l, u, v, c, a)
    with pytest.raises(RuntimeError, match="assert"):
        vp.Session(circ)
    circ.close()


@pytest.mark.parametrize("name", ["custom_a", "custom_b"])
def test_all_gate_types_full_protocol_vs_reference(vp, golden, name):
    """The real reference's transcript (GKR + commitment) and FRI steps on circuits with every gate type and assert gates."""
    import os
    import custom_circuits as cc
    from conftest import GOLDEN
    g = golden[name]
    c = vp.Circuit.custom(*cc.make(g["custom"]["seed"], g["custom"]["sizes"]))
    assert c.hash() == g["circuit_hash"]
    gold = open(os.path.join(GOLDEN, g["transcript"]), "rb").read()
    _both_modes(vp, c, gold[g["gkr_slice"][0]:g["gkr_slice"][1]])
    s = vp.Session(c)
    for batched in (False, True):
        tr, ok = s.prove_full(batched=batched)
        assert ok and tr == gold
    r, roots_gold, fin_gold = _fri_golden(golden, name)
    roots, fin = s.fri_commit(r)
    assert roots == roots_gold and np.array_equal(fin, fin_gold)
    s.close(); c.close()


@pytest.mark.parametrize("name,blocks", [("sha256_x1", 1), ("sha256_x16", 16)])
def test_complete_protocol_with_commitment_verification(vp, golden, pws_path, name, blocks):
    """verifier::verify() end to end (src/verifier.cpp:134-189): commit_private, interactive GKR, commit_public, FRI
    commit phase, 33 query repetitions answered by vp_fri_open and checked by the host verifier.  ONE unbroken run, no recorded
    challenge injected: the transcript up to all_sum, every FRI Merkle root, the fold challenges and the final codeword equal what
    the real reference recorded (the host verifier consumes the draws of the reference's fft_gkr, vpd_verifier.cpp:92, so its FRI
    challenges are the reference's)."""
    import os
    from conftest import GOLDEN
    c = vp.Circuit.from_pws(pws_path, blocks, seed=1)
    s = vp.Session(c)
    tr, ok, times = s.prove_and_verify_full(reps=33)
    assert ok
    assert tr == open(os.path.join(GOLDEN, golden[name]["transcript"]), "rb").read()
    r_gold, roots_gold, fin_gold = _fri_golden(golden, name)
    roots, fin, r = s.last_fri()
    assert np.array_equal(r, r_gold), "FRI fold challenges differ from the reference's"
    assert roots == roots_gold and np.array_equal(fin, fin_gold)
    if name == "sha256_x1":
        # fft_gkr INSIDE the protocol (vpd_verifier.cpp:92), at the stream position the reference runs it: the device's messages equal the
        # real reference's record of that very call (tests/golden/fftgkr_sha256_x1.bin), and its time is part of the commitment's prove time
        assert s.last_fft_gkr() == open(os.path.join(GOLDEN, "fftgkr_sha256_x1.bin"), "rb").read()
        assert 0 < times["pc_fft_gkr_sec"] < times["pc_prove_sec"]
    s.close(); c.close()


def test_complete_protocol_custom_gates(vp, golden):
    import os
    import custom_circuits as cc
    from conftest import GOLDEN
    g = golden["custom_b"]
    c = vp.Circuit.custom(*cc.make(g["custom"]["seed"], g["custom"]["sizes"]))
    s = vp.Session(c)
    tr, ok, _ = s.prove_and_verify_full(reps=8)
    assert ok and tr == open(os.path.join(GOLDEN, g["transcript"]), "rb").read()
    r_gold, roots_gold, fin_gold = _fri_golden(golden, "custom_b")
    roots, fin, r = s.last_fri()
    assert np.array_equal(r, r_gold) and roots == roots_gold and np.array_equal(fin, fin_gold)
    s.close(); c.close()


def test_cli_runs_the_reference_flow(vp, pws_path):
    """virgo_plus_run <file.pws> (the reference's command line, script/run.sh:11): whole protocol on the GPU, the
    reference's result lines on stdout."""
    import subprocess
    out = subprocess.run([vp.CLI, pws_path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "Verification pass" in out.stderr
    for key in ("Input size 7226", "Prove Time", "verify time", "proof size", "Polynomial commitment: prove time"):
        assert key in out.stdout


@pytest.mark.gpu
def test_cli_fiat_shamir_mode(vp, pws_path, tmp_path):
    """virgo_plus_run --fs: non-interactive proof, verified from the dumped bytes alone by a second verifier object and by the library."""
    import subprocess
    dump = tmp_path / "proof.bin"
    out = subprocess.run([vp.CLI, pws_path, "--fs", "--seed", "1", "--dump", str(dump)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "Verification pass" in out.stderr and "proof size" in out.stdout
    c = vp.Circuit.from_pws(pws_path, 1, seed=1)  # the same witness as the CLI drew
    assert c.verify_fs(dump.read_bytes())
    c.close()


def test_plan_tuner_choice_is_shared_between_sessions_of_a_circuit(vp, pws_path):
    """The plan tuner runs once per plan shape and process: a second session of the same circuit takes the first one's choice (same options in
    effect, a much cheaper first proof) and proves the same transcript; a session with a field pinned by the caller tunes for itself.  (x5: a
    block count no other test of this process proves — the table is process-wide.)"""
    import time
    c = vp.Circuit.from_pws(pws_path, 5, seed=1)
    first, opts, trs = [], [], []
    for k in range(2):
        s = vp.Session(c)
        s.draw_tape()
        t0 = time.perf_counter()
        tr, _ = s.prove_gkr()
        first.append(time.perf_counter() - t0)
        assert s.check(tr, device_predicates=True)[0]
        trs.append(tr)
        o = s.options_in_effect()
        opts.append(tuple(getattr(o, f) for f in ("fuse_combine", "fold_branches", "plan_align", "fuse_min_log", "sf3b_grid", "graph_explicit")))
        s.close()
    assert trs[0] == trs[1] and opts[0] == opts[1]
    assert first[1] < 0.6 * first[0], first
    s = vp.Session(c, options=vp.Options(sf3b_grid=448))          # another plan shape as far as the tuner is concerned
    s.draw_tape()
    tr, _ = s.prove_gkr()
    assert tr == trs[0] and s.options_in_effect().sf3b_grid == 448
    s.close()
    # the choice carried by hand (what another PROCESS would do): vp_plan_tuning_get on a tuned context, vp_plan_tuning_set on a fresh one before its first
    # proof; a layout nobody measured here is taken as given too, and the transcript does not depend on it
    L = vp.lib_gpu()
    L.vp_plan_tuning_get.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)]; L.vp_plan_tuning_set.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)]
    for lay in (None, (0, 0, 1, 19, 320, 0)):
        s = vp.Session(c, options=vp.Options(plan_autotune=1))
        v = (ctypes.c_int32 * 6)(*(lay if lay else opts[0]))
        assert L.vp_plan_tuning_set(s.gpu_ctx(), v) == 0
        s.draw_tape()
        tr, _ = s.prove_gkr()
        got = (ctypes.c_int32 * 6)()
        assert L.vp_plan_tuning_get(s.gpu_ctx(), got) == 0
        assert tr == trs[0] and tuple(got) == tuple(v)
        assert L.vp_plan_tuning_set(s.gpu_ctx(), v) != 0              # too late: the plan is recorded
        s.close()
    c.close()


_CHECKED_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import vp_loader
vp = vp_loader.load()
vp.lib_host()
assert vp.lib_gpu().vp_checked_build() == 1, "VP_LIBGPU did not select the checked library"
pws, gold_path, inject = sys.argv[1], sys.argv[2], sys.argv[3] == "1"
c = vp.Circuit.from_pws(pws, 1, seed=1)
s = vp.Session(c)
if inject:
    s.draw_tape()
    try:
        s.prove_gkr()
    except RuntimeError as e:
        print("REFUSED:", e, flush=True)
        sys.exit(0)
    print("NOT REFUSED", flush=True)
    sys.exit(1)
gold = open(gold_path, "rb").read()
tr, ok = s.prove_full(batched=True)
assert ok and tr == gold, "checked build: batched protocol transcript differs"
tr_i, _, ok_i = s.prove_interactive()
assert ok_i and tr_i == gold[32:32 + len(tr_i)], "checked build: interactive transcript differs"
s.draw_protocol_tape()
tr_p, roots, fin, sec = s.prove_protocol()
assert tr_p == gold
print("CHECKED OK", flush=True)
"""


@pytest.mark.parametrize("inject", [False, True])
def test_checked_build_passes_the_protocol_and_reports_a_violated_index_check(vp, golden, pws_path, inject):
    """The -DVP_CHECKED flavour of libvpgpu.so (csrc/vp_check.h; built by vp.build() under tools/_build/checked, loaded through VP_LIBGPU in a
    fresh process): device-side index checks in the gather / scatter kernels.  With them compiled in, the whole protocol at x1 — batched proof with
    the commitment, interactive proof, the one-pass prover — still reproduces the real reference's transcript and no check fires; with layer 0's
    bound shrunk to one wire (VP_CHECKED_INJECT) the first proof is refused with VP_EHIP and the site of the violated check, without a trap."""
    import subprocess, sys
    assert os.path.exists(vp.LIB_GPU_CHECKED), "vp.build() did not produce the checked library"
    env = dict(os.environ, VP_LIBGPU=vp.LIB_GPU_CHECKED)
    if inject:
        env["VP_CHECKED_INJECT"] = "1"
    gold = os.path.join(GOLDEN_DIR, golden["sha256_x1"]["transcript"])
    r = subprocess.run([sys.executable, "-c", _CHECKED_WORKER % ROOT, pws_path, gold, "1" if inject else "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-2000:])
    if inject:
        assert "REFUSED" in r.stdout and "device check failed: site" in r.stdout
    else:
        assert "CHECKED OK" in r.stdout


# ---- round 5 -------------------------------------------------------------------------------------------------------------------------------------
def test_commit_private_after_pc_load_input_of_complex_data_on_an_evaluated_context(vp, pws_path):
    """a context that evaluated a REAL witness (vreal = 1: vp_commit_private pairs two slices per transform and reads .re only) and is then handed
    COMPLEX inputs through vp_pc_load_input must commit to all of them: the root equals a fresh context's, bit length 19 (the paired path's sizes)."""
    L = vp.lib_gpu()
    c = vp.Circuit.from_pws(pws_path, 64, seed=1)          # input layer 2^19
    s = vp.Session(c)
    r_real, _ = s.commit_private()
    n = c.layer_bitlen(0)
    rng = np.random.default_rng(31)
    inp = rng.integers(0, P, size=(1 << n, 2), dtype=np.uint64)
    ctx = s.gpu_ctx()
    assert L.vp_pc_load_input(ctx, inp.ctypes.data, inp.shape[0], n) == 0
    root = ctypes.create_string_buffer(32)
    assert L.vp_commit_private(ctx, ctypes.cast(root, ctypes.c_void_p)) == 0, L.vp_last_error(ctx)
    fresh = ctypes.c_void_p()
    assert L.vp_create(0, ctypes.byref(fresh)) == 0
    assert L.vp_pc_load_input(fresh, inp.ctypes.data, inp.shape[0], n) == 0
    root2 = ctypes.create_string_buffer(32)
    assert L.vp_commit_private(fresh, ctypes.cast(root2, ctypes.c_void_p)) == 0
    L.vp_destroy(fresh)
    assert root.raw == root2.raw and root.raw != r_real
    s.close(); c.close()


@pytest.mark.parametrize("blocks", [16, 256])
def test_protocol_pass_forms_give_the_same_bytes(vp, golden, pws_path, blocks):
    """vph_prove_protocol_ex (vphost.h): synchronous calls | deferred completion (vp_set_deferred: every call queued without a host wait, collected by vp_flush) |
    deferred + the next pass's commit_private queued behind this pass's FRI folds — the same transcript, FRI roots and final codeword (x16: the real reference's),
    pass after pass, and the context serves a synchronous entry point afterwards (what is pending is finished first)."""
    c = vp.Circuit.from_pws(pws_path, blocks, seed=1)
    s = vp.Session(c)
    s.draw_protocol_tape()
    ref = s.prove_protocol(deferred=False)
    if blocks == 16:
        g = golden["sha256_x16"]
        assert ref[0] == open(os.path.join(GOLDEN_DIR, g["transcript"]), "rb").read()
        fri = open(os.path.join(GOLDEN_DIR, g["fri"]), "rb").read()
        assert ref[1] == b"".join(fri[48 * k + 16:48 * k + 48] for k in range(g["fri_steps"]))
    same = lambda a: a[0] == ref[0] and a[1] == ref[1] and np.array_equal(a[2], ref[2])
    for _ in range(2):
        assert same(s.prove_protocol(deferred=True))
    for _ in range(3):
        assert same(s.prove_protocol(queue_next=True))
    n = ctypes.c_int(-1)
    assert vp.lib_gpu().vp_pending(s.gpu_ctx(), ctypes.byref(n)) == 0 and n.value == 1          # the next pass's commit_private
    assert same(s.prove_protocol(deferred=True))                                                 # finds it, starts at the GKR part
    assert vp.lib_gpu().vp_pending(s.gpu_ctx(), ctypes.byref(n)) == 0 and n.value == 0
    assert same(s.prove_protocol(queue_next=True))
    tr_i, _, ok_i = s.prove_interactive()                                                        # a synchronous entry point with a head pending
    assert ok_i and tr_i == ref[0][32:32 + len(tr_i)]
    s.draw_protocol_tape()
    assert same(s.prove_protocol(deferred=False))
    root, _ = s.commit_private()
    assert root == ref[0][:32]
    s.close(); c.close()


def test_deferred_calls_return_at_once_and_flush_fills_the_outputs(vp, pws_path):
    """The C ABI of the deferred mode on its own: with vp_set_deferred(ctx, 1) vp_commit_private returns before its root is there (the buffer is untouched until
    vp_flush), vp_pending counts it, vp_flush(ctx, -1) writes the synchronous call's bytes; a vp_evaluate in between finishes what is pending by itself."""
    L = vp.lib_gpu()
    L.vp_set_deferred.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.vp_flush.argtypes = [ctypes.c_void_p, ctypes.c_int]
    c = vp.Circuit.from_pws(pws_path, 64, seed=1)
    s = vp.Session(c)
    ctx = s.gpu_ctx()
    want, _ = s.commit_private()
    assert L.vp_set_deferred(ctx, 1) == 0
    root = ctypes.create_string_buffer(b"\xee" * 32, 32)
    n = ctypes.c_int(0)
    assert L.vp_commit_private(ctx, ctypes.cast(root, ctypes.c_void_p)) == 0
    assert L.vp_pending(ctx, ctypes.byref(n)) == 0 and n.value == 1
    assert root.raw == b"\xee" * 32
    assert L.vp_flush(ctx, -1) == 0
    assert root.raw == want
    root2 = ctypes.create_string_buffer(b"\xee" * 32, 32)
    assert L.vp_commit_private(ctx, ctypes.cast(root2, ctypes.c_void_p)) == 0
    vals = s.layer_values(1)                                    # any other entry point: finishes the pending call first
    assert root2.raw == want and len(vals)
    assert L.vp_pending(ctx, ctypes.byref(n)) == 0 and n.value == 0
    assert L.vp_set_deferred(ctx, 0) == 0
    s.close(); c.close()


def test_fft_gkr_begun_and_never_collected_does_not_wedge_the_context(vp, ctx):
    """a pass that fails between vp_fft_gkr_begin and vp_fft_gkr_end used to leave every later begin on the context failing with "not collected".
    Now vp_fft_gkr_cancel drops such a run, and a second begin without it drains and drops the stale one by itself; the messages of the run that follows are
    the synchronous call's."""
    L = vp.lib_gpu()
    L.vp_fft_gkr_cancel.argtypes = [ctypes.c_void_p]
    lg = 7
    nt, nm = ctypes.c_uint64(0), ctypes.c_uint64(0)
    assert L.vp_fft_gkr_sizes(lg, ctypes.byref(nt), ctypes.byref(nm)) == 0
    tape = np.random.default_rng(5).integers(0, P, size=(nt.value, 2), dtype=np.uint64)
    want = np.zeros((nm.value, 2), dtype=np.uint64); w = ctypes.c_uint64(0)
    assert L.vp_fft_gkr(ctx, lg, tape.ctypes.data, nt.value, want.ctypes.data, nm.value, ctypes.byref(w)) == 0
    assert L.vp_fft_gkr_begin(ctx, lg, tape.ctypes.data, nt.value) == 0
    assert L.vp_fft_gkr_begin(ctx, lg, tape.ctypes.data, nt.value) == 0              # the first run is dropped, not an error
    assert L.vp_fft_gkr_cancel(ctx) == 0
    assert L.vp_fft_gkr_cancel(ctx) == 0                                             # nothing pending: still VP_OK
    got = np.zeros_like(want)
    assert L.vp_fft_gkr_end(ctx, got.ctypes.data, nm.value, ctypes.byref(w)) != 0    # nothing to collect
    assert L.vp_fft_gkr_begin(ctx, lg, tape.ctypes.data, nt.value) == 0
    assert L.vp_fft_gkr_end(ctx, got.ctypes.data, nm.value, ctypes.byref(w)) == 0
    assert np.array_equal(got, want)


def test_hundreds_of_deferred_calls_without_a_flush_keep_their_bytes(vp, pws_path):
    """the pinned staging ring (8 MB) used to wrap over regions that still-pending deferred calls had staged their results in — ~256 deferred
    vp_fri_final calls (32 KB each) without a vp_flush returned VP_OK with the bytes of LATER calls.  Now a request that would run into live regions finishes the
    queued calls first (and the queue itself is capped): 600 calls, every output the synchronous call's bytes."""
    L = vp.lib_gpu()
    L.vp_fri_final.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    c = vp.Circuit.from_pws(pws_path, 16, seed=1)
    s = vp.Session(c)
    s.draw_protocol_tape()
    s.prove_protocol()
    _, fin, _ = s.last_fri()
    want = np.ascontiguousarray(fin).tobytes()
    ctx = s.gpu_ctx()
    n_calls = 600
    outs = [np.full((2048, 2), 0xEE, dtype=np.uint64) for _ in range(n_calls)]
    assert L.vp_set_deferred(ctx, 1) == 0
    for o in outs:
        assert L.vp_fri_final(ctx, o.ctypes.data) == 0
    n = ctypes.c_int(0)
    assert L.vp_pending(ctx, ctypes.byref(n)) == 0 and 0 < n.value <= 256          # the queue is capped; the ring made earlier calls finish
    assert L.vp_flush(ctx, -1) == 0
    assert L.vp_set_deferred(ctx, 0) == 0
    bad = [i for i, o in enumerate(outs) if o.tobytes() != want]
    assert not bad, "deferred calls with wrong bytes: %s" % bad[:10]
    s.close(); c.close()


def test_vp_warm_changes_no_byte_and_plan_cache_file_round_trips(vp, golden, pws_path, tmp_path, monkeypatch):
    """vp_warm(VP_WARM_COMMITMENT) sets up the commitment's tables / buffers / pinned staging ahead of the first prover call (include/vpgpu.h): the pass behind it and
    vp_commit_public from the caller's (pageable) vector through the pinned staging give the reference's bytes.  VP_PLAN_CACHE: the first session writes the
    tuner's choice to the file, a second context (plan table of the process cleared by using another circuit size is not needed: the line is there) reads it."""
    g = golden["sha256_x64"]
    gold = open(os.path.join(ROOT, "tests", "golden", g["transcript"]), "rb").read()
    cache = tmp_path / "plan_cache.txt"
    monkeypatch.setenv("VP_PLAN_CACHE", str(cache))
    c = vp.Circuit.from_pws(pws_path, 64, seed=1)
    s = vp.Session(c)
    s.warm(); s.warm()                                               # idempotent
    s.draw_protocol_tape()
    tr, roots, fin, _ = s.prove_protocol()
    assert tr == gold
    lines = [l.split() for l in open(cache).read().splitlines() if l.strip()]
    assert len(lines) >= 1 and all(len(l) == 7 for l in lines)      # key + six values
    # commit_public with the vector handed over from host memory (the reference's boundary): staged through pinned memory in pieces
    pub = s.eq_table(s.last_point())
    s.commit_private()
    root_h, inner, all_sum, _ = s.commit_public(pub)
    assert root_h + inner + all_sum == gold[len(gold) - (32 + 16 + 65 * 16):]
    s.close()
    s2 = vp.Session(c)                                               # the tuner's choice comes from the table / file: same bytes
    s2.draw_tape()
    tr2, _ = s2.prove_gkr()
    assert tr2 == gold[g["gkr_slice"][0]:g["gkr_slice"][1]]
    s2.close(); c.close()


def _pc_masked_record(name):
    import pc_masked_inputs as pmi
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "pc_masked.json")))[name]
    rec = open(os.path.join(ROOT, "tests", "golden", meta["record"]), "rb").read()
    fri = open(os.path.join(ROOT, "tests", "golden", meta["fri"]), "rb").read()
    return pmi.inputs(name), meta, rec, fri


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["n13_zero", "n13_m5", "n13_m64", "n16_m100", "n19_m3000", "n13_m300", "n13_m2000"])
def test_commitment_with_mask_slices_vs_real_reference(vp, name):
    """SURVEY 8a a12 with CONTENT in the 65th slice (lib/virgo/src/poly_commit.h:42,55-86,138-161,187-191): the reference's own prover only ever passes one zero
    (src/prover.cpp:526), so the goldens come from poly_commit_prover::commit_private_array / commit_public_array / commit_phase called directly by
    oracle/_ref/ref_run --pc-masked (tests/golden/make_pc_masked.py).  Roots of both oracles, all_sum[65] (the mask slice's term last), every FRI root, the final
    codeword, the mask slice's final codeword and openings of both oracles and of two FRI levels (65 pairs each, the mask slice's last) — byte for byte.  `n13_zero`
    runs the UNMASKED entry points against a record of the same layout."""
    x, meta, rec, fri = _pc_masked_record(name)
    L = vp.lib_gpu()
    n, st = x["n"], meta["fri_steps"]
    M = 1 << (n - 1)
    ctx = ctypes.c_void_p()
    assert L.vp_create(0, ctypes.byref(ctx)) == 0
    try:
        vals = np.ascontiguousarray(x["values"])
        assert L.vp_pc_load_input(ctx, vals.ctypes.data, vals.shape[0], n) == 0
        root_l, root_h = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
        inner, alls = np.zeros(2, np.uint64), np.zeros((65, 2), np.uint64)
        pub = np.ascontiguousarray(x["pub"])
        if name.endswith("_zero"):
            assert L.vp_commit_private(ctx, ctypes.cast(root_l, ctypes.c_void_p)) == 0
            assert L.vp_commit_public(ctx, pub.ctypes.data, pub.shape[0], inner.ctypes.data, alls.ctypes.data, ctypes.cast(root_h, ctypes.c_void_p)) == 0
        else:
            L.vp_commit_private_masked.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
            L.vp_commit_public_masked.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
            pm, qm = np.ascontiguousarray(x["pri_mask"]), np.ascontiguousarray(x["pub_mask"])
            rc = L.vp_commit_private_masked(ctx, pm.ctypes.data, pm.shape[0], ctypes.cast(root_l, ctypes.c_void_p))
            assert rc == 0, L.vp_last_error(ctx)
            rc = L.vp_commit_public_masked(ctx, pub.ctypes.data, pub.shape[0], qm.ctypes.data, qm.shape[0], inner.ctypes.data, alls.ctypes.data, ctypes.cast(root_h, ctypes.c_void_p))
            assert rc == 0, L.vp_last_error(ctx)
        assert root_l.raw == rec[:32], "merkle_root_l"
        assert root_h.raw == rec[32:64], "merkle_root_h"
        assert alls.tobytes() == rec[64:64 + 65 * 16], "all_sum"
        at = 64 + 65 * 16
        vals130 = np.zeros((130, 2), np.uint64); path = ctypes.create_string_buffer(32 * 40); plen = ctypes.c_int(0)
        L.vp_fri_open.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        for oracle in (0, 1):
            for leaf in (0, 5, M // 2 - 1):
                assert L.vp_fri_open(ctx, oracle, leaf, vals130.ctypes.data, ctypes.cast(path, ctypes.c_void_p), len(path), ctypes.byref(plen)) == 0
                assert vals130.tobytes() == rec[at:at + 130 * 16], ("opening", oracle, leaf)
                at += 130 * 16
        rr = np.frombuffer(b"".join(fri[48 * k:48 * k + 16] for k in range(st)), dtype=np.uint64).reshape(st, 2).copy()
        roots = ctypes.create_string_buffer(32 * st)
        rc = L.vp_fri_commit(ctx, rr.ctypes.data, st, ctypes.cast(roots, ctypes.c_void_p))
        assert rc == 0, L.vp_last_error(ctx)
        assert roots.raw == b"".join(fri[48 * k + 16:48 * k + 48] for k in range(st)), "FRI roots"
        fin = np.zeros((2048, 2), np.uint64)
        assert L.vp_fri_final(ctx, fin.ctypes.data) == 0
        assert fin.tobytes() == fri[48 * st:48 * st + 2048 * 16], "final codeword"
        L.vp_fri_final_mask.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        fm = np.zeros((32, 2), np.uint64)
        assert L.vp_fri_final_mask(ctx, fm.ctypes.data) == 0
        assert fm.tobytes() == fri[48 * st + 2048 * 16:48 * st + 2048 * 16 + 32 * 16], "final codeword of the mask slice"
        for lvl in (0, 2):
            assert L.vp_fri_open(ctx, 2 + lvl, 3, vals130.ctypes.data, ctypes.cast(path, ctypes.c_void_p), len(path), ctypes.byref(plen)) == 0
            assert vals130.tobytes() == rec[at:at + 130 * 16], ("FRI level opening", lvl)
            at += 130 * 16
        assert at == len(rec)
    finally:
        L.vp_destroy(ctx)


@pytest.mark.parametrize("name", ["n13_m5", "n13_m64", "n16_m100", "n19_m3000", "n13_m300"])
def test_reference_verifier_decides_device_masked_commitments(name, tmp_path):
    """The mask slice end to end behind the reference's OWN verifier: oracle/_ref/ref_run_vpgpu_masked (integration/masked_main.cpp) commits with
    vp_commit_private_masked / vp_commit_public_masked and hands the roots and all_sum[65] to the unmodified poly_commit_verifier::verify_poly_commitment
    (vpd_verifier.cpp:76-328) with the public mask — 33 random queries, every opening, Merkle path, fold and final codeword (the mask slice's among them) served
    from HBM through INTEGRATION.md's forwarding files.  The verdict must be the one the reference's verifier gives the reference's own prover
    (tests/test_oracle_golden.py, pc_masked_inputs.REFERENCE_VERIFIER_ACCEPTS): ACCEPT up to a slice's message length, the same last-check REJECT beyond it."""
    import subprocess
    import pc_masked_inputs as pmi
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_run_vpgpu_masked")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_run_vpgpu_masked not built (needs the reference tree at build time)")
    inp = tmp_path / "in.bin"
    pmi.write_case_file(name, str(inp))
    r = subprocess.run([exe, str(inp)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=dict(os.environ, VPI_TRACE="1"))
    want = pmi.REFERENCE_VERIFIER_ACCEPTS[name]
    assert ("verify_poly_commitment ACCEPT" in r.stdout) == want and ("verify_poly_commitment REJECT" in r.stdout) == (not want), (r.stdout[-400:], r.stderr[-1500:])
    assert r.returncode == (0 if want else 1)
    if want:
        st = pmi.CASES[name][0] - 6
        assert "open_init 66 open_step %d " % (33 * st) in r.stderr and "rand_consumers 0" in r.stderr, r.stderr[-1500:]
    else:
        assert "Fri msk rs code check fail" in r.stderr


def test_masked_commitment_limits_and_state(vp):
    """The masked entry points at their edges: a mask that pads to fewer than 8 elements is refused (the reference's own transforms of that size read stale scratch),
    one longer than half a slice too (mask_position_gap 1), one that pads to more than 2^16 elements is VP_ELIMIT, the unmasked public calls refuse a masked private commitment, the public mask may not outgrow the private
    one's padded length, and vp_commit_private returns the context to the zero mask (same root as before)."""
    import pc_masked_inputs as pmi
    x = pmi.inputs("n13_m5")
    L = vp.lib_gpu()
    L.vp_commit_private_masked.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
    L.vp_commit_public_masked.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    ctx = ctypes.c_void_p()
    assert L.vp_create(0, ctypes.byref(ctx)) == 0
    try:
        vals = np.ascontiguousarray(x["values"]); pub = np.ascontiguousarray(x["pub"])
        assert L.vp_pc_load_input(ctx, vals.ctypes.data, vals.shape[0], 13) == 0
        root0, root1, root2 = (ctypes.create_string_buffer(32) for _ in range(3))
        assert L.vp_commit_private(ctx, ctypes.cast(root0, ctypes.c_void_p)) == 0
        m = np.ones((200, 2), np.uint64)
        assert L.vp_commit_private_masked(ctx, m.ctypes.data, 1, ctypes.cast(root1, ctypes.c_void_p)) == -1 and b"fewer than 8" in L.vp_last_error(ctx)    # VP_EINVAL
        assert L.vp_commit_private_masked(ctx, m.ctypes.data, 4, ctypes.cast(root1, ctypes.c_void_p)) == -1
        m3 = np.ones((3000, 2), np.uint64)
        assert L.vp_commit_private_masked(ctx, m3.ctypes.data, 3000, ctypes.cast(root1, ctypes.c_void_p)) == -1 and b"half a slice" in L.vp_last_error(ctx)  # gap 1: the reference asserts against it
        inner, alls = np.zeros(2, np.uint64), np.zeros((65, 2), np.uint64)
        assert L.vp_commit_public_masked(ctx, pub.ctypes.data, pub.shape[0], m.ctypes.data, 5, inner.ctypes.data, alls.ctypes.data, ctypes.cast(root2, ctypes.c_void_p)) == -1   # no masked private commitment
        pm = np.ascontiguousarray(x["pri_mask"])
        assert L.vp_commit_private_masked(ctx, pm.ctypes.data, pm.shape[0], ctypes.cast(root1, ctypes.c_void_p)) == 0 and root1.raw != root0.raw
        assert L.vp_commit_public(ctx, pub.ctypes.data, pub.shape[0], inner.ctypes.data, alls.ctypes.data, ctypes.cast(root2, ctypes.c_void_p)) == -1             # wants vp_commit_public_masked
        assert L.vp_commit_public_masked(ctx, pub.ctypes.data, pub.shape[0], m.ctypes.data, 9, inner.ctypes.data, alls.ctypes.data, ctypes.cast(root2, ctypes.c_void_p)) == -1   # 9 > padded length 8
        assert L.vp_commit_private(ctx, ctypes.cast(root2, ctypes.c_void_p)) == 0 and root2.raw == root0.raw
        assert L.vp_commit_public(ctx, pub.ctypes.data, pub.shape[0], inner.ctypes.data, alls.ctypes.data, ctypes.cast(root2, ctypes.c_void_p)) == 0
        big = np.zeros((1 << 19, 2), np.uint64); big[:, 0] = 1
        assert L.vp_pc_load_input(ctx, big.ctypes.data, big.shape[0], 19) == 0
        mb = np.ones((100000, 2), np.uint64)
        assert L.vp_commit_private_masked(ctx, mb.ctypes.data, 100000, ctypes.cast(root1, ctypes.c_void_p)) == -5 and b"2^16" in L.vp_last_error(ctx)        # VP_ELIMIT: pads to 2^17
    finally:
        L.vp_destroy(ctx)

