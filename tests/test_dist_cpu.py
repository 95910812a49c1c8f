"""CPU: the N>1 control plane of bench.py (one process per rank, barrier, max-over-ranks of the elapsed
time, sum of the proofs) with world_size 2 over gloo.  The data path itself has no collective."""
import os
import subprocess
import sys

from conftest import ROOT

WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import bench
world, rank, local = bench.dist_setup(2)
assert world == 2 and rank == int(os.environ["RANK"])
bench.barrier(world)
elapsed, proofs = bench.aggregate(world, 1.0 + rank, 5.0)
assert abs(elapsed - 2.0) < 1e-12 and abs(proofs - 10.0) < 1e-12, (elapsed, proofs)
bench.barrier(world)
import torch.distributed as dist
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_two_rank_aggregation_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


def test_single_rank_aggregate_is_identity():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.aggregate(1, 3.5, 7.0) == (3.5, 7.0)


SHARD_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np
import bench, vp_loader
vp = vp_loader.load()
world, rank, local = bench.dist_setup(2)
gold = open(%r, "rb").read()
gold = gold[: len(gold) // 16 * 16]
# rank r holds every second 48-byte message of the transcript, zeros elsewhere: the layout a chain-sharded proof produces
a = np.frombuffer(gold, dtype=np.uint64).copy()
msg = (np.arange(a.size) // 6) %% 2
a[msg != rank] = 0
full = vp.allreduce_transcript(a.tobytes())
assert full == gold, "assembled transcript differs"
bench.barrier(world)
import torch.distributed as dist
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharded_transcript_allreduce_gloo(tmp_path, golden):
    """The one data-path collective of a chain-sharded proof (virgo-plus_amd.allreduce_transcript): two ranks hold disjoint
    slices of a golden transcript, the all-reduce (gloo here, RCCL on the GPUs) returns the whole transcript on both."""
    import os as _os
    from conftest import GOLDEN
    script = tmp_path / "w.py"
    script.write_text(SHARD_WORKER % (ROOT, _os.path.join(GOLDEN, golden["sha256_x1"]["transcript"])))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29534")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_sum_transcripts_is_u64_wraparound_sum(vp):
    import numpy as np
    a = np.array([2 ** 64 - 1, 0, 5], dtype=np.uint64).tobytes()
    b = np.array([1, 7, 0], dtype=np.uint64).tobytes()
    assert np.frombuffer(vp.sum_transcripts([a, b]), dtype=np.uint64).tolist() == [0, 7, 5]


PC_SHARD_WORKER = r"""
import ctypes, hashlib, os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import torch, torch.distributed as dist
import bench, oracle_binding as ob
world, rank, local = bench.dist_setup(2)
W, n = 2, 9                                  # input layer of 2^9 wires: 64 slices of N = 8, codewords of M = 256, 128 leaves
N, M, S = 1 << (n - 6), 1 << (n - 1), 64 // W
L = ob.lib()
oc = ob.Circuit.randomize(2, n, seed=5)
inp = np.zeros((1 << n, 2), np.uint64)
L.orc_circuit_inputs(oc.h, inp.ctypes.data)
# reference point: the oracle's unsharded commit_private (pinned to the real reference's Merkle roots)
root = ctypes.create_string_buffer(32)
L.orc_commit_private.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
assert L.orc_commit_private(oc.h, root) == 0
# 1. this rank encodes ITS slices (poly_commit.h:101-107: iFFT of the slice, then evaluation on the 2^(n-1)-th roots)
L.orc_ifft.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
L.orc_fft.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
mine = np.zeros((S, M, 2), np.uint64)
for sl in range(S):
    s = rank * S + sl
    coef = np.zeros((N, 2), np.uint64)
    L.orc_ifft(np.ascontiguousarray(inp[s * N:(s + 1) * N]).ctypes.data, N, coef.ctypes.data)
    L.orc_fft(coef.ctypes.data, N, M, mine[sl].ctypes.data)
# 2. all-to-all to position ownership: position j = 32 a + b goes to rank a mod W (both halves a and a + N/2 share the residue)
a_of = np.arange(M) >> 5
send = [np.ascontiguousarray(mine[:, (a_of %% W) == d]) for d in range(W)]            # [S][M/W] per destination
got = [torch.zeros(send[0].shape, dtype=torch.int64) for _ in range(W)]
for src in range(W):                                                                  # gloo has no all_to_all: W broadcasts of the block meant for me
    for dst in range(W):
        t = torch.from_numpy(send[dst].view(np.int64).copy()) if src == rank else torch.zeros(send[0].shape, dtype=torch.int64)
        dist.broadcast(t, src)
        if dst == rank:
            got[src] = t
loc = np.concatenate([g.numpy().view(np.uint64) for g in got], axis=0)                # [64][M/W][2], slices in rank order
pos = np.nonzero((a_of %% W) == rank)[0]                                              # the global positions I own, ascending
at = {int(j): k for k, j in enumerate(pos)}
# 3. my leaves (fri.cpp:96-124: 64 slices + the zero mask pair, chained), then five local tree levels (32 cosets of one position)
node5 = []
for a in range(rank, N // 2, W):
    lv = []
    for b in range(32):
        j = 32 * a + b
        h = bytes(32)
        for s in range(64):
            h = hashlib.sha3_256(loc[s, at[j]].tobytes() + loc[s, at[j + M // 2]].tobytes() + h).digest()
        h = hashlib.sha3_256(bytes(32) + h).digest()
        lv.append(h)
    while len(lv) > 1:
        lv = [hashlib.sha3_256(lv[2 * i] + lv[2 * i + 1]).digest() for i in range(len(lv) // 2)]
    node5.append(lv[0])
# 4. all-gather the level-5 nodes, put them in global order (a = a' W + rank), build the top on every rank
mine5 = torch.tensor(list(b"".join(node5)), dtype=torch.uint8)
allg = [torch.zeros_like(mine5) for _ in range(W)]
dist.all_gather(allg, mine5)
per = len(node5)
lvl = [None] * (per * W)
for r in range(W):
    raw = bytes(allg[r].tolist())
    for k in range(per):
        lvl[k * W + r] = raw[32 * k:32 * k + 32]
while len(lvl) > 1:
    lvl = [hashlib.sha3_256(lvl[2 * i] + lvl[2 * i + 1]).digest() for i in range(len(lvl) // 2)]
assert lvl[0] == root.raw, "sharded assembly does not reproduce the unsharded Merkle root"
bench.barrier(world)
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharded_commitment_assembly_gloo(tmp_path):
    """The ownership scheme of the sharded commitment (virgo-plus_amd/csrc/vpgpu_pc_shard.inc) on the CPU with two gloo ranks and the
    oracle's transforms: slices per rank -> exchange to position ownership (a mod W) -> local leaf chains and five tree levels ->
    all-gather of the level-5 nodes -> top of the tree; the assembled root equals the oracle's unsharded commit_private root."""
    script = tmp_path / "w.py"
    script.write_text(PC_SHARD_WORKER % (ROOT, ROOT))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29535")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
