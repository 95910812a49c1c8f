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
