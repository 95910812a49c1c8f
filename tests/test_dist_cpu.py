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
