"""One rank of a multi-process sharded proof + sharded commitment (run under torch.distributed.run by tests/test_gpu_parity.py):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node W --master-addr 127.0.0.1 --master-port P tests/dist_sharded_worker.py TRANSPORT BLOCKS
TRANSPORT = rccl   one GPU per rank: the transcript all-reduce and the commitment's all-to-all / all-gather are RCCL calls inside the C ABI
          = host   ranks share GPUs (the one-GPU box): transcript through torch/gloo, commitment collectives through the host transport
                   (vp_shard_exchange_get / _put) — the rank logic of a multi-PROCESS run with the data path on the CPU.
Every rank holds the same instance (seed 1); results are compared with the real reference's golden data.  Prints `RANK r OK`."""
import gzip
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    transport, blocks = sys.argv[1], int(sys.argv[2])
    import vp_loader
    vp = vp_loader.load()
    vp.lib_host()
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    ndev = torch.cuda.device_count()
    dev = local % ndev
    if transport == "rccl":
        assert ndev >= world, "RCCL needs one GPU per rank"
        torch.cuda.set_device(dev)
        dist.init_process_group(backend="cpu:gloo,cuda:nccl", rank=rank, world_size=world)
    else:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
    g = golden["sha256_x%d" % blocks]
    gold_full = open(os.path.join(ROOT, "tests", "golden", g["transcript"]), "rb").read()
    gold = gold_full[g["gkr_slice"][0]:g["gkr_slice"][1]]
    with tempfile.TemporaryDirectory() as tmp:
        pws = os.path.join(tmp, "SHA256_64.pws")
        with gzip.open(os.path.join(ROOT, "tests", "golden", "SHA256_64.pws.gz"), "rb") as f, open(pws, "wb") as o:
            o.write(f.read())
        circ = vp.Circuit.from_pws(pws, blocks, seed=1)
    sess = vp.Session(circ, device=dev)
    sess.draw_tape()
    full, ok = sess.prove_full(batched=True)
    assert ok and full == gold_full, "unsharded proof differs from the reference"
    inputs, pub, n_bits = sess.layer_values(0), sess.eq_table(sess.last_point()), circ.layer_bitlen(0)
    # ---- the proof, chains dealt to the ranks, ONE all-reduce
    sess.set_shard(rank, world)
    if transport == "rccl":
        sess.attach_comm(rank, world)
        assert sess.comm_count() == world
        tr, _ = sess.prove_gkr()                       # all-reduced inside the call
        assert tr == gold, "chain-sharded proof (RCCL) differs"
        sess.set_shard_split(11)                       # long chains cut by index: same collective, export area included
        tr, _ = sess.prove_gkr()
        assert tr == gold, "index-split proof (RCCL) differs"
        # the same index-split proof without the V_u exchange ahead of the graph (VP_SPLIT_VU=0: every rank adds up the whole layer; no extra
        # all-reduce) — both forms over the same communicator, the same bytes
        os.environ["VP_SPLIT_VU"] = "0"
        s0 = vp.Session(circ, device=dev)
        del os.environ["VP_SPLIT_VU"]
        s0.draw_tape()                                  # the same tape: F::init() reseeds (fieldElement.cpp:362-367)
        s0.set_shard(rank, world); s0.attach_comm(rank, world); s0.set_shard_split(11)
        tr0, _ = s0.prove_gkr()
        assert tr0 == gold, "index-split proof (RCCL, VP_SPLIT_VU=0) differs"
        s0.close()
    else:
        part, _ = sess.prove_gkr()
        assert part != gold                            # this rank's slices only
        tr = vp.allreduce_transcript(part)
        assert tr == gold, "chain-sharded proof (gloo) differs"
    sess.close(); circ.close()
    # ---- the commitment, slices dealt to the ranks
    st = n_bits - 6
    fri = open(os.path.join(ROOT, "tests", "golden", g["fri"]), "rb").read()
    rr = np.frombuffer(b"".join(fri[48 * k:48 * k + 16] for k in range(st)), dtype=np.uint64).reshape(st, 2).copy()
    sc = vp.ShardedCommitmentRank(inputs, n_bits, rank, world, device=dev, transport=transport)
    if transport == "rccl":
        assert sc.comm_count() == world
    root_l = sc.commit_private()
    root_h, inner, all_sum = sc.commit_public(pub)
    roots, fin = sc.fri_commit(rr)
    assert root_l == gold_full[:32], "sharded merkle_root_l differs"
    assert root_h + inner + all_sum == gold_full[len(gold_full) - (32 + 16 + 65 * 16):], "sharded root_h / input_0 / all_sum differ"
    assert roots == b"".join(fri[48 * k + 16:48 * k + 48] for k in range(st)), "sharded FRI roots differ"
    assert fin.tobytes() == fri[48 * st:48 * st + 2048 * 16], "sharded final codeword differs"
    sc.close()
    dist.barrier()
    print("RANK %d OK" % rank, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
