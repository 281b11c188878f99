"""The multi-rank path on CPU: two processes (gloo), each owning a contiguous index range of the batch, exchange one
fixed-size statistics record each and rank 0 merges them.  This is the N > 1 flow of bench.py with the GPU kernels
replaced by the oracle, so that the sharding, the record encoding and the collective are exercised without a GPU."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from mpc_rs_amd import shard, synth
    from oracle import oracle_ffi as orc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    start, count = shard.shard_range(total, world, rank)
    d = synth.make_batch("two_mass", count, start=start)
    out, _ = orc.qp_build_batch(d["A"], d["B"], d["C"], d["W3"], synth.dense_w4(d["Q"], d["f"]), d["x0"], d["ref"], d["f"],
                                d["v"], threads=1, want=("H",))
    st = shard.host_stats(out["H"])
    words = np.zeros(6, dtype=np.int64)
    words[0], words[1] = st["count"], st["non_finite"]
    words[2:5] = np.array([st["sum"], st["sum_sq"], st["max_abs"]]).view(np.int64)
    words[5] = np.array([st["checksum"]], dtype=np.uint64).view(np.int64)[0]
    rec = torch.from_numpy(words)
    recs = [torch.zeros_like(rec) for _ in range(world)]
    dist.all_gather(recs, rec)
    dist.barrier()
    if rank == 0:
        q.put(shard.merge_stats([shard.stats_from_words(r.numpy()) for r in recs]))
    dist.destroy_process_group()


def test_two_rank_shards_merge_to_the_single_rank_record():
    import torch.multiprocessing as mp
    from mpc_rs_amd import shard, synth
    from oracle import oracle_ffi as orc
    total, world = 37, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    merged = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = synth.make_batch("two_mass", total)
    out, _ = orc.qp_build_batch(d["A"], d["B"], d["C"], d["W3"], synth.dense_w4(d["Q"], d["f"]), d["x0"], d["ref"], d["f"],
                                d["v"], threads=1, want=("H",))
    whole = shard.host_stats(out["H"])
    assert merged["count"] == whole["count"] == total * 400
    assert merged["checksum"] == whole["checksum"]  # bitwise: a shard's plants and results do not depend on the world size
    assert merged["non_finite"] == 0 and merged["max_abs"] == whole["max_abs"]
    assert merged["sum_sq"] == pytest.approx(whole["sum_sq"], rel=1e-13)
