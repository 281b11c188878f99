"""The library's RCCL exchange (mpcx_comm_*, mpcx_gather_stats, mpcx_gather_f64_dev) on one GPU: a one-rank
communicator must return the record mpcx_stats_f64 returns and gather an array unchanged -- from a plain C process
(tests/c_abi/comm_check.c) and in this process through ctypes.  More ranks need more GPUs (RCCL refuses two ranks on
one device); the sharding and merging they would do is covered on CPU by tests/test_sharded_cpu.py."""
import os
import subprocess

import numpy as np
import pytest
import torch

from mpc_rs_amd import ffi, shard

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "c_abi", "comm_check")


def test_plain_c_one_rank_communicator():
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", os.path.join(HERE, "c_abi")], check=True, capture_output=True)
    p = subprocess.run([EXE], capture_output=True, text=True, timeout=180)
    assert p.returncode == 0, (p.stdout, p.stderr)
    assert "comm ok world=1 rank=0 count=100003" in p.stdout and "bad_root rc=-2" in p.stdout


def test_plain_c_rank_of_a_world_through_an_id_file(tmp_path):
    """The multi-process form of the same program (`comm_check WORLD RANK ID_FILE`, one process per GPU on a node) with
    the only world a one-GPU box can form: rank 0 publishes the communicator id as a file, exactly as it does for 8."""
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", os.path.join(HERE, "c_abi")], check=True, capture_output=True)
    idf = tmp_path / "mpcx.id"
    p = subprocess.run([EXE, "1", "0", str(idf), "0"], capture_output=True, text=True, timeout=180)
    assert p.returncode == 0, (p.stdout, p.stderr)
    assert idf.stat().st_size == ffi.COMM_ID_BYTES
    assert "comm ok world=1 rank=0 count=100003" in p.stdout
    bad = subprocess.run([EXE, "2", "2", str(idf)], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 2 and "usage" in bad.stdout


def test_one_rank_communicator_in_process(ctx):
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(9)
    host = rng.standard_normal(70001)
    host[5] = np.inf
    buf = torch.from_numpy(host).to(dev)
    sp = torch.cuda.current_stream().cuda_stream
    want = ctx.stats(buf.data_ptr(), buf.numel())
    comm = ffi.Comm(ctx, ffi.comm_get_id(), 1, 0)
    try:
        recs = comm.gather_stats(buf.data_ptr(), buf.numel(), sp)
        assert len(recs) == 1 and recs[0] == want
        assert shard.merge_stats(recs)["checksum"] == shard.host_stats(host)["checksum"] and recs[0]["non_finite"] == 1
        dst = torch.zeros_like(buf)
        comm.gather_dev(buf.data_ptr(), buf.numel(), dst.data_ptr(), 0, sp)
        comm.synchronize()
        torch.cuda.synchronize()
        assert torch.equal(torch.nan_to_num(dst, posinf=1e300), torch.nan_to_num(buf, posinf=1e300))
        with pytest.raises(ffi.MpcxError) as ei:
            comm.gather_dev(buf.data_ptr(), buf.numel(), dst.data_ptr(), 3, sp)
        assert ei.value.code == ffi.ERR_DIMS
    finally:
        comm.close()
