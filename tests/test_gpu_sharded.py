"""Sharded == unsharded, on one GPU: the arithmetic of a multi-GPU run without the other GPUs.

A batch of `total` plants is cut into `world` contiguous index ranges (shard.shard_range, what rank g of `world` owns).
Every shard is generated from its own start index (synth.make_batch(start=...)), built through mpcx_qp_build_f64_dev
and reduced on the device to one mpcx_stats record per output (mpcx_stats_f64_dev): exactly what a rank does before the
records are exchanged.  The merged records must equal the records of ONE call over the whole batch -- count, non-finite
count, max |.| and the order-free checksum of the bit patterns bitwise, sums to rounding -- because an instance's
plant and results depend on its global index only (controller.rs:16-99: instances share nothing but the weights).
The first and last instance of the last shard (start > 0) are also held against the CPU oracle.

`total` is ragged (world does not divide it, and no shard is a multiple of a wavefront or of the tile width), world is 2
and 8 (config 5's eight ranks).  Interleaved layouts pad their last tile, so records are taken over the instance-major
re-arrangement (mpcx_relayout_f64_dev) of each output: the padding is never part of a record.
"""
import numpy as np
import pytest
import torch

from helpers import RTOL, assert_close, oracle_qp

pytestmark = pytest.mark.gpu

OUTS = ("Phi", "Gamma", "H", "fvec")


def _build_records(ctx, dev, d, count, il, f, v):
    """One mpcx_qp_build_f64_dev call over `d` (count instances, layout il) -> ({output: record}, {output: W=1 tensor})."""
    from mpc_rs_amd import ffi, layout as L, shard
    n, m, r = d["n"], d["m"], d["r"]
    FR, VM = f * r, v * m
    sp = torch.cuda.current_stream().cuda_stream

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    bufs = {k: to_dev(L.pack(d[k], il)) for k in ("A", "B", "C")}
    bufs["x0"] = to_dev(L.pack(d["x0"].reshape(count, n, 1), il))
    bufs["ref"] = to_dev(L.pack(d["ref"].reshape(count, FR, 1), il))
    bufs["W3"], bufs["Q"] = to_dev(L.colmajor(d["W3"])), to_dev(L.colmajor(d["Q"]))
    elems = {"Phi": FR * n, "Gamma": FR * VM, "H": VM * VM, "fvec": VM}
    outs = {k: torch.full((L.array_elems(count, il, e),), float("nan"), dtype=torch.float64, device=dev) for k, e in elems.items()}
    dims = ffi.make_dims(n, m, r, f, v, count, 0, il)
    ctx.qp_build_dev(dims, bufs["A"].data_ptr(), bufs["B"].data_ptr(), bufs["C"].data_ptr(), bufs["W3"].data_ptr(),
                     bufs["Q"].data_ptr(), bufs["x0"].data_ptr(), bufs["ref"].data_ptr(), outs["Phi"].data_ptr(),
                     outs["Gamma"].data_ptr(), outs["H"].data_ptr(), outs["fvec"].data_ptr(), sp)
    recs, flat = {}, {}
    for k, e in elems.items():
        if il == 1:
            flat[k] = outs[k]
        else:
            flat[k] = torch.full((count * e,), float("nan"), dtype=torch.float64, device=dev)
            ctx.relayout_dev(count, e, il, outs[k].data_ptr(), 1, flat[k].data_ptr(), sp)
        rec = torch.zeros(6, dtype=torch.int64, device=dev)
        ctx.stats_dev(flat[k].data_ptr(), count * e, rec.data_ptr(), sp)
        torch.cuda.synchronize()
        recs[k] = shard.stats_from_words(rec.cpu().numpy())
    return recs, flat


CASES = [
    # workload, total instances, interleave, f, v
    ("two_mass", 8 * 4096 + 5, 4096, 20, 20),   # the sweep of config 5, native tiled layout, ragged last tile per shard
    ("two_mass", 8 * 515 + 3, 0, 20, 20),       # W = batch (not a multiple of 64: the non-uniform addressing path)
    ("msd", 8 * 130 + 7, 64, 12, 7),            # run-time horizon lane kernel with held inputs
    ("quadrotor", 8 * 9 + 5, 1, 50, 50),        # one workgroup per instance
    ("long", 8 * 2 + 5, 1, 120, 60),            # workgroup kernel, held inputs
]


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("name,total,il,f,v", CASES)
def test_merged_shard_records_equal_the_one_call_record(ctx, name, total, il, f, v, world):
    from mpc_rs_amd import shard, synth
    from oracle import oracle_ffi as orc
    dev = torch.device("cuda", 0)
    whole_d = synth.make_batch(name, total, f=f, v=v)
    whole, _ = _build_records(ctx, dev, whole_d, total, il, f, v)
    per_shard = []
    covered = 0
    last = None
    for g in range(world):
        start, count = shard.shard_range(total, world, g)
        assert start == covered and count > 0
        covered += count
        d = synth.make_batch(name, count, start=start, f=f, v=v)
        recs, flat = _build_records(ctx, dev, d, count, il, f, v)
        per_shard.append(recs)
        last = (d, count, flat)
    assert covered == total
    for k in OUTS:
        merged = shard.merge_stats([p[k] for p in per_shard])
        assert merged["count"] == whole[k]["count"], k
        assert merged["non_finite"] == whole[k]["non_finite"] == 0, k
        assert merged["checksum"] == whole[k]["checksum"], f"{k}: a shard's results differ in some bit from the one-call run"
        assert merged["max_abs"] == whole[k]["max_abs"], k
        assert merged["sum_sq"] == pytest.approx(whole[k]["sum_sq"], rel=1e-12), k
        assert merged["sum"] == pytest.approx(whole[k]["sum"], rel=1e-9, abs=1e-9 * np.sqrt(whole[k]["sum_sq"])), k
    # the last shard (start > 0): its first and last instance against the oracle
    d, count, flat = last
    pick = [0, count - 1]
    sub = {k: (d[k][pick] if k in ("A", "B", "C", "x0", "ref") else d[k]) for k in d}
    want = oracle_qp(orc, sub)
    n, m, r = d["n"], d["m"], d["r"]
    shapes = {"Phi": (f * r, n), "Gamma": (f * r, v * m), "H": (v * m, v * m), "fvec": (v * m, 1)}
    for k, (rows, cols) in shapes.items():
        got = flat[k].view(count, cols, rows)[pick].transpose(1, 2).cpu().numpy()
        assert_close(got, want[k].reshape(2, rows, cols), f"{name} world {world} last shard {k}", RTOL)


def test_shard_ranges_of_config_5_tile_the_sweep():
    """8,388,608 plants over 8 ranks: contiguous, disjoint, 1,048,576 each (the per-GPU batch of the benchmark)."""
    from mpc_rs_amd import shard
    total = 8 << 20
    edges = [shard.shard_range(total, 8, g) for g in range(8)]
    assert [c for _, c in edges] == [1 << 20] * 8
    assert [s for s, _ in edges] == [g << 20 for g in range(8)]
