"""The streaming form of the condensed-QP build (Markov vectors in a device workspace): what serves every shape beyond the
on-chip working set of the workgroup kernels, so that mpcx_qp_build has no size wall where form_lifted_matrices
(controller.rs:16-99) has none.

(1) Forced on (MPCX_OPT_QP_STREAMING = 1) for shapes the other kernels also take: against the oracle, and against the
    default dispatch to rounding.  (2) The largest shapes the workgroup kernels hold and the first ones they do not
    (n=12 m=4 r=12: f = v = 127 | 128; n=8 m=2 r=2: f = v = 852 | 853), plus shapes far beyond (quadrotor plant at f = 200,
    n=8 m=2 r=2 at f = 2500), against the oracle.  Tolerance: 1e-10 relative Frobenius per output and per instance."""
import numpy as np
import pytest

from helpers import assert_close, oracle_qp, random_stable_plants
from mpc_rs_amd import ffi, synth
from oracle import oracle_ffi as orc

pytestmark = pytest.mark.gpu
OUTS = ("Phi", "Gamma", "H", "fvec")


@pytest.fixture()
def streaming_ctx():
    c = ffi.Context(0)
    c.set_option(ffi.OPT_QP_STREAMING, 1)
    yield c
    c.close()


def _data(rng, batch, n, m, r, f, v, rho=0.9, banded=False):
    A, B, C = random_stable_plants(rng, batch, n, m, r, rho=rho)
    if banded:
        w3, q = synth.block_weights(m, r, v, 0.3, 0.7, 2.0, all_elements=False)
    else:
        w3, q = rng.standard_normal((v * m, v * m)), rng.standard_normal((r, r))
    return dict(A=A, B=B, C=C, W3=w3, Q=q, x0=rng.standard_normal((batch, n)), ref=rng.standard_normal((batch, f * r)), f=f, v=v)


def _run(c, d, il=1, want=OUTS):
    return c.qp_build(d["A"], d["B"], d["C"], d["f"], d["v"], W3=d["W3"], W4=d["Q"], x0=d["x0"], ref=d["ref"], interleave=il, want=want)


@pytest.mark.parametrize("n,m,r,f,v", [(4, 1, 1, 20, 20), (2, 1, 1, 7, 3), (5, 2, 2, 9, 9), (12, 4, 12, 10, 10), (12, 4, 12, 9, 4),
                                       (8, 2, 2, 40, 40), (8, 2, 2, 30, 7), (7, 3, 5, 6, 6), (9, 1, 1, 12, 2), (16, 5, 3, 5, 1),
                                       (20, 2, 2, 6, 6), (3, 6, 2, 8, 5), (10, 2, 70, 4, 3), (30, 1, 2, 25, 25), (1, 1, 1, 1, 1),
                                       (6, 9, 2, 7, 7), (8, 2, 3, 41, 17)])
def test_streaming_form_against_the_oracle_and_the_default_kernels(ctx, streaming_ctx, n, m, r, f, v):
    rng = np.random.default_rng(n * 13 + m * 7 + r * 3 + f + v)
    d = _data(rng, 7, n, m, r, f, v)
    want = oracle_qp(orc, d)
    got = _run(streaming_ctx, d)
    base = _run(ctx, d)
    for k in OUTS:
        assert_close(got[k], want[k], f"streaming n{n} m{m} r{r} f{f} v{v} {k}")
        assert_close(got[k], base[k], f"streaming vs default n{n} m{m} r{r} f{f} v{v} {k}")
    # exact zeros above Gamma's block diagonal
    for kb in range(1, v):
        assert not got["Gamma"][:, : kb * r, kb * m:(kb + 1) * m].any()
    # output subsets: bit-identical to the full call; an interleaved batch (through the instance-major copy) as well
    for sub in (("Gamma",), ("H",), ("fvec",), ("Phi", "H")):
        part = _run(streaming_ctx, d, want=sub)
        for k in sub:
            assert np.array_equal(part[k], got[k]), f"{k} alone differs from the full call"
    il = _run(streaming_ctx, d, il=64)
    for k in OUTS:
        assert np.array_equal(il[k], got[k]), f"{k}: interleaved batch differs"


@pytest.mark.parametrize("n,m,r,f,v,batch,fast", [
    (12, 4, 12, 127, 127, 3, False), (12, 4, 12, 128, 128, 3, False),   # last shape on chip | first through the workspace
    (8, 2, 2, 852, 852, 2, True), (8, 2, 2, 853, 853, 2, True),
    (12, 4, 12, 200, 200, 2, True), (12, 4, 12, 200, 60, 2, False),      # the quadrotor plant at f = 200
    (8, 2, 2, 2500, 40, 2, False), (8, 2, 2, 2500, 2500, 1, True),        # n=8 m=2 r=2 at f = 2500
])
def test_shapes_at_and_beyond_the_on_chip_limit(ctx, n, m, r, f, v, batch, fast):
    rng = np.random.default_rng(f + v)
    d = _data(rng, batch, n, m, r, f, v, rho=0.97, banded=True)
    # (fast: the oracle's O(f) form with block weights -- itself checked against the reference-structured oracle in
    # tests/test_oracle_cpu.py -- where the reference's O(f v) loop and dense W4 product would take minutes)
    want, _ = orc.qp_build_batch(d["A"], d["B"], d["C"], d["W3"], d["Q"] if fast else synth.dense_w4(d["Q"], f), d["x0"], d["ref"],
                                 f, v, threads=4, fast=fast)
    got = _run(ctx, d)
    for k in OUTS:
        assert_close(got[k], want[k], f"n{n} m{m} r{r} f{f} v{v} {k}")


def test_reserve_covers_the_streaming_workspace():
    """After mpcx_reserve a build beyond the on-chip limit does not allocate."""
    import torch
    from mpc_rs_amd import layout as L
    c = ffi.Context(0)
    try:
        dev = torch.device("cuda", 0)
        n, m, r, f, v, il = 12, 4, 12, 140, 140, 1
        rng = np.random.default_rng(1)
        sp = torch.cuda.current_stream().cuda_stream

        def run(b, reserve):
            d = _data(rng, b, n, m, r, f, v, banded=True)
            t = {k: torch.from_numpy(L.pack(d[k], il)).to(dev) for k in ("A", "B", "C")}
            t["x0"] = torch.from_numpy(L.pack(d["x0"].reshape(b, n, 1), il)).to(dev)
            t["ref"] = torch.from_numpy(L.pack(d["ref"].reshape(b, f * r, 1), il)).to(dev)
            w3, q = torch.from_numpy(L.colmajor(d["W3"])).to(dev), torch.from_numpy(L.colmajor(d["Q"])).to(dev)
            outs = [torch.zeros(L.array_elems(b, il, e), dtype=torch.float64, device=dev)
                    for e in (f * r * n, f * r * v * m, (v * m) ** 2, v * m)]
            dims = ffi.make_dims(n, m, r, f, v, b, 0, il)
            torch.cuda.synchronize()
            if reserve:
                c.reserve(dims)
                torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info(0)[0]
            c.qp_build_dev(dims, t["A"].data_ptr(), t["B"].data_ptr(), t["C"].data_ptr(), w3.data_ptr(), q.data_ptr(),
                           t["x0"].data_ptr(), t["ref"].data_ptr(), *[o.data_ptr() for o in outs], sp)
            torch.cuda.synchronize()
            free1 = torch.cuda.mem_get_info(0)[0]  # (before torch allocates temporaries of its own below)
            assert all(bool(torch.isfinite(o).all()) for o in outs)
            return free0, free1
        run(1, False)
        free0, free1 = run(6, True)
        assert free1 == free0, "a build through the streaming form allocated after mpcx_reserve"
    finally:
        c.close()
