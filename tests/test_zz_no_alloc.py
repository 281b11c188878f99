"""The allocation-free discretisation entry point (mpcx_c2d_backward_euler_f64_noalloc_dev) against the existing
mpcx_c2d_backward_euler_f64_dev on the same device-resident inputs: same kernels, so A, B and info must be bit-identical
(the bar for this path vs the oracle is 1e-10 relative Frobenius; the existing entry point is held to it in
tests/test_gpu_plants.py).  Two calls in a row, the second smaller, so stale contents of the shared static scratch
would show; then the limits."""
import numpy as np
import pytest
import torch

from mpc_rs_amd import ffi, layout as L

pytestmark = pytest.mark.gpu


def _plants(rng, batch, n, m):
    Ac = rng.standard_normal((batch, n, n)) - 2.0 * np.eye(n)
    Bc = rng.standard_normal((batch, n, m))
    return Ac, Bc


def _run(ctx, entry, Ac, Bc, dt, il):
    batch, n, m = Ac.shape[0], Ac.shape[1], Bc.shape[2]
    dev = torch.device("cuda", 0)
    d = ffi.make_dims(n, m, 1, 1, 1, batch, 0, il)
    ac = torch.from_numpy(L.pack(Ac, il)).to(dev)
    bc = torch.from_numpy(L.pack(Bc, il)).to(dev)
    A = torch.full((L.array_elems(batch, il, n * n),), float("nan"), dtype=torch.float64, device=dev)
    B = torch.full((L.array_elems(batch, il, n * m),), float("nan"), dtype=torch.float64, device=dev)
    info = torch.full((batch,), -1, dtype=torch.int32, device=dev)
    sp = torch.cuda.current_stream().cuda_stream
    entry(d, ac.data_ptr(), bc.data_ptr(), dt, A.data_ptr(), B.data_ptr(), info.data_ptr(), sp)
    torch.cuda.synchronize()
    return (L.unpack(A.cpu().numpy(), batch, n, n, il), L.unpack(B.cpu().numpy(), batch, n, m, il), info.cpu().numpy())


@pytest.mark.parametrize("il", [1, 0, 64])
def test_noalloc_c2d_matches_existing_entry_point_twice(ctx, il):
    rng = np.random.default_rng(77 + il)
    # general-n path (static scratch): a larger call, then a smaller one with another shape, then a lane-kernel shape
    # and two systems beyond the one-wavefront solver kernels (the blocked solver, its workspace behind the matrices)
    for batch, n, m in ((700, 12, 4), (90, 9, 2), (33, 40, 3), (300, 4, 1), (150, 80, 3), (9, 120, 2)):
        Ac, Bc = _plants(rng, batch, n, m)
        Ac[batch // 2] = 0.0
        Ac[batch // 2, 0, 0] = 4.0  # dt = 0.25: I - Ac dt has an exactly zero first row and column (fused or not): flagged, NaN outputs
        want = _run(ctx, ctx.c2d_dev, Ac, Bc, 0.25, il)
        got = _run(ctx, ctx.c2d_noalloc_dev, Ac, Bc, 0.25, il)
        assert np.array_equal(got[2], want[2]) and want[2][batch // 2] == 1 and want[2].sum() == 1
        assert np.array_equal(got[0], want[0], equal_nan=True), f"A differs (batch {batch}, n {n}, W {il})"
        assert np.array_equal(got[1], want[1], equal_nan=True), f"B differs (batch {batch}, n {n}, W {il})"
        ok = want[2] == 0
        assert np.all(np.isfinite(got[0][ok])) and np.all(np.isfinite(got[1][ok]))
        # and the result is the discretisation: (I - Ac dt) A = I to rounding
        resid = np.einsum("bij,bjk->bik", np.eye(n) - Ac[ok] * 0.25, got[0][ok]) - np.eye(n)
        assert np.abs(resid).max() < 1e-10


def test_noalloc_c2d_limits(ctx):
    dev = torch.device("cuda", 0)
    z = torch.zeros(16, dtype=torch.float64, device=dev)
    # beyond the static scratch: rejected before any launch (the pointers are never touched)
    n = 12
    batch = (1 << 21) // (n * n) + 1
    d = ffi.make_dims(n, 1, 1, 1, 1, batch, 0, 1)
    with pytest.raises(ffi.MpcxError) as ei:
        ctx.c2d_noalloc_dev(d, z.data_ptr(), z.data_ptr(), 0.01, z.data_ptr(), z.data_ptr(), None, 0)
    assert ei.value.code == ffi.ERR_UNSUPPORTED and "static scratch" in ei.value.message
    # beyond the solver: rejected
    d = ffi.make_dims(1300, 1, 1, 1, 1, 1, 0, 1)
    with pytest.raises(ffi.MpcxError) as ei:
        ctx.c2d_noalloc_dev(d, z.data_ptr(), z.data_ptr(), 0.01, z.data_ptr(), z.data_ptr(), None, 0)
    assert ei.value.code == ffi.ERR_UNSUPPORTED
    # matrices that fit, but no room behind them for one instance of the blocked solver's workspace: rejected
    d = ffi.make_dims(80, 1, 1, 1, 1, (1 << 21) // 6400, 0, 1)
    with pytest.raises(ffi.MpcxError) as ei:
        ctx.c2d_noalloc_dev(d, z.data_ptr(), z.data_ptr(), 0.01, z.data_ptr(), z.data_ptr(), None, 0)
    assert ei.value.code == ffi.ERR_UNSUPPORTED and "solver workspace" in ei.value.message
    # exactly at the limit is accepted
    batch = (1 << 21) // (n * n)
    rng = np.random.default_rng(3)
    Ac, Bc = _plants(rng, batch, n, 2)
    want = _run(ctx, ctx.c2d_dev, Ac, Bc, 0.02, 1)
    got = _run(ctx, ctx.c2d_noalloc_dev, Ac, Bc, 0.02, 1)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
