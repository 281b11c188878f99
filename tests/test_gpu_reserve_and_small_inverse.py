"""(1) mpcx_reserve: after it, the `_dev` calls of those dimensions allocate nothing (free device memory does not
move across them).  (2) Systems of 1..4 unknowns, where nalgebra's try_inverse takes closed-form inverses that fail on
`determinant == 0.0` (model.rs:56, controller.rs:93-95; oracle: orc_try_inverse).  The device evaluates the same closed
forms with multiply and add kept separate, so it must flag exactly the instances the oracle flags -- exactly singular
matrices and matrices whose determinant cancels to zero in rounding (two columns equal to the last bit) alike -- and
agree to 1e-10 x cond on the rest.  (The reference holds no fixture for inversion failures: the oracle's restatement of
nalgebra's published closed forms is what pins this.)"""
import numpy as np
import pytest
import torch

from mpc_rs_amd import ffi, layout as L
from oracle import oracle_ffi as orc

pytestmark = pytest.mark.gpu


def test_reserve_then_dev_calls_do_not_allocate():
    # a context of its own: the session's may already hold a scratch area larger than anything asked for here
    ctx = ffi.Context(0)
    try:
        _reserve_then_dev_calls(ctx)
    finally:
        ctx.close()


def _reserve_then_dev_calls(ctx):
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n, m, r, f, v = 8, 2, 2, 80, 80  # 160 unknowns: blocked solver workspace; n = 8: general discretisation
    il = 64  # interleaved batch on the workgroup kernels: instance-major copy in context scratch
    rng = np.random.default_rng(4)
    sp = torch.cuda.current_stream().cuda_stream
    W3 = torch.eye(v * m, dtype=torch.float64, device=dev).contiguous()
    Q = torch.eye(r, dtype=torch.float64, device=dev).contiguous()

    def run_all(batch, reserve_first):
        d = ffi.make_dims(n, m, r, f, v, batch, 0, il)

        def dv(elems, fill=None):
            t = torch.empty(L.array_elems(batch, il, elems), dtype=torch.float64, device=dev)
            t.copy_(torch.from_numpy(rng.standard_normal(t.numel()) * 0.1 if fill is None else np.full(t.numel(), fill)))
            return t
        A, B, Cm, x0, ref = dv(n * n), dv(n * m), dv(r * n), dv(n), dv(f * r)
        Phi, Gam, H, fv = dv(f * r * n, 0.0), dv(f * r * v * m, 0.0), dv((v * m) ** 2, 0.0), dv(v * m, 0.0)
        gain, useq = dv(v * m * f * r, 0.0), dv(v * m, 0.0)
        Ad, Bd = dv(n * n, 0.0), dv(n * m, 0.0)
        info = torch.zeros(batch, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        before = torch.cuda.mem_get_info(0)[0]
        if reserve_first:
            ctx.reserve(d)
            torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info(0)[0]
        ctx.qp_build_dev(d, A.data_ptr(), B.data_ptr(), Cm.data_ptr(), W3.data_ptr(), Q.data_ptr(), x0.data_ptr(), ref.data_ptr(),
                         Phi.data_ptr(), Gam.data_ptr(), H.data_ptr(), fv.data_ptr(), sp)
        ctx.qp_gain_dev(d, H.data_ptr(), Gam.data_ptr(), Q.data_ptr(), gain.data_ptr(), info.data_ptr(), sp)
        ctx.qp_solve_dev(d, H.data_ptr(), fv.data_ptr(), useq.data_ptr(), info.data_ptr(), sp)
        ctx.c2d_dev(d, A.data_ptr(), B.data_ptr(), 0.01, Ad.data_ptr(), Bd.data_ptr(), info.data_ptr(), sp)
        torch.cuda.synchronize()
        free1 = torch.cuda.mem_get_info(0)[0]
        assert bool(torch.isfinite(gain).all()) and bool(torch.isfinite(useq).all()) and bool(torch.isfinite(Ad).all())
        return before, free0, free1

    # one instance first: every kernel involved gets loaded (the runtime allocates for code objects on first launch)
    run_all(1, reserve_first=False)
    before, free0, free1 = run_all(192, reserve_first=True)
    assert free0 < before, "mpcx_reserve was expected to grow the context's scratch for the larger batch"
    assert free1 == free0, "a _dev call allocated device memory after mpcx_reserve"


def test_reserve_covers_builds_with_fewer_outputs_when_the_relay_cap_binds():
    """With a small MPCX_OPT_RELAY_CAP_BYTES the instance-major copy holds a whole number of instances under the cap; a
    build that asks for Gamma alone has a smaller per-instance size, so its chunk holds more instances and chunk x size can
    exceed what the all-outputs chunk takes.  After mpcx_reserve neither form may allocate."""
    ctx = ffi.Context(0)
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        n, m, r, f, v, il, batch = 8, 2, 2, 40, 40, 64, 256
        per_full = n * n + n * m + r * n + n + f * r + f * r * n + (f * r) * (v * m) + (v * m) ** 2 + v * m
        per_gamma = n * n + n * m + r * n + (f * r) * (v * m)
        cap = 8 * (127 * per_full + per_full // 2)  # all outputs: one group of 64 instances; Gamma alone: four groups
        assert (cap // (8 * per_gamma)) // 64 * 64 * per_gamma > 64 * per_full
        ctx.set_option(ffi.OPT_RELAY_CAP_BYTES, cap)
        rng = np.random.default_rng(12)
        sp = torch.cuda.current_stream().cuda_stream

        def dv(elems, b):
            return torch.from_numpy(rng.standard_normal(L.array_elems(b, il, elems)) * 0.1).to(dev)
        W3 = torch.eye(v * m, dtype=torch.float64, device=dev).contiguous()
        Q = torch.eye(r, dtype=torch.float64, device=dev).contiguous()

        def run(b, reserve):
            d = ffi.make_dims(n, m, r, f, v, b, 0, il)
            A, B, Cm, x0, ref = dv(n * n, b), dv(n * m, b), dv(r * n, b), dv(n, b), dv(f * r, b)
            Phi, H, fv = dv(f * r * n, b), dv((v * m) ** 2, b), dv(v * m, b)
            Gam = torch.zeros(L.array_elems(b, il, f * r * v * m), dtype=torch.float64, device=dev)  # (padding lanes stay zero)
            Gam2 = torch.zeros_like(Gam)
            torch.cuda.synchronize()
            if reserve:
                ctx.reserve(d)
                torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info(0)[0]
            ctx.qp_build_dev(d, A.data_ptr(), B.data_ptr(), Cm.data_ptr(), None, None, None, None, None, Gam2.data_ptr(), None, None, sp)
            torch.cuda.synchronize()
            free1 = torch.cuda.mem_get_info(0)[0]
            ctx.qp_build_dev(d, A.data_ptr(), B.data_ptr(), Cm.data_ptr(), W3.data_ptr(), Q.data_ptr(), x0.data_ptr(), ref.data_ptr(),
                             Phi.data_ptr(), Gam.data_ptr(), H.data_ptr(), fv.data_ptr(), sp)
            torch.cuda.synchronize()
            free2 = torch.cuda.mem_get_info(0)[0]
            assert torch.equal(Gam, Gam2)  # chunking differs between the two calls, the results must not
            return free0, free1, free2
        run(1, reserve=False)  # code objects loaded
        free0, free1, free2 = run(batch, reserve=True)
        assert free1 == free0, "the Gamma-only build allocated after mpcx_reserve"
        assert free2 == free0, "the all-outputs build allocated after mpcx_reserve"
    finally:
        ctx.close()


@pytest.mark.parametrize("k", [1, 2, 3, 4])
def test_small_systems_closed_form_as_try_inverse(ctx, k):
    rng = np.random.default_rng(40 + k)
    nb = 256
    H = rng.standard_normal((nb, k, k)) + 2.0 * np.eye(k)                 # well conditioned
    Hs = H.copy()
    Hs[:, :, 0] = 0.0                                                      # exactly singular: a column of exact zeros
    near = rng.standard_normal((nb, k, k))
    if k > 1:
        near[:, :, -1] = near[:, :, 0] * (1.0 + 1e-16 * rng.integers(-4, 5, (nb, 1)))  # rank k-1 to rounding
    else:
        near[:] = 1e-300
    allH = np.concatenate([H, Hs, near])
    rhs = rng.standard_normal((allH.shape[0], k))
    u, info = ctx.qp_solve(allH, rhs, m=1)
    ok_ref = np.array([orc.try_inverse(allH[b])[0] for b in range(allH.shape[0])])
    # well conditioned: both succeed and agree
    assert np.all(info[:nb] == 0) and np.all(ok_ref[:nb])
    for b in range(nb):
        want = -orc.try_inverse(allH[b])[1] @ rhs[b]
        tol = 1e-10 * max(np.linalg.cond(allH[b]), 1.0)
        assert np.linalg.norm(u[b] - want) <= tol * max(np.linalg.norm(want), 1e-300)
    # exactly singular from exact zeros: both fail, the device leaves NaN
    assert np.all(info[nb:2 * nb] == 1) and not np.any(ok_ref[nb:2 * nb])
    assert np.all(np.isnan(u[nb:2 * nb]))
    # singular to rounding: the determinant test decides, instance by instance, as in the closed forms of the oracle
    dev_fail, ref_fail = info[2 * nb:] == 1, ~ok_ref[2 * nb:]
    print(f"k={k}: near-singular instances flagged: device {int(dev_fail.sum())}, closed form {int(ref_fail.sum())}, "
          f"disagree {int((dev_fail != ref_fail).sum())} of {nb}")
    if k > 1:
        assert ref_fail.sum() > 0, "the family was meant to contain determinants that cancel exactly"
    assert np.array_equal(dev_fail, ref_fail)
    assert np.all(np.isnan(u[2 * nb:][dev_fail])) and np.all(np.isfinite(u[2 * nb:][~dev_fail]))


@pytest.mark.parametrize("n", [1, 2, 3, 4])
def test_c2d_small_plants_flag_what_the_closed_form_flags(ctx, n):
    rng = np.random.default_rng(70 + n)
    nb, dt = 200, 0.25
    # entries on a 2^-18 grid and dt a power of two: I - Ac dt is formed without rounding
    Ac = np.round((rng.standard_normal((nb, n, n)) - 2.0 * np.eye(n)) * 2.0 ** 18) / 2.0 ** 18
    if n > 1:  # I - Ac dt with two equal columns in a third of the instances: the determinant cancels to exactly zero
        e_first, e_last = np.eye(n)[:, 0], np.eye(n)[:, -1]
        for b in range(0, nb, 3):
            Ac[b, :, -1] = Ac[b, :, 0] + (e_last - e_first) / dt
    Bc = rng.standard_normal((nb, n, 1))
    A, B, info = ctx.c2d(Ac, Bc, dt, interleave=0)
    st = np.array([orc.c2d_backward_euler(Ac[b], Bc[b], dt)[0] for b in range(nb)])
    assert np.array_equal(info != 0, st != orc.ORC_OK)
    if n > 1:
        assert (info != 0).sum() > 0  # (for n = 4 rounding in the triple products keeps many determinants off zero)
    for b in np.nonzero(info == 0)[0]:
        _, Aw, Bw = orc.c2d_backward_euler(Ac[b], Bc[b], dt)
        tol = 1e-10 * max(np.linalg.cond(np.eye(n) - Ac[b] * dt), 1.0)
        assert np.linalg.norm(A[b] - Aw) <= tol * np.linalg.norm(Aw) and np.linalg.norm(B[b] - Bw) <= tol * max(np.linalg.norm(Bw), 1e-300)


@pytest.mark.parametrize("n", [1, 2, 3, 4])
def test_c2d_with_three_inputs_rounds_the_product_before_the_subtraction(ctx, n):
    """n <= 4 with m = 3 has no one-lane instantiation and goes through the general path.  `identity - mat_ac.scale(dt)`
    (model.rs:56) rounds Ac dt first: with Ac[0][0] = 3 and dt = 1/3 the product rounds to exactly 1, the first row and
    column of I - Ac dt vanish and try_inverse fails on a zero determinant; a fused multiply-subtract would leave 2^-54
    there and invert.  Flags must be the oracle's, instance by instance."""
    rng = np.random.default_rng(170 + n)
    nb, dt = 96, 1.0 / 3.0
    assert 3.0 * dt == 1.0 and np.float64(3.0) * np.float64(dt) == 1.0
    Ac = rng.standard_normal((nb, n, n)) - 2.0 * np.eye(n)
    for b in range(0, nb, 4):
        Ac[b, 0, :] = 0.0
        Ac[b, :, 0] = 0.0
        Ac[b, 0, 0] = 3.0
    Bc = rng.standard_normal((nb, n, 3))
    for il in (0, 1):
        A, B, info = ctx.c2d(Ac, Bc, dt, interleave=il)
        st = np.array([orc.c2d_backward_euler(Ac[b], Bc[b], dt)[0] for b in range(nb)])
        assert np.array_equal(info != 0, st != orc.ORC_OK)
        assert (info != 0).sum() == nb // 4
        for b in np.nonzero(info == 0)[0]:
            _, Aw, Bw = orc.c2d_backward_euler(Ac[b], Bc[b], dt)
            tol = 1e-10 * max(np.linalg.cond(np.eye(n) - Ac[b] * dt), 1.0)
            assert np.linalg.norm(A[b] - Aw) <= tol * np.linalg.norm(Aw) and np.linalg.norm(B[b] - Bw) <= tol * max(np.linalg.norm(Bw), 1e-300)
