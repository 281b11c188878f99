"""GPU parity of the batched solve / gain (reference: controller.rs:93-96, :130-134) against the oracle and numpy.
Tolerance: relative Frobenius error <= 1e-10 * cond(H) (a solve amplifies rounding by the condition number)."""
import numpy as np
import pytest

from helpers import oracle_qp, random_stable_plants, rel_fro
from mpc_rs_amd import synth
from oracle import oracle_ffi as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,batch,il", [("two_mass", 300, 0), ("two_mass", 70, 1), ("msd", 129, 64)])
def test_solve_and_gain_on_workloads(ctx, name, batch, il):
    d = synth.make_batch(name, batch)
    want = oracle_qp(orc, d)
    u, info = ctx.qp_solve(want["H"], want["fvec"], d["m"], interleave=il)
    assert np.all(info == 0)
    ref_u = np.stack([-np.linalg.solve(want["H"][b], want["fvec"][b, :, 0]) for b in range(batch)])
    cond = max(np.linalg.cond(h) for h in want["H"][:8])
    assert rel_fro(u, ref_u) <= 1e-10 * cond
    gain, info = ctx.qp_gain(want["H"], want["Gamma"], d["Q"], d["m"], d["r"], d["f"], d["v"], interleave=il)
    assert np.all(info == 0)
    w4 = synth.dense_w4(d["Q"], d["f"])
    for b in range(0, batch, max(batch // 6, 1)):
        st, O, M, H, G = orc.form_lifted_matrices(d["A"][b], d["B"][b], d["C"][b], d["f"], d["v"], d["W3"], w4)
        assert st == orc.ORC_OK and rel_fro(gain[b], G) <= 1e-10 * cond
        # first move of the reference: gain (ref - O x) == -H^-1 fvec
        s = d["ref"][b] - O @ d["x0"][b]
        assert abs((gain[b] @ s)[0] - u[b, 0]) <= 1e-9 * cond * max(abs(u[b, 0]), 1e-30)


@pytest.mark.parametrize("vm,nrhs_m", [(1, 1), (2, 1), (7, 1), (20, 2), (64, 4), (65, 5), (100, 1)])
def test_solve_shapes_and_pivoting(ctx, vm, nrhs_m):
    rng = np.random.default_rng(vm)
    batch = 17
    H = rng.standard_normal((batch, vm, vm))
    H[:, 0, 0] = 0.0  # forces a row exchange in the first step
    fv = rng.standard_normal((batch, vm))
    u, info = ctx.qp_solve(H, fv, 1, interleave=1)
    if vm == 1:
        assert np.all(info == 1) and np.all(np.isnan(u))
        return
    assert np.all(info == 0)
    ref = np.stack([-np.linalg.solve(H[b], fv[b]) for b in range(batch)])
    cond = max(np.linalg.cond(h) for h in H)
    assert rel_fro(u, ref) <= 1e-11 * cond


def test_singular_instances_are_flagged_not_fatal(ctx):
    rng = np.random.default_rng(3)
    batch, vm = 12, 6
    H = rng.standard_normal((batch, vm, vm)) + 4 * np.eye(vm)
    H[3] = 0.0                      # exactly singular: Err("Matrix inversion failed") in the reference
    H[7, :, 2] = 0.0                # a zero column
    H[9, 1, 1] = np.nan
    fv = rng.standard_normal((batch, vm))
    u, info = ctx.qp_solve(H, fv, 1, interleave=0)
    assert list(np.nonzero(info)[0]) == [3, 7, 9]
    assert np.all(np.isnan(u[[3, 7, 9]])) and np.all(np.isfinite(np.delete(u, [3, 7, 9], axis=0)))
    good = [b for b in range(batch) if b not in (3, 7, 9)]
    ref = np.stack([-np.linalg.solve(H[b], fv[b]) for b in good])
    assert rel_fro(u[good], ref) <= 1e-10
    # the all-elements weight blocks of Controller::new with repeated input columns: the reference's own failure case
    A, C = 0.5 * np.eye(2), np.eye(2)
    B = np.ones((2, 2))
    qp = ctx.qp_build(A[None], B[None], C[None], 2, 2, W3=np.zeros((4, 4)), W4=np.eye(2), want=("Gamma", "H"))
    _, info = ctx.qp_gain(qp["H"], qp["Gamma"], np.eye(2), 2, 2, 2, 2)
    st, *_ = orc.form_lifted_matrices(A, B, C, 2, 2, np.zeros((4, 4)), np.eye(4))
    assert info[0] == 1 and st == orc.ORC_ERR_INVERSION


@pytest.mark.parametrize("il", [0, 1, 64])
@pytest.mark.parametrize("batch", [1, 7, 8, 61])
def test_reference_horizon_rows_in_registers(ctx, il, batch):
    """20 unknowns with 20 right-hand sides (gain, single output) or one (solve): the register-resident elimination.
    Row exchanges at several steps, singular / NaN instances flagged, ragged last workgroup, every layout."""
    rng = np.random.default_rng(100 * batch + il)
    vm = fr = 20
    H = rng.standard_normal((batch, vm, vm))
    H[:, 0, 0] = 0.0            # exchange in the first step
    H[:, 5, :5] = 0.0           # row 5 cannot be a pivot before step 5
    H[:, 5, 5] = 1e-9           # and is a poor one there
    Gamma = rng.standard_normal((batch, fr, vm))
    q = np.array([[2.5]])
    fv = rng.standard_normal((batch, vm))
    bad = [b for b in (2, 5) if b < batch]
    if 2 < batch:
        H[2, :, 7] = 0.0        # zero column: singular
    if 5 < batch:
        H[5, 3, 4] = np.nan
    good = [b for b in range(batch) if b not in bad]
    gain, info = ctx.qp_gain(H, Gamma, q, 1, 1, 20, 20, interleave=il)
    assert list(np.nonzero(info)[0]) == bad
    assert np.all(np.isnan(gain[bad])) and np.all(np.isfinite(gain[good]))
    u, info_s = ctx.qp_solve(H, fv, 1, interleave=il)
    assert list(np.nonzero(info_s)[0]) == bad
    assert np.all(np.isnan(u[bad])) and np.all(np.isfinite(u[good]))
    for b in good:
        cond = np.linalg.cond(H[b])
        want = np.linalg.solve(H[b], Gamma[b].T * q[0, 0])
        assert rel_fro(gain[b], want) <= 1e-11 * cond, f"gain of instance {b}"
        assert rel_fro(u[b], -np.linalg.solve(H[b], fv[b])) <= 1e-11 * cond, f"solve of instance {b}"


@pytest.mark.parametrize("vm,m,fr,r,il", [(65, 1, 70, 1, 1), (100, 2, 150, 3, 1), (130, 1, 140, 1, 0), (200, 4, 600, 12, 1), (48, 1, 500, 1, 1),
                                          (300, 2, 320, 2, 1), (530, 1, 530, 1, 1), (904, 2, 1356, 3, 1),
                                          (80, 2, 800, 20, 1), (272, 4, 2176, 32, 0)])  # weight blocks too wide for the LDS copy
def test_large_systems_blocked_solver(ctx, vm, m, fr, r, il):
    """Beyond 64 unknowns (or right-hand sides that do not fit LDS): the blocked solver on a device workspace."""
    rng = np.random.default_rng(vm + fr)
    batch = 5
    v, f = vm // m, fr // r
    H = rng.standard_normal((batch, vm, vm))
    H[:, 0, 0] = 0.0                                  # row exchange in the very first step
    H[:, 17, :] *= 1e-3                               # a badly scaled row: pivoting matters
    fv = rng.standard_normal((batch, vm))
    u, info = ctx.qp_solve(H, fv, m, interleave=il)
    assert np.all(info == 0)
    ref = np.stack([-np.linalg.solve(H[b], fv[b]) for b in range(batch)])
    cond = max(np.linalg.cond(h) for h in H)
    assert rel_fro(u, ref) <= 1e-11 * cond
    Gamma = rng.standard_normal((batch, fr, vm))
    Q = rng.standard_normal((r, r))
    gain, info = ctx.qp_gain(H, Gamma, Q, m, r, f, v, interleave=il)
    assert np.all(info == 0)
    w4 = np.kron(np.eye(f), Q)
    want = np.stack([np.linalg.solve(H[b], Gamma[b].T @ w4) for b in range(batch)])
    assert rel_fro(gain, want) <= 1e-11 * cond


@pytest.mark.parametrize("vm", [48, 64, 96, 400])
def test_large_singular_is_flagged(ctx, vm):
    rng = np.random.default_rng(8)
    batch = 4
    H = rng.standard_normal((batch, vm, vm)) + 10 * np.eye(vm)
    H[1, :, 40] = 0.0
    if vm <= 64:
        # two equal rows far apart: an exactly zero pivot late in the elimination, because both rows see the same
        # operations in the same order -- the column-by-column kernels (up to 64 unknowns) share that with the reference's
        # LU; the panel-wise solver treats a pivot row and an outside row differently and promises no exact zero here
        H[2, vm - 9, :] = H[2, 3, :]
    else:
        H[2, vm - 30, :] = 0.0                        # a zero row stays zero through every update: the last pivot is exactly zero
    fv = rng.standard_normal((batch, vm))
    u, info = ctx.qp_solve(H, fv, 1)
    assert list(info) == [0, 1, 1, 0]
    assert np.all(np.isnan(u[1])) and np.all(np.isnan(u[2]))
    ref = np.stack([-np.linalg.solve(H[b], fv[b]) for b in (0, 3)])
    assert rel_fro(u[[0, 3]], ref) <= 1e-10


def test_largest_system_and_the_first_refused(ctx):
    """1232 unknowns is the largest system the blocked solver takes (a block of 16 columns of all rows in LDS); one more
    is refused with MPCX_ERR_UNSUPPORTED instead of being attempted."""
    from mpc_rs_amd import ffi
    rng = np.random.default_rng(1232)
    vm, batch = 1232, 2
    H = rng.standard_normal((batch, vm, vm)) + 30.0 * np.eye(vm)
    H[:, 5, :] *= 1e-2
    fv = rng.standard_normal((batch, vm))
    u, info = ctx.qp_solve(H, fv, 1)
    assert np.all(info == 0)
    ref = np.stack([-np.linalg.solve(H[b], fv[b]) for b in range(batch)])
    assert rel_fro(u, ref) <= 1e-11 * max(np.linalg.cond(h) for h in H)
    with pytest.raises(ffi.MpcxError) as err:
        ctx.qp_solve(rng.standard_normal((1, vm + 1, vm + 1)), rng.standard_normal((1, vm + 1)), 1)
    assert err.value.code == -4  # MPCX_ERR_UNSUPPORTED
