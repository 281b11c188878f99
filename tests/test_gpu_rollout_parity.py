"""GPU parity of the batched rollout (simulator.rs system_simulate / controller.rs propagate_dynamics) against the
C oracle, plus the consistency property that ties the rollout to the QP build:
    Phi x0 + Gamma u_seq == [y_1 ... y_f]   when u is held at u_(v-1) past the control horizon.
Tolerance: relative Frobenius error <= 1e-10 (tests/helpers.py RTOL).
"""
import numpy as np
import pytest

from helpers import assert_close, random_stable_plants
from mpc_rs_amd import synth
from oracle import oracle_ffi as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,batch", [("msd", 300), ("two_mass", 129)])
@pytest.mark.parametrize("interleave", [0, 64, 1])
def test_rollout_step_response(ctx, name, batch, interleave):
    d = synth.make_batch(name, batch)
    X, Y, _ = orc.rollout_batch(d["A"], d["B"], d["C"], np.zeros((batch, d["n"])), d["U"], threads=4)
    got = ctx.rollout(d["A"], d["B"], d["C"], np.zeros((batch, d["n"])), d["U"], interleave=interleave)
    assert_close(got["X"], X, f"{name} W={interleave} X")
    assert_close(got["Y"], Y, f"{name} W={interleave} Y")


@pytest.mark.parametrize("n,m,r,T", [(2, 1, 1, 50), (4, 2, 2, 33), (3, 2, 1, 7), (12, 4, 12, 60), (8, 2, 2, 90),
                                     (17, 3, 5, 20), (64, 64, 64, 3), (5, 1, 1, 1), (6, 2, 2, 0)])
@pytest.mark.parametrize("interleave", [0, 1])
def test_rollout_shapes_per_instance_inputs(ctx, n, m, r, T, interleave):
    rng = np.random.default_rng(n * 31 + m * 7 + r + T)
    batch = 70
    A, B, C = random_stable_plants(rng, batch, n, m, r)
    x0 = rng.standard_normal((batch, n))
    U = rng.standard_normal((batch, m, T))
    X, Y, _ = orc.rollout_batch(A, B, C, x0, U, threads=4)
    got = ctx.rollout(A, B, C, x0, U, interleave=interleave)
    assert got["X"].shape == (batch, n, T + 1) and got["Y"].shape == (batch, r, T)
    if T == 0:
        assert np.all(got["X"] == 0.0)  # simulator.rs:13-17: x0 is only copied inside the loop
        return
    assert_close(got["X"], X, f"n{n} m{m} r{r} T{T} W={interleave} X")
    assert_close(got["Y"], Y, f"n{n} m{m} r{r} T{T} W={interleave} Y")


@pytest.mark.parametrize("n,m,r", [(4, 1, 10), (8, 2, 16), (12, 4, 14), (2, 1, 12), (6, 3, 8), (10, 2, 12)])
@pytest.mark.parametrize("shared", [False, True])
@pytest.mark.parametrize("interleave", [1, 0])
def test_rollout_more_outputs_than_states(ctx, n, m, r, shared, interleave):
    """r > n: the output rows, not the state rows, size the flush of Y (even n, r take the paired-store path at W = 1)."""
    rng = np.random.default_rng(1000 + n * 31 + m * 7 + r)
    batch, T = 37, 29  # T not a multiple of the 8-step chunk
    A, B, C = random_stable_plants(rng, batch, n, m, r)
    x0 = rng.standard_normal((batch, n))
    U = rng.standard_normal((m, T)) if shared else rng.standard_normal((batch, m, T))
    X, Y, _ = orc.rollout_batch(A, B, C, x0, U, threads=4)
    got = ctx.rollout(A, B, C, x0, U, interleave=interleave)
    assert_close(got["X"], X, f"n{n} m{m} r{r} shared={shared} W={interleave} X")
    assert_close(got["Y"], Y, f"n{n} m{m} r{r} shared={shared} W={interleave} Y")


@pytest.mark.parametrize("n,m,r,T", [(12, 4, 12, 200), (12, 4, 12, 29), (10, 3, 8, 9), (8, 2, 6, 8), (6, 4, 8, 7), (2, 1, 2, 1),
                                     (4, 2, 12, 17), (12, 1, 2, 401), (12, 4, 12, 0), (6, 2, 4, 192), (8, 4, 8, 193)])
def test_rollout_four_lanes_per_instance(ctx, n, m, r, T):
    """Large instance-major batches with a shared input sequence, even n, r <= 12 and m <= 4 take the kernel with four lanes
    per instance (state exchanged inside the quad, inputs staged in LDS 192 steps at a time): ragged batch, T around the
    8-step chunk and beyond one staging block, every plant different (simulator.rs:5-35)."""
    rng = np.random.default_rng(9000 + n * 131 + m * 17 + r * 3 + T)
    batch = 12288 + 5
    A, B, C = random_stable_plants(rng, batch, n, m, r, rho=0.95)
    x0 = rng.standard_normal((batch, n))
    U = rng.standard_normal((m, T))
    X, Y, _ = orc.rollout_batch(A, B, C, x0, U, threads=8)
    got = ctx.rollout(A, B, C, x0, U, interleave=1)
    assert_close(got["X"], X, f"n{n} m{m} r{r} T{T} X")
    assert_close(got["Y"], Y, f"n{n} m{m} r{r} T{T} Y")
    if T in (29, 9):  # one output alone: the same bits as in the full call
        for k in ("X", "Y"):
            part = ctx.rollout(A, B, C, x0, U, interleave=1, want=(k,))
            assert np.array_equal(part[k], got[k]), f"{k} alone differs from the full call"


def test_propagate_dynamics_is_rollout_with_T1(ctx):
    rng = np.random.default_rng(5)
    A, B, C = random_stable_plants(rng, 40, 4, 1, 1)
    x = rng.standard_normal((40, 4))
    u = rng.standard_normal((40, 1, 1))
    got = ctx.rollout(A, B, C, x, u, interleave=0)
    for b in range(40):
        xn, y = orc.propagate_dynamics(A[b], B[b], C[b], u[b, :, 0], x[b])
        assert_close(got["X"][b:b + 1, :, 1], xn[None], "x_next")
        assert_close(got["Y"][b:b + 1, :, 0], y[None], "y")


@pytest.mark.parametrize("name,batch,il,f,v", [("two_mass", 500, 0, 20, 20), ("two_mass", 200, 0, 20, 6),
                                               ("quadrotor", 4, 1, 12, 5), ("long", 3, 1, 60, 60)])
def test_prediction_consistency(ctx, name, batch, il, f, v):
    d = synth.make_batch(name, batch, f=f, v=v)
    n, m, r = d["n"], d["m"], d["r"]
    rng = np.random.default_rng(11)
    useq = rng.standard_normal((batch, v, m))
    U = np.concatenate([useq, np.repeat(useq[:, -1:, :], f - v + 1, axis=1)], axis=1).transpose(0, 2, 1)  # (batch, m, f+1)
    qp = ctx.qp_build(d["A"], d["B"], d["C"], f, v, interleave=il, want=("Phi", "Gamma"))
    ro = ctx.rollout(d["A"], d["B"], d["C"], d["x0"], np.ascontiguousarray(U), interleave=il)
    pred = np.einsum("bij,bj->bi", qp["Phi"], d["x0"]) + np.einsum("bij,bj->bi", qp["Gamma"], useq.reshape(batch, v * m))
    ys = ro["Y"][:, :, 1:f + 1].transpose(0, 2, 1).reshape(batch, f * r)  # y_1..y_f stacked
    assert_close(pred, ys, f"{name} prediction consistency", tol=1e-9)
