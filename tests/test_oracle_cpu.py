"""CPU tests of the oracle itself (no GPU): the C restatement against the independent numpy restatement, against the
committed golden vectors, against values derived by hand from the reference's formulas for its own scenario
(src/main.rs:15-24, 44-51), and against the curves read off the two charts the reference ships
(tests/golden/*_png_trace.json, extracted by tools/extract_png_trace.py).
"""
import glob
import json
import os

import numpy as np
import pytest

from helpers import rel_fro
from mpc_rs_amd import synth
from oracle import np_oracle as npo
from oracle import oracle_ffi as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _dc_motor(dt=0.05):
    Ac, Bc, Cc = orc.dc_motor_continuous()
    st, A, B = orc.c2d_backward_euler(Ac, Bc, dt)
    assert st == orc.ORC_OK
    return A, B, Cc


def test_dc_motor_discretisation_known_values():
    A, B, C = _dc_motor()
    # backward Euler of model.rs:47-58 with the defaults of model.rs:27-31, dt = 0.05
    np.testing.assert_allclose(A, [[0.6666464652586285, 0.030302112057210385], [-0.0006060422411442077, 0.9090633617163115]], rtol=1e-13)
    np.testing.assert_allclose(B[:, 0], [0.003030211205721039, 0.09090633617163116], rtol=1e-13)
    np.testing.assert_array_equal(C, [[1.0, 0.0]])
    A2, B2 = npo.c2d_backward_euler(*npo.dc_motor_continuous()[:2], 0.05)
    assert rel_fro(A, A2) < 1e-14 and rel_fro(B, B2) < 1e-14


def test_two_mass_model_matches_numpy():
    Ac, Bc, Cc = orc.two_mass_continuous()
    Ac2, Bc2, Cc2 = npo.two_mass_continuous()
    np.testing.assert_array_equal(Ac, Ac2)
    np.testing.assert_array_equal(Bc, Bc2)
    np.testing.assert_array_equal(Cc, Cc2)
    st, A, B = orc.c2d_backward_euler(Ac, Bc, 0.05)
    A2, B2 = npo.c2d_backward_euler(Ac2, Bc2, 0.05)
    assert st == orc.ORC_OK and rel_fro(A, A2) < 1e-13 and rel_fro(B, B2) < 1e-13


def test_weights_structure():
    w1, w2, w3, w4 = orc.controller_weights(1, 1, 20, 20, 1.1e-9, 1e-4, 10.0)
    n1, n2, n3, n4 = npo.controller_weights(1, 1, 20, 20, 1.1e-9, 1e-4, 10.0)
    for a, b in ((w1, n1), (w2, n2), (w3, n3), (w4, n4)):
        assert np.array_equal(a, b)
    # m = 1: W3 is tridiagonal, diag [q0+qo, 2qo, ..., 2qo, qo], off-diagonal -qo (controller.rs:149-177)
    assert w3[0, 0] == pytest.approx(1.1e-9 + 1e-4) and w3[5, 5] == pytest.approx(2e-4) and w3[19, 19] == pytest.approx(1e-4)
    assert w3[3, 4] == pytest.approx(-1e-4) and w3[0, 2] == 0.0
    # the blocks are from_element blocks: every entry carries the weight (controller.rs:164-165, 184)
    _, w2m, _, w4m = orc.controller_weights(2, 3, 4, 2, 0.5, 0.25, 7.0)
    assert np.array_equal(w2m[:2, :2], np.full((2, 2), 0.5)) and np.array_equal(w2m[2:, 2:], np.full((2, 2), 0.25))
    assert np.array_equal(w4m[3:6, 3:6], np.full((3, 3), 7.0)) and w4m[0, 3] == 0.0


def test_scenario_known_answers():
    """Values derived from the reference's formulas for its own scenario (not produced by the Rust code)."""
    A, B, C = _dc_motor()
    f = v = 20
    _, _, w3, w4 = orc.controller_weights(1, 1, f, v, 1.1e-9, 1e-4, 10.0)
    st, O, M, H, G = orc.form_lifted_matrices(A, B, C, f, v, w3, w4)
    assert st == orc.ORC_OK
    np.testing.assert_allclose(O[0], [0.6666464652586285, 0.03030211205721038], rtol=1e-12)
    np.testing.assert_allclose(O[19], [2.5493554266479e-4, 1.851218229365717e-2], rtol=1e-11)
    assert M[0, 0] == pytest.approx(0.003030211205721039, rel=1e-13) and M[19, 19] == M[0, 0]
    assert M[19, 0] == pytest.approx(0.0018512182293657176, rel=1e-12)
    assert np.all(np.triu(M, 1) == 0.0)
    assert H[0, 0] == pytest.approx(0.0038152174019254687, rel=1e-12)
    assert H[19, 19] == pytest.approx(1.9182179951277354e-4, rel=1e-12)
    assert H[0, 19] == pytest.approx(5.609582222859058e-5, rel=1e-12)
    assert 260 < np.linalg.cond(H) < 275
    Y, X = orc.system_simulate(A, B, C, np.ones((1, orc.lib().orc_step_n_samples(10.0, 0.05))), np.zeros(2))
    assert Y.shape == (1, 200) and X.shape == (2, 201)
    assert Y[0, 0] == 0.0 and Y[0, 1] == pytest.approx(0.003030211205721039, rel=1e-13)
    assert Y[0, 2] == pytest.approx(0.007804944780385261, rel=1e-13) and Y[0, 199] == pytest.approx(0.09990009919268994, rel=1e-12)
    st, traj = orc.pulse_trajectory(300)
    S, Out, In = orc.closed_loop(A, B, C, f, v, O, G, traj, 280, np.zeros(2))
    assert In[0] == pytest.approx(159.64050095605938, rel=1e-10)
    # held-input structure, f = 6, v = 3: the last column cumulates the Markov parameters (controller.rs:67-80)
    _, _, w3b, w4b = orc.controller_weights(1, 1, 6, 3, 1.1e-9, 1e-4, 10.0)
    st, O6, M6, H6, _ = orc.form_lifted_matrices(A, B, C, 6, 3, w3b, w4b)
    np.testing.assert_allclose(M6[3:, 2], [0.00780494478038526, 0.01349210340491539, 0.0195597249006968], rtol=1e-12)
    g = M6[:, 0]
    np.testing.assert_allclose(M6[2:, 2], np.cumsum(g)[:4], rtol=1e-13)


def test_c_oracle_matches_numpy_restatement_and_fma_variant():
    rng = np.random.default_rng(3)
    for (n, m, r, f, v) in [(2, 1, 1, 20, 20), (4, 1, 1, 20, 20), (4, 2, 3, 9, 4), (12, 4, 12, 8, 8), (8, 2, 2, 25, 25), (3, 1, 2, 7, 1)]:
        A = rng.standard_normal((n, n))
        A *= 0.9 / max(abs(np.linalg.eigvals(A)))
        B, C = rng.standard_normal((n, m)), rng.standard_normal((r, n))
        W3, W4 = rng.standard_normal((v * m, v * m)), rng.standard_normal((f * r, f * r))
        x0, ref = rng.standard_normal(n), rng.standard_normal(f * r)
        st, O, M, H, G = orc.form_lifted_matrices(A, B, C, f, v, W3, W4)
        O2, M2 = npo.lifted_matrices(A, B, C, f, v)
        assert st == orc.ORC_OK
        assert rel_fro(O, O2) < 1e-13 and rel_fro(M, M2) < 1e-13 and rel_fro(H, npo.hessian(M2, W3, W4)) < 1e-13
        assert rel_fro(orc.gradient(O, M, W4, x0, ref), npo.gradient(O2, M2, W4, x0, ref)) < 1e-12
        assert rel_fro(G, npo.gain_matrix(M2, W3, W4)) < 1e-8 * np.linalg.cond(H)
        st3, O3, M3, H3, _ = orc.form_lifted_matrices(A, B, C, f, v, W3, W4, variant="fma")
        # separate multiply-add vs fused: bounds the FMA ambiguity of the reference's arithmetic
        assert rel_fro(O3, O) < 1e-14 and rel_fro(M3, M) < 1e-14 and rel_fro(H3, H) < 1e-13
        U = rng.standard_normal((m, 30))
        Y, X = orc.system_simulate(A, B, C, U, x0)
        Y2, X2 = npo.system_simulate(A, B, C, U, x0)
        assert rel_fro(Y, Y2) < 1e-13 and rel_fro(X, X2) < 1e-13
        xn, y = orc.propagate_dynamics(A, B, C, U[:, 0], x0)
        assert rel_fro(xn, X[:, 1]) < 1e-15 and rel_fro(y, Y[:, 0]) < 1e-15


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "qp_*.npz"))))
def test_c_oracle_matches_golden_vectors(path):
    g = np.load(path, allow_pickle=False)
    f, v = int(g["f"]), int(g["v"])
    out, _ = orc.qp_build_batch(g["A"], g["B"], g["C"], g["W3"], synth.dense_w4(g["Q"], f), g["x0"], g["ref"], f, v, threads=2)
    for k in ("Phi", "Gamma", "H", "fvec"):
        assert rel_fro(out[k], g[k]) < 1e-12, k
    X, Y, _ = orc.rollout_batch(g["A"], g["B"], g["C"], g["x0"], g["U"], threads=2)
    assert rel_fro(X, g["X"]) < 1e-13 and rel_fro(Y, g["Y"]) < 1e-13


def test_try_inverse_and_failure_paths():
    rng = np.random.default_rng(0)
    for n in range(1, 9):
        a = rng.standard_normal((n, n)) + n * np.eye(n)
        ok, inv = orc.try_inverse(a)
        assert ok and np.allclose(inv @ a, np.eye(n), atol=1e-12)
    for n in (1, 2, 3, 4, 6):
        ok, _ = orc.try_inverse(np.zeros((n, n)))
        assert not ok
    # all-elements weight blocks with m > 1 make H exactly singular when Gamma's columns repeat:
    # Err("Matrix inversion failed") of controller.rs:93-95
    A, C = 0.5 * np.eye(2), np.eye(2)
    B = np.array([[1.0, 1.0], [1.0, 1.0]])
    st, O, M, H, G = orc.form_lifted_matrices(A, B, C, 2, 2, np.zeros((4, 4)), np.eye(4))
    assert st == orc.ORC_ERR_INVERSION and G is None
    st, _, _ = orc.c2d_backward_euler(np.eye(2) / 0.05, np.ones((2, 1)), 0.05)  # I - Ac dt = 0
    assert st == orc.ORC_ERR_INVERSION


def test_pulse_trajectory():
    st, t = orc.pulse_trajectory(300)
    assert st == orc.ORC_OK and np.array_equal(t, npo.pulse_trajectory(300))
    assert t[:100].sum() == 100 and t[100:200].sum() == 0 and t[200:].sum() == 100
    st, _ = orc.pulse_trajectory(301)  # 301 - 200 != 100: copy_from would panic (trajectory.rs:7-8)
    assert st == orc.ORC_ERR_SHAPE
    with pytest.raises(ValueError):
        npo.pulse_trajectory(301)


def _check_trace(name, y, tol):
    t = json.load(open(os.path.join(GOLDEN, name)))
    kk = np.arange(len(y))
    hp = t["x_halfpixel"]
    assert len(t["x"]) > 90
    for x, lo, hi in zip(t["x"], t["y_lo"], t["y_hi"]):
        seg = [np.interp(min(max(x + d, 0), len(y) - 1), kk, y) for d in (-hp, 0.0, hp)]
        assert max(seg) >= lo - tol and min(seg) <= hi + tol, (name, x, seg, lo, hi)


def test_oracle_reproduces_the_reference_charts():
    """The reference's own recorded outputs: the curves of step_response.png and control.png, to within one pixel."""
    A, B, C = _dc_motor()
    Y, _ = orc.system_simulate(A, B, C, np.ones((1, 200)), np.zeros(2))
    _check_trace("step_response_png_trace.json", Y[0], 5e-5)
    _, _, w3, w4 = orc.controller_weights(1, 1, 20, 20, 1.1e-9, 1e-4, 10.0)
    st, O, M, H, G = orc.form_lifted_matrices(A, B, C, 20, 20, w3, w4)
    _, traj = orc.pulse_trajectory(300)
    S, Out, In = orc.closed_loop(A, B, C, 20, 20, O, G, traj, 280, np.zeros(2))
    _check_trace("control_png_trace.json", Out[0], 2e-3)
    S2, Out2, In2 = npo.closed_loop(A, B, C, O, G, traj, 280, np.zeros(2))
    assert rel_fro(Out, Out2) < 1e-12 and rel_fro(In, In2) < 1e-12


@pytest.mark.parametrize("name,batch,f,v", [("two_mass", 40, 20, 20), ("two_mass", 9, 9, 4), ("msd", 17, 6, 1),
                                            ("quadrotor", 2, 7, 7), ("quadrotor", 2, 8, 3), ("long", 1, 30, 30)])
def test_fast_cpu_build_matches_the_oracle(name, batch, f, v):
    """bench.py's "optimised CPU" baseline (recurrences, block weights) against the reference-structured oracle."""
    from mpc_rs_amd import synth
    d = synth.make_batch(name, batch, f=f, v=v)
    w4 = synth.dense_w4(d["Q"], f)
    want, _ = orc.qp_build_batch(d["A"], d["B"], d["C"], d["W3"], w4, d["x0"], d["ref"], f, v)
    got, _ = orc.qp_build_batch(d["A"], d["B"], d["C"], d["W3"], w4, d["x0"], d["ref"], f, v, threads=2, fast=True)
    for k in ("Phi", "Gamma", "H", "fvec"):
        err = np.linalg.norm((got[k] - want[k]).ravel()) / max(np.linalg.norm(want[k].ravel()), 1e-300)
        assert err <= 1e-10, (k, err)


def test_mimo_closed_loop_restatement_reduces_to_the_reference_loop():
    """orc_closed_loop_mimo (window stacked time-major, any m, r) is bit-identical to the reference's SISO loop
    (main.rs:51-71 over controller.rs:101-135) for m = r = 1 and agrees with an independent numpy loop for m, r > 1."""
    rng = np.random.default_rng(8)
    f, v, T = 6, 4, 19
    n, m, r = 3, 1, 1
    A, B, C = rng.standard_normal((n, n)) * 0.3, rng.standard_normal((n, m)), rng.standard_normal((r, n))
    Phi, gain = rng.standard_normal((f * r, n)) * 0.3, rng.standard_normal((v * m, f * r)) * 0.05
    traj, x0 = rng.uniform(0, 1, T + f), rng.uniform(-1, 1, n)
    S, Out, In = orc.closed_loop(A, B, C, f, v, Phi, gain, traj, T, x0)
    X, U, Y = orc.closed_loop_mimo(A, B, C, f, v, Phi, gain, traj.reshape(1, -1), T, x0)
    assert np.array_equal(S, X) and np.array_equal(In, U[0]) and np.array_equal(Out[:, 1:], Y)
    n, m, r = 5, 2, 3
    A, B, C = rng.standard_normal((n, n)) * 0.3, rng.standard_normal((n, m)), rng.standard_normal((r, n))
    Phi, gain = rng.standard_normal((f * r, n)) * 0.3, rng.standard_normal((v * m, f * r)) * 0.05
    ref, x0 = rng.uniform(0, 1, (r, T + f)), rng.uniform(-1, 1, n)
    X, U, Y = orc.closed_loop_mimo(A, B, C, f, v, Phi, gain, ref, T, x0)
    Xn = np.zeros((n, T + 1))
    Xn[:, 0] = x0
    for t in range(T):
        s = ref[:, t:t + f].T.reshape(-1) - Phi @ Xn[:, t]
        u = (gain @ s)[:m]
        assert np.allclose(u, U[:, t], rtol=1e-12, atol=1e-14) and np.allclose(C @ Xn[:, t], Y[:, t], rtol=1e-12, atol=1e-14)
        Xn[:, t + 1] = A @ Xn[:, t] + B @ u
    assert np.allclose(Xn, X, rtol=1e-11, atol=1e-13)
