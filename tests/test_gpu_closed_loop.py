"""GPU parity of the batched receding-horizon loop (reference: src/main.rs:51-71 over controller.rs:101-135) against
the oracle: orc_closed_loop for m = r = 1, the only case the reference handles, and orc_closed_loop_mimo (the same
products with the window stacked time-major) for m, r > 1.
Tolerance: relative Frobenius error <= 1e-10 on states, inputs and outputs, given the same Phi and gain."""
import json
import os

import numpy as np
import pytest

from helpers import assert_close, oracle_qp, random_stable_plants
from mpc_rs_amd import synth
from mpc_rs_amd.control import model, trajectory
from oracle import oracle_ffi as orc

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def np_closed_loop(A, B, C, x0, Phi, gain, ref, T, f):
    """ref r x (T+f).  The formulas of controller.rs:119-132 and :106-112 with the window stacked time-major."""
    n, m, r = A.shape[0], B.shape[1], C.shape[0]
    X, U, Y = np.zeros((n, T + 1)), np.zeros((m, T)), np.zeros((r, T))
    X[:, 0] = x0
    for t in range(T):
        s = ref[:, t:t + f].T.reshape(-1) - Phi @ X[:, t]
        U[:, t] = (gain @ s)[:m]
        Y[:, t] = C @ X[:, t]
        X[:, t + 1] = A @ X[:, t] + B @ U[:, t]
    return X, U, Y


def _siso_case(name, batch, f, v):
    d = synth.make_batch(name, batch, f=f, v=v)
    w4 = synth.dense_w4(d["Q"], f)
    Phi, gain = [], []
    for b in range(batch):
        st, O, M, H, G = orc.form_lifted_matrices(d["A"][b], d["B"][b], d["C"][b], f, v, d["W3"], w4)
        assert st == orc.ORC_OK
        Phi.append(O)
        gain.append(G)
    return d, np.stack(Phi), np.stack(gain)


@pytest.mark.parametrize("name,batch,il,T,f,v,shared", [
    ("two_mass", 200, 0, 60, 20, 20, True),
    ("two_mass", 130, 64, 37, 20, 20, False),
    ("two_mass", 70, 128, 5, 6, 3, False),
    ("msd", 96, 0, 41, 20, 20, False),
    ("msd", 9, 1, 33, 20, 20, True),       # W = 1: one wavefront per instance
    ("two_mass", 5, 1, 19, 6, 3, False),
])
def test_siso_against_oracle(ctx, name, batch, il, T, f, v, shared):
    d, Phi, gain = _siso_case(name, batch, f, v)
    rng = np.random.default_rng(T)
    pulse = ((np.arange(T + f) // 7) % 2 == 0).astype(np.float64).reshape(1, -1)
    ref = pulse if shared else rng.uniform(0, 1, (batch, 1, T + f))
    got = ctx.closed_loop(d["A"], d["B"], d["C"], d["x0"], Phi, gain, ref, T, f, v, interleave=il)
    X, U, Y = np.zeros((batch, 4 if name == "two_mass" else 2, T + 1)), np.zeros((batch, 1, T)), np.zeros((batch, 1, T))
    for b in range(batch):
        tr = (ref if shared else ref[b]).reshape(-1)
        S, Out, In = orc.closed_loop(d["A"][b], d["B"][b], d["C"][b], f, v, Phi[b], gain[b], tr, T, d["x0"][b])
        X[b], U[b, 0], Y[b] = S, In, Out[:, 1:]          # main.rs:63 records C x_t after the step: outputs[t+1] = y_t
    assert_close(got["X"], X, "states")
    assert_close(got["U"], U, "inputs")
    assert_close(got["Y"], Y, "outputs")


@pytest.mark.parametrize("n,m,r,f,v,T,il,shared", [
    (4, 2, 2, 8, 5, 21, 0, False),      # lane kernel, MIMO
    (3, 2, 1, 7, 7, 16, 64, True),
    (4, 1, 2, 9, 4, 8, 0, False),
    (5, 2, 3, 6, 4, 19, 0, False),      # one wavefront per instance
    (12, 4, 12, 10, 6, 17, 1, True),
    (7, 5, 2, 40, 9, 11, 1, False),     # m > 4, f r > 64
    (2, 1, 1, 20, 20, 3, 1, False),     # T < chunk
    (9, 3, 3, 5, 5, 0, 1, False),       # T = 0
])
def test_mimo_against_oracle(ctx, n, m, r, f, v, T, il, shared):
    rng = np.random.default_rng(n * 100 + m * 10 + r)
    batch = 23
    A, B, C = random_stable_plants(rng, batch, n, m, r)
    x0 = rng.uniform(-1, 1, (batch, n))
    Phi = rng.standard_normal((batch, f * r, n)) * 0.3
    gain = rng.standard_normal((batch, v * m, f * r)) * (0.2 / np.sqrt(f * r))
    ref = rng.uniform(0, 1, (r, T + f)) if shared else rng.uniform(0, 1, (batch, r, T + f))
    got = ctx.closed_loop(A, B, C, x0, Phi, gain, ref, T, f, v, interleave=il)
    # the C oracle's loop for any m, r (orc_closed_loop_mimo: the products and order of compute_control_input and
    # propagate_dynamics); the numpy restatement above must agree with it
    want = [orc.closed_loop_mimo(A[b], B[b], C[b], f, v, Phi[b], gain[b], ref if shared else ref[b], T, x0[b]) for b in range(batch)]
    for b in (0, batch - 1):
        for w, q in zip(want[b], np_closed_loop(A[b], B[b], C[b], x0[b], Phi[b], gain[b], ref if shared else ref[b], T, f)):
            assert np.allclose(w, q, rtol=1e-11, atol=1e-13)
    assert got["X"].shape == (batch, n, T + 1)
    assert_close(got["X"], np.stack([w[0] for w in want]), "states")
    if T:
        assert_close(got["U"], np.stack([w[1] for w in want]), "inputs")
        assert_close(got["Y"], np.stack([w[2] for w in want]), "outputs")


def test_device_chain_reproduces_the_reference_chart(ctx):
    """build -> gain -> closed loop, all on the device, for the scenario of src/main.rs:14-71; the controlled output must
    pass through the curve of the reference's control.png."""
    mdl = model.dc_motor.Model(model.dc_motor.Parameters(), 0.05)
    A, B, C = mdl.get_mat_a()[None], mdl.get_mat_b()[None], mdl.get_mat_c()[None]
    f = v = 20
    T = 300 - f
    _, _, w3, w4 = orc.controller_weights(1, 1, f, v, 1.1e-9, 1e-4, 10.0)
    qp = ctx.qp_build(A, B, C, f, v, W3=w3, W4=w4[:1, :1], want=("Phi", "Gamma", "H"))
    gain, info = ctx.qp_gain(qp["H"], qp["Gamma"], w4[:1, :1], 1, 1, f, v)
    assert info[0] == 0
    traj = trajectory.generate_pulse_trajectory(300).reshape(1, -1)
    out = ctx.closed_loop(A, B, C, np.zeros((1, 2)), qp["Phi"], gain, traj, T, f, v)
    outputs = np.concatenate([[0.0], out["Y"][0, 0]])
    t = json.load(open(os.path.join(GOLDEN, "control_png_trace.json")))
    kk, hp = np.arange(len(outputs)), t["x_halfpixel"]
    for x, lo, hi in zip(t["x"], t["y_lo"], t["y_hi"]):
        seg = [np.interp(min(max(x + dx, 0), len(outputs) - 1), kk, outputs) for dx in (-hp, 0.0, hp)]
        assert max(seg) >= lo - 2e-3 and min(seg) <= hi + 2e-3, x
    st, O, M, H, G = orc.form_lifted_matrices(A[0], B[0], C[0], f, v, w3, w4)
    S, Out, In = orc.closed_loop(A[0], B[0], C[0], f, v, O, G, traj.reshape(-1), T, np.zeros(2))
    assert_close(out["U"][0], In[None], "inputs", tol=1e-8)      # through H^-1, cond(H) ~ 268
    assert_close(out["X"][0][None], S[None], "states", tol=1e-8)


def test_closed_loop_errors(ctx):
    from mpc_rs_amd import ffi
    A, B, C = np.eye(2)[None], np.ones((1, 2, 1)), np.ones((1, 1, 2))
    with pytest.raises(ffi.MpcxError) as e:
        d = ffi.make_dims(2, 1, 1, 4, 4, 1, -1, 1)
        ctx.closed_loop_dev(d, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1)
    assert e.value.code == ffi.ERR_DIMS
    with pytest.raises(ffi.MpcxError) as e:
        ctx.closed_loop_dev(ffi.make_dims(2, 1, 1, 4, 4, 1, 3, 1), None, 1, 1, 1, 1, 1, 1, 1, 1, 1)
    assert e.value.code == ffi.ERR_NULL
