"""The reference's own scenario (src/main.rs:14-71) through the host mirror of its API, on the GPU, against the oracle
and against the curves read off the reference's charts."""
import json
import os

import numpy as np
import pytest

from helpers import assert_close
from mpc_rs_amd.control import controller, model, simulator, trajectory
from oracle import oracle_ffi as orc

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _trace_ok(name, y, tol):
    t = json.load(open(os.path.join(GOLDEN, name)))
    kk, hp = np.arange(len(y)), t["x_halfpixel"]
    for x, lo, hi in zip(t["x"], t["y_lo"], t["y_hi"]):
        seg = [np.interp(min(max(x + d, 0), len(y) - 1), kk, y) for d in (-hp, 0.0, hp)]
        assert max(seg) >= lo - tol and min(seg) <= hi + tol, (name, x)


def test_main_rs_scenario(ctx):
    sampling_dt = 0.05
    mdl = model.dc_motor.Model(model.dc_motor.Parameters(), sampling_dt)
    step_response, step, _ = simulator.step(mdl, 10.0)
    assert step_response.shape == (1, 200) and step.shape == (1, 200)
    _trace_ok("step_response_png_trace.json", step_response[0], 5e-5)
    assert simulator.compute_system_response(mdl, step, np.zeros(2)) == list(step_response[0])

    f = v = 20
    time_steps = 300
    traj = trajectory.generate_pulse_trajectory(time_steps)
    mpc = controller.mpc.Controller.new(mdl, f, v, 0.0000000011, 0.0001, 10.0, traj)
    states = [np.zeros(2)]
    outputs = [mdl.get_mat_c() @ states[0]]
    inputs = []
    for t in range(time_steps - f):
        u = mpc.compute_control_input(t, states[t])
        nxt, y = mpc.propagate_dynamics(u, states[t])
        states.append(nxt)
        outputs.append(y)
        inputs.append(u[0])
    outputs = np.array(outputs)[:, 0]
    _trace_ok("control_png_trace.json", outputs, 2e-3)

    A, B, C = mdl.get_mat_a(), mdl.get_mat_b(), mdl.get_mat_c()
    _, _, w3, w4 = orc.controller_weights(1, 1, f, v, 1.1e-9, 1e-4, 10.0)
    st, O, M, H, G = orc.form_lifted_matrices(A, B, C, f, v, w3, w4)
    S, Out, In = orc.closed_loop(A, B, C, f, v, O, G, traj, time_steps - f, np.zeros(2))
    assert_close(np.array(inputs)[None], In[None], "closed-loop inputs", tol=1e-8)   # through H^-1, cond(H) ~ 268
    assert_close(outputs[None], Out, "closed-loop outputs", tol=1e-8)
    o2, m2, g2 = controller.mpc.Controller.form_lifted_matrices(A, B, C, f, v, w3, w4)
    assert_close(o2[None], O[None], "mat_o")
    assert_close(m2[None], M[None], "mat_m")
    assert_close(g2[None], G[None], "gain", tol=1e-8)
    # first_move_only (an addition): the controller keeps row 0 of the gain and produces the same loop
    lean = controller.mpc.Controller.new(mdl, f, v, 0.0000000011, 0.0001, 10.0, traj, first_move_only=True)
    assert lean._gain_matrix.shape == (1, f)
    assert_close(lean._gain_matrix[None], G[None, :1], "first gain row", tol=1e-8)
    X2, U2, Y2 = lean.closed_loop(time_steps - f, np.zeros(2))
    assert_close(U2, In[None], "closed-loop inputs from the first gain row", tol=1e-8)
    assert_close(Y2, Out, "closed-loop outputs from the first gain row", tol=1e-8)
    assert abs(lean.compute_control_input(0, np.zeros(2))[0] - In[0]) <= 1e-8 * abs(In[0])


def test_error_behaviour(ctx):
    with pytest.raises(ValueError):
        trajectory.generate_pulse_trajectory(301)
    A, C = 0.5 * np.eye(2), np.eye(2)
    B = np.ones((2, 2))
    with pytest.raises(RuntimeError, match="Matrix inversion failed"):
        controller.mpc.Controller.form_lifted_matrices(A, B, C, 2, 2, np.zeros((4, 4)), np.eye(4))
    with pytest.raises(ValueError):
        controller.mpc.Controller.form_lifted_matrices(A, B, C, 2, 2, np.zeros((3, 3)), np.eye(4))


def test_controller_additions_mimo_and_device_loop(ctx):
    """Additions to the reference API: q I weight blocks, the m-input move and the whole loop in one device call."""
    rng = np.random.default_rng(5)

    class Plant(model.DiscreteStateSpaceModel):
        def __init__(self):
            self.a = np.array([[0.9, 0.1, 0.0], [0.0, 0.8, 0.1], [0.05, 0.0, 0.7]])
            self.b = rng.standard_normal((3, 2))
            self.c = np.array([[1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])

        def get_mat_a(self): return self.a
        def get_mat_b(self): return self.b
        def get_mat_c(self): return self.c
        def get_sampling_dt(self): return 0.1

    plant = Plant()
    f, v, steps = 8, 5, 30
    traj = rng.uniform(0, 1, (steps + f, 2))
    # the reference's all-elements blocks with two identical input columns give an exactly singular Hessian: same error
    class Twin(Plant):
        def __init__(self):
            super().__init__()
            self.a, self.b, self.c = 0.5 * np.eye(2), np.ones((2, 2)), np.eye(2)
    with pytest.raises(RuntimeError, match="Matrix inversion failed"):
        controller.mpc.Controller.new(Twin(), 2, 2, 0.0, 0.0, 1.0, np.zeros((10, 2)))
    mpc = controller.mpc.Controller.new(plant, f, v, 1e-3, 1e-2, 10.0, traj, all_elements=False)
    X, U, Yout = mpc.closed_loop(steps, np.array([0.3, -0.2, 0.1]))
    x = np.array([0.3, -0.2, 0.1])
    for t in range(steps):
        u = mpc.compute_control_input_mimo(t, x)
        assert_close(U[:, t][None], u[None], "input", tol=1e-9)
        x = plant.a @ x + plant.b @ u
        assert_close(X[:, t + 1][None], x[None], "state", tol=1e-9)
    assert Yout.shape == (2, steps + 1)
    # the gain itself against numpy, through the oracle's weights
    _, _, w3, w4 = orc.controller_weights(2, 2, f, v, 1e-3, 1e-2, 10.0)
    w2 = np.zeros((v * 2, v * 2)); w1 = np.zeros_like(w2)
    for i in range(v):
        w1[2 * i:2 * i + 2, 2 * i:2 * i + 2] = np.eye(2)
        if i: w1[2 * i:2 * i + 2, 2 * i - 2:2 * i] = -np.eye(2)
        w2[2 * i:2 * i + 2, 2 * i:2 * i + 2] = (1e-3 if i == 0 else 1e-2) * np.eye(2)
    w3 = w1.T @ w2 @ w1
    w4 = 10.0 * np.eye(f * 2)
    st, O, M, H, G = orc.form_lifted_matrices(plant.a, plant.b, plant.c, f, v, w3, w4)
    assert st == orc.ORC_OK
    assert_close(mpc._gain_matrix[None], G[None], "gain", tol=1e-8 * np.linalg.cond(H))


def test_dense_w4_gain_matches_block(ctx):
    """form_lifted_matrices with a W4 that is not block-repeated goes down the dense path on the device."""
    rng = np.random.default_rng(6)
    A = np.array([[0.9, 0.2], [-0.1, 0.8]]); B = np.array([[0.0], [1.0]]); C = np.array([[1.0, 0.0]])
    f, v = 6, 4
    w3 = np.eye(v) * 0.1
    w4 = np.diag(rng.uniform(1, 2, f))          # time-varying output weight: not a repeated block
    o, m_, g = controller.mpc.Controller.form_lifted_matrices(A, B, C, f, v, w3, w4)
    st, O, M, H, G = orc.form_lifted_matrices(A, B, C, f, v, w3, w4)
    assert_close(g[None], G[None], "gain (dense W4)", tol=1e-9 * np.linalg.cond(H))
