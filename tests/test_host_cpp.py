"""The C++ host API (mpc-rs_amd/host/control_sys_rs.hpp: the reference's public API, name for name, above the C ABI)
driven by tests/host_cpp/host_check.cpp and compared with the CPU oracle.  Without a GPU the same binary must fail
loudly (DeviceError, exit code 3): the host side has no arithmetic of its own to fall back to."""
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
DIR = os.path.join(HERE, "host_cpp")


def _exe(name):
    path = os.path.join(DIR, name)
    if not os.path.exists(path):
        subprocess.run(["make", "-C", DIR, name], check=True, capture_output=True)
    return path


def _parse(stdout):
    vals = {}
    for line in stdout.splitlines():
        toks = line.split()
        if toks[0] == "mimo_err":
            vals["mimo_err"] = " ".join(toks[1:])
            continue
        rows, cols = int(toks[1]), int(toks[2])
        vals[toks[0]] = np.array([float(t) for t in toks[3:]]).reshape(cols, rows).T  # printed column-major
    return vals


def _rel(a, b):
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))


def test_cpp_host_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu tests")
    p = subprocess.run([_exe("host_check")], capture_output=True, text=True, timeout=120)
    assert p.returncode == 3 and "device_error -5" in p.stdout, (p.returncode, p.stdout, p.stderr)
    p = subprocess.run([_exe("demo_main")], capture_output=True, text=True, timeout=120)
    assert p.returncode == 3 and "computes on the GPU only" in p.stderr, (p.returncode, p.stdout, p.stderr)


@pytest.fixture(scope="module")
def host():
    p = subprocess.run([_exe("host_check")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    return _parse(p.stdout)


@pytest.mark.gpu
def test_models_step_and_trajectory(host):
    from oracle import oracle_ffi as orc
    Ac, Bc, Cc = orc.dc_motor_continuous()
    _, Ad, Bd = orc.c2d_backward_euler(Ac, Bc, 0.05)
    assert _rel(host["motor_A"], Ad) <= 1e-12 and _rel(host["motor_B"], Bd) <= 1e-12 and np.array_equal(host["motor_C"], Cc)
    Ac4, Bc4, Cc4 = orc.two_mass_continuous()
    _, A4, B4 = orc.c2d_backward_euler(Ac4, Bc4, 0.05)
    assert _rel(host["two_mass_A"], A4) <= 1e-12 and _rel(host["two_mass_B"], B4) <= 1e-12
    assert np.array_equal(host["two_mass_C"], Cc4) and host["motor_dt"][0, 0] == 0.05
    Y, X = orc.system_simulate(Ad, Bd, Cc, np.ones((1, 200)), np.zeros(2))          # simulator.rs:50-63
    assert host["step_response"].shape == (1, 200) and host["step_states"].shape == (2, 201)
    assert _rel(host["step_response"], Y) <= 1e-10 and _rel(host["step_states"], X) <= 1e-10
    assert np.array_equal(host["step_input"], np.ones((1, 200)))
    assert list(host["step_empty"][0]) == [0, 0, 1]                                 # T = 0: Y r x 0, X n x 1
    U = np.sin(0.3 * np.arange(50)).reshape(1, 50)
    Y2, _ = orc.system_simulate(A4, B4, Cc4, U, 0.1 * np.arange(1, 5))
    assert _rel(host["response_two_mass"], Y2[:1]) <= 1e-10
    st, traj = orc.pulse_trajectory(300)
    assert st == 0 and np.array_equal(host["trajectory"], traj)
    assert orc.pulse_trajectory(301)[0] != 0 and host["trajectory_301"][0, 0] == 1  # the reference panics there


@pytest.mark.gpu
def test_controller_scenario_of_the_reference_demo(host):
    from oracle import oracle_ffi as orc
    f = v = 20
    Ac, Bc, Cc = orc.dc_motor_continuous()
    _, Ad, Bd = orc.c2d_backward_euler(Ac, Bc, 0.05)
    _, _, w3, w4 = orc.controller_weights(1, 1, f, v, 1.1e-9, 1e-4, 10.0)
    st, O, M, H, G = orc.form_lifted_matrices(Ad, Bd, Cc, f, v, w3, w4)
    assert st == 0
    assert _rel(host["motor_mat_o"], O) <= 1e-10
    assert _rel(host["motor_gain"], G) <= 1e-10 * np.linalg.cond(H)
    _, traj = orc.pulse_trajectory(300)
    S, Out, In = orc.closed_loop(Ad, Bd, Cc, f, v, O, G, traj.reshape(-1), 280, np.zeros(2))
    # per-step calls (compute_control_input + propagate_dynamics, main.rs:51-71) and the one-call loop
    for pre in ("loop", "loop1"):
        assert _rel(host[pre + "_inputs"].reshape(-1), In) <= 1e-8
        assert _rel(host[pre + "_states"], S) <= 1e-8
    assert _rel(host["loop_outputs"].reshape(-1), Out[0, 1:]) <= 1e-8                # outputs[t+1] = y_t (main.rs:63)
    assert _rel(host["loop1_outputs"], Out) <= 1e-8                                  # with outputs[0] = C x0 (main.rs:41)
    assert host["slice_panic"][0, 0] == 1
    # the controller that keeps only the first gain row (first_move_only): row 0 of the gain, same loop, same first move
    assert host["lean_gain"].shape == (1, 20) and _rel(host["lean_gain"], G[:1]) <= 1e-10 * np.linalg.cond(H)
    assert _rel(host["lean_inputs"].reshape(-1), In) <= 1e-8 and _rel(host["lean_states"], S) <= 1e-8
    assert _rel(host["lean_outputs"], Out) <= 1e-8 and abs(host["lean_u0"][0, 0] - In[0]) <= 1e-8 * abs(In[0])


@pytest.mark.gpu
def test_form_lifted_matrices_held_input_dense_weight_and_errors(host):
    from oracle import oracle_ffi as orc
    f, v = 6, 3
    Ac4, Bc4, Cc4 = orc.two_mass_continuous()
    _, A4, B4 = orc.c2d_backward_euler(Ac4, Bc4, 0.05)
    w3 = np.array([[0.5 + 0.1 * i if i == j else 0.01 * (i + 2 * j) for j in range(v)] for i in range(v)])
    st, O, M, H, G = orc.form_lifted_matrices(A4, B4, Cc4, f, v, w3, 3.0 * np.eye(f))
    assert st == 0
    for k, want in (("flm_O", O), ("flm_M", M), ("flm_H", H)):
        assert _rel(host[k], want) <= 1e-10, k
    assert _rel(host["flm_gain"], G) <= 1e-10 * np.linalg.cond(H)
    w4 = host["flm_dense_w4"]
    assert not np.allclose(w4, np.diag(np.diag(w4)))
    st, _, _, Hd, Gd = orc.form_lifted_matrices(A4, B4, Cc4, f, v, w3, w4)
    assert st == 0 and _rel(host["flm_dense_gain"], Gd) <= 1e-10 * np.linalg.cond(Hd)
    assert host["shape_panic"][0, 0] == 1
    assert host["mimo_err"] == "Matrix inversion failed"                              # controller.rs:95
    # q I blocks: gain and move against a numpy restatement of controller.rs:92-96, :125-130
    a, b, c = np.array([[0.9, 0.1], [0.0, 0.8]]), np.eye(2), np.eye(2)
    m = r = 2
    fm = vm = 4
    w1 = np.eye(vm * m) - np.eye(vm * m, k=-m)
    w3m = w1.T @ (1e-3 * np.eye(vm * m) @ w1)
    w4m = np.eye(fm * r)
    st, Om, Mm, Hm, Gm = orc.form_lifted_matrices(a, b, c, fm, vm, w3m, w4m)
    assert st == 0 and _rel(host["mimo_mat_o"], Om) <= 1e-10 and _rel(host["mimo_gain"], Gm) <= 1e-10 * np.linalg.cond(Hm)
    move = (Gm @ (np.ones(fm * r) - Om @ np.array([0.3, -0.2])))[:m]
    assert _rel(host["mimo_move"].reshape(-1), move) <= 1e-9


@pytest.mark.gpu
def test_batched_additions(host):
    from oracle import oracle_ffi as orc
    B, n, f, v, T = 8, 4, 20, 20, 40
    A = np.zeros((B, 4, 4)); Bm = np.zeros((B, 4, 1)); Cm = np.zeros((B, 1, 4))
    for i in range(B):
        s = 1.0 + 0.01 * i
        Ac, Bc, Cc = orc.two_mass_continuous(2.0 * s, 2.0 / s, 100.0 * s, 200.0, 1.0 * s, 5.0)
        _, A[i], Bm[i] = orc.c2d_backward_euler(Ac, Bc, 0.05)
        Cm[i] = Cc
    got_a = host["batch_A"].T.reshape(B, 4, 4).transpose(0, 2, 1)
    assert _rel(got_a, A) <= 1e-12
    x0 = np.array([[0.05 * (i + k) for k in range(n)] for i in range(B)])
    ref = np.tile(np.r_[np.ones(10), np.zeros(10)], (B, 1))
    _, _, w3, w4 = orc.controller_weights(1, 1, f, v, 1.1e-9, 1e-4, 10.0)
    want, _ = orc.qp_build_batch(A, Bm, Cm, w3, w4, x0, ref, f, v)
    shapes = {"phi": ("Phi", f, n), "gamma": ("Gamma", f, v), "h": ("H", v, v), "fvec": ("fvec", v, 1)}
    for k, (name, rows, cols) in shapes.items():
        got = host["batch_" + k].T.reshape(B, cols, rows).transpose(0, 2, 1)
        assert _rel(got, want[name]) <= 1e-10, k
    gain = host["batch_gain"].T.reshape(B, f, v).transpose(0, 2, 1)
    for i in range(B):
        st, _, _, H, G = orc.form_lifted_matrices(A[i], Bm[i], Cm[i], f, v, w3, w4)
        assert st == 0 and _rel(gain[i], G) <= 1e-10 * np.linalg.cond(H)
    assert host["batch_info"][0, 0] == 0
    X, Y, _ = orc.rollout_batch(A, Bm, Cm, x0, np.ones((1, T)))
    assert _rel(host["batch_X"].T.reshape(B, T + 1, n).transpose(0, 2, 1), X) <= 1e-10
    assert _rel(host["batch_Y"].T.reshape(B, T, 1).transpose(0, 2, 1), Y) <= 1e-10


@pytest.mark.gpu
def test_demo_main_prints_the_reference_charts_series():
    import json
    p = subprocess.run([_exe("demo_main")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.stdout[-500:], p.stderr[-500:])
    step = [float(l.split(",")[2]) for l in p.stdout.splitlines() if l.startswith("step,")]
    loop = [l.split(",") for l in p.stdout.splitlines() if l.startswith("loop,")]
    assert len(step) == 200 and len(loop) == 280
    # the curves of the reference's own charts (step_response.png, control.png), to one pixel
    def trace_ok(name, y, tol):
        with open(os.path.join(HERE, "golden", name)) as fh:
            t = json.load(fh)
        kk, hp = np.arange(len(y)), t["x_halfpixel"]
        for x, lo, hi in zip(t["x"], t["y_lo"], t["y_hi"]):
            seg = [np.interp(min(max(x + d, 0), len(y) - 1), kk, y) for d in (-hp, 0.0, hp)]
            assert max(seg) >= lo - tol and min(seg) <= hi + tol, (name, x)
    trace_ok("step_response_png_trace.json", np.array(step), 5e-5)
    outputs = np.r_[0.0, [float(l[3]) for l in loop]]                                # outputs[0] = C x0 (main.rs:41)
    trace_ok("control_png_trace.json", outputs, 2e-3)
