"""Second, structurally independent CPU restatement of the reference's MPC hot path, in numpy.

TEST INFRASTRUCTURE ONLY (the checker, never the product): imported by tests/, tools/gen_golden.py and
``__graft_entry__.smoke()``.  PARITY UNPINNED: the reference holds no fixtures for this path and cannot be
built here, so this file and ``mpc_oracle.c`` are two independent readings of the same reference lines that
must agree with each other to ~1e-13; neither is the output of the Rust code.

Where ``mpc_oracle.c`` follows the reference's loop structure (powers re-accumulated as ``pow_a = pow_a * A``),
this file writes each block from its closed form with ``numpy.linalg.matrix_power`` so that an indexing slip in
one restatement cannot hide in the other.  Matrices are ordinary 2-D numpy arrays (row, col).
"""
from __future__ import annotations

import math

import numpy as np


def controller_weights(m: int, r: int, f: int, v: int, q0: float, q_other: float, pred_weight: float):
    """W1, W2, W3 = W1^T (W2 W1), W4 exactly as Controller::new builds them (controller.rs:149-185).

    The W2 and W4 diagonal blocks are ``from_element`` blocks: every entry of the block carries the weight
    (controller.rs:164-165, 184), not only its diagonal.
    """
    vm, fr = v * m, f * r
    w1 = np.zeros((vm, vm))
    for i in range(v):
        w1[i * m:(i + 1) * m, i * m:(i + 1) * m] = np.eye(m)
        if i >= 1:
            w1[i * m:(i + 1) * m, (i - 1) * m:i * m] = -np.eye(m)
    w2 = np.zeros((vm, vm))
    for i in range(v):
        w2[i * m:(i + 1) * m, i * m:(i + 1) * m] = np.full((m, m), q0 if i == 0 else q_other)
    w3 = w1.T @ (w2 @ w1)
    w4 = np.zeros((fr, fr))
    for i in range(f):
        w4[i * r:(i + 1) * r, i * r:(i + 1) * r] = np.full((r, r), pred_weight)
    return w1, w2, w3, w4


def lifted_matrices(A, B, C, f: int, v: int):
    """O (Phi) and M (Gamma) from their closed forms (controller.rs:32-90).

    Block row i of O is C A^(i+1).  Block (i, k) of M is C A^(i-k) B for k <= i and k < v-1 or i < v;
    for rows i >= v the last block column holds C (sum_{s=0}^{i-v+1} A^s) B (controller.rs:67-80): the
    input is held at its last value beyond the control horizon.
    """
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    C = np.asarray(C, dtype=np.float64)
    n, m, r = A.shape[0], B.shape[1], C.shape[0]
    mp = np.linalg.matrix_power
    O = np.zeros((f * r, n))
    for i in range(f):
        O[i * r:(i + 1) * r, :] = C @ mp(A, i + 1)
    M = np.zeros((f * r, v * m))
    for i in range(f):
        for k in range(min(i, v - 1) + 1):
            if i >= v and k == v - 1:
                acc = np.zeros((n, n))
                for s in range(i - v + 2):
                    acc = acc + mp(A, s)
                blk = C @ acc @ B
            else:
                blk = C @ mp(A, i - k) @ B
            M[i * r:(i + 1) * r, k * m:(k + 1) * m] = blk
    return O, M


def hessian(M, W3, W4):
    """H = M^T W4 M + W3, the matrix handed to try_inverse at controller.rs:92-94."""
    return M.T @ W4 @ M + W3


def gradient(O, M, W4, x0, ref):
    """fvec = M^T W4 (O x0 - ref); the reference applies H^-1 M^T W4 to (ref - O x) (controller.rs:96,125-130)."""
    return M.T @ W4 @ (O @ np.asarray(x0, dtype=np.float64).reshape(-1) - np.asarray(ref, dtype=np.float64).reshape(-1))


def gain_matrix(M, W3, W4):
    """gain = (M^T W4 M + W3)^-1 M^T W4 (controller.rs:92-96)."""
    return np.linalg.solve(hessian(M, W3, W4), M.T @ W4)


def system_simulate(A, B, C, U, x0):
    """(Y r x T, X n x (T+1)) as simulator.rs:5-35 returns them; U is m x T."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    C = np.asarray(C, dtype=np.float64)
    U = np.asarray(U, dtype=np.float64)
    n, r, T = A.shape[0], C.shape[0], U.shape[1]
    X = np.zeros((n, T + 1))
    Y = np.zeros((r, T))
    if T > 0:
        X[:, 0] = np.asarray(x0, dtype=np.float64).reshape(-1)
    for i in range(T):
        Y[:, i] = C @ X[:, i]
        X[:, i + 1] = A @ X[:, i] + B @ U[:, i]
    return Y, X


def step_n_samples(duration: float, dt: float) -> int:
    """simulator.rs:57."""
    return int(math.floor(duration / dt))


def c2d_backward_euler(Ac, Bc, dt: float):
    """A = (I - Ac dt)^-1, B = A (Bc dt) (model.rs:55-58, 138-141)."""
    Ac = np.asarray(Ac, dtype=np.float64)
    Bc = np.asarray(Bc, dtype=np.float64)
    A = np.linalg.inv(np.eye(Ac.shape[0]) - Ac * dt)
    return A, A @ (Bc * dt)


def dc_motor_continuous(b=0.1, j=0.01, k=0.01, l=0.5, r=1.0):
    """model.rs:27-31 (defaults) and :47-52."""
    Ac = np.array([[-b / j, k / j], [-k / l, -r / l]])
    Bc = np.array([[0.0], [1.0 / l]])
    Cc = np.array([[1.0, 0.0]])
    return Ac, Bc, Cc


def two_mass_continuous(m1=2.0, m2=2.0, k1=100.0, k2=200.0, d1=1.0, d2=5.0):
    """model.rs:107-112 (defaults) and :128-135."""
    Ac = np.array([
        [0.0, 1.0, 0.0, 0.0],
        [-(k1 + k2) / m1, -(d1 + d2) / m1, k2 / m1, d2 / m1],
        [0.0, 0.0, 0.0, 1.0],
        [k2 / m2, d2 / m2, -k2 / m2, -d2 / m2],
    ])
    Bc = np.array([[0.0], [0.0], [0.0], [1.0 / m2]])
    Cc = np.array([[1.0, 0.0, 0.0, 0.0]])
    return Ac, Bc, Cc


def pulse_trajectory(time_steps: int):
    """trajectory.rs:3-11; raises where copy_from would panic on a shape mismatch."""
    third = time_steps // 3
    if time_steps - 2 * time_steps // 3 != third:
        raise ValueError("copy_from shape mismatch (trajectory.rs:7-8)")
    traj = np.zeros((time_steps, 1))
    traj[:third, 0] = 1.0
    traj[2 * time_steps // 3:, 0] = 1.0
    return traj


def closed_loop(A, B, C, O, gain, traj_total, steps: int, x0):
    """main.rs:51-71 with compute_control_input (controller.rs:117-135) and propagate_dynamics (:101-115)."""
    A = np.asarray(A)
    n, r, f = A.shape[0], C.shape[0], O.shape[0] // C.shape[0]
    states = np.zeros((n, steps + 1))
    outputs = np.zeros((r, steps + 1))
    inputs = np.zeros(steps)
    states[:, 0] = np.asarray(x0).reshape(-1)
    outputs[:, 0] = C @ states[:, 0]
    traj = np.asarray(traj_total).reshape(-1)
    for t in range(steps):
        s = traj[t:t + f * r] - O @ states[:, t]
        u = (gain @ s)[0]
        states[:, t + 1] = A @ states[:, t] + B[:, 0] * u
        outputs[:, t + 1] = C @ states[:, t]
        inputs[t] = u
    return states, outputs, inputs
