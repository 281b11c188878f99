"""Seeded synthetic plant batches (the benchmark and test workloads).

Every instance draws from its own counter-based SplitMix64 stream keyed by (seed, global instance index), so a
shard of the batch is the same set of plants whatever the number of ranks.  Continuous-time plants are
discretised with the reference's backward-Euler rule, A = (I - Ac dt)^-1, B = A (Bc dt) (model.rs:55-58).

Workloads (dims n, m, r, horizon f = v):
  "msd"      2, 1, 1, 20   mass-spring-damper, dt = 0.01, M~U(0.5,2), k~U(50,200), d~U(0.5,5)
  "two_mass" 4, 1, 1, 20   the reference's two-mass plant (model.rs:128-135) with each default parameter
                           (model.rs:107-112) scaled by 1 + 0.1 U(-1,1), dt = 0.05; weights of main.rs:44-46
  "quadrotor" 12, 4, 12, 50  hover linearisation, C = I, mass~U(0.8,1.2), inertia~U(0.005,0.02), dt = 0.02
  "long"     8, 2, 2, 400  random contractive plant, dt = 0.05
"""
from __future__ import annotations

import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

WORKLOADS = {
    "msd": dict(n=2, m=1, r=1, f=20, v=20, T=200, seed=0x5EED0002),
    "two_mass": dict(n=4, m=1, r=1, f=20, v=20, T=200, seed=0x5EED0005),
    "quadrotor": dict(n=12, m=4, r=12, f=50, v=50, T=200, seed=0x5EED0003),
    "long": dict(n=8, m=2, r=2, f=400, v=400, T=400, seed=0x5EED0004),
}


def uniform(seed: int, idx: np.ndarray, draw: int) -> np.ndarray:
    """Draw number `draw` of each stream in `idx`: U[0, 1) with 53 random bits."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) ^ idx.astype(np.uint64)) * np.uint64(0xD1342543DE82EF95) + np.uint64(draw + 1) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


class _Draws:
    def __init__(self, seed, idx):
        self.seed, self.idx, self.k = seed, idx, 0

    def u(self, lo=0.0, hi=1.0):
        x = uniform(self.seed, self.idx, self.k)
        self.k += 1
        return lo + (hi - lo) * x

    def mat(self, rows, cols, lo=-1.0, hi=1.0):
        out = np.empty((self.idx.size, rows, cols))
        for i in range(rows):
            for j in range(cols):
                out[:, i, j] = self.u(lo, hi)
        return out


def c2d_backward_euler(Ac: np.ndarray, Bc: np.ndarray, dt: float):
    """Batched model.rs:55-58."""
    n = Ac.shape[-1]
    A = np.linalg.inv(np.eye(n) - Ac * dt)
    return A, A @ (Bc * dt)


def block_weights(m: int, r: int, v: int, q0: float, q_other: float, pred_weight: float, all_elements: bool = True):
    """W3 (v m x v m) and the r x r block of W4 as Controller::new builds them (controller.rs:149-185).

    all_elements=True reproduces the reference's ``from_element`` blocks (every entry of a block carries the
    weight); False puts the weight on the block diagonals only (q I), which keeps H non-singular for m, r > 1.
    """
    vm = v * m
    w1 = np.zeros((vm, vm))
    w2 = np.zeros((vm, vm))
    blk = (lambda q: np.full((m, m), q)) if all_elements else (lambda q: q * np.eye(m))
    for i in range(v):
        w1[i * m:(i + 1) * m, i * m:(i + 1) * m] = np.eye(m)
        if i >= 1:
            w1[i * m:(i + 1) * m, (i - 1) * m:i * m] = -np.eye(m)
        w2[i * m:(i + 1) * m, i * m:(i + 1) * m] = blk(q0 if i == 0 else q_other)
    w3 = w1.T @ (w2 @ w1)
    q = np.full((r, r), pred_weight) if all_elements else pred_weight * np.eye(r)
    return w3, q


def dense_w4(q_block: np.ndarray, f: int) -> np.ndarray:
    r = q_block.shape[0]
    w4 = np.zeros((f * r, f * r))
    for i in range(f):
        w4[i * r:(i + 1) * r, i * r:(i + 1) * r] = q_block
    return w4


def make_batch(name: str, batch: int, start: int = 0, f: int | None = None, v: int | None = None, seed: int | None = None):
    """Returns a dict with A (batch,n,n), B (batch,n,m), C (batch,r,n), x0 (batch,n), ref (batch,f*r),
    W3 (vm,vm), Q (r,r), U (m,T) shared step input, and the dims."""
    cfg = dict(WORKLOADS[name])
    if f is not None:
        cfg["f"] = f
    if v is not None:
        cfg["v"] = v
    if seed is not None:
        cfg["seed"] = seed
    n, m, r, f, v = cfg["n"], cfg["m"], cfg["r"], cfg["f"], cfg["v"]
    idx = np.arange(start, start + batch, dtype=np.uint64)
    d = _Draws(cfg["seed"], idx)
    if name == "msd":
        M, k, dmp = d.u(0.5, 2.0), d.u(50.0, 200.0), d.u(0.5, 5.0)
        Ac = np.zeros((batch, 2, 2))
        Ac[:, 0, 1] = 1.0
        Ac[:, 1, 0] = -k / M
        Ac[:, 1, 1] = -dmp / M
        Bc = np.zeros((batch, 2, 1))
        Bc[:, 1, 0] = 1.0 / M
        Cm = np.tile(np.array([[1.0, 0.0]]), (batch, 1, 1))
        dt = 0.01
        w3, q = block_weights(m, r, v, 1.1e-9, 1e-4, 10.0)
    elif name == "two_mass":
        base = np.array([2.0, 2.0, 100.0, 200.0, 1.0, 5.0])
        m1, m2, k1, k2, d1, d2 = (base[i] * (1.0 + 0.1 * d.u(-1.0, 1.0)) for i in range(6))
        Ac = np.zeros((batch, 4, 4))
        Ac[:, 0, 1] = 1.0
        Ac[:, 1, 0] = -(k1 + k2) / m1
        Ac[:, 1, 1] = -(d1 + d2) / m1
        Ac[:, 1, 2] = k2 / m1
        Ac[:, 1, 3] = d2 / m1
        Ac[:, 2, 3] = 1.0
        Ac[:, 3, 0] = k2 / m2
        Ac[:, 3, 1] = d2 / m2
        Ac[:, 3, 2] = -k2 / m2
        Ac[:, 3, 3] = -d2 / m2
        Bc = np.zeros((batch, 4, 1))
        Bc[:, 3, 0] = 1.0 / m2
        Cm = np.tile(np.array([[1.0, 0.0, 0.0, 0.0]]), (batch, 1, 1))
        dt = 0.05
        w3, q = block_weights(m, r, v, 1.1e-9, 1e-4, 10.0)
    elif name == "quadrotor":
        g = 9.81
        mass = d.u(0.8, 1.2)
        J = [d.u(0.005, 0.02) for _ in range(3)]
        Ac = np.zeros((batch, 12, 12))
        for i in range(3):
            Ac[:, i, 3 + i] = 1.0      # p' = v
            Ac[:, 6 + i, 9 + i] = 1.0  # eta' = omega
        Ac[:, 3, 7] = g                # vx' = g theta
        Ac[:, 4, 6] = -g               # vy' = -g phi
        Bc = np.zeros((batch, 12, 4))
        Bc[:, 5, 0] = 1.0 / mass
        for i in range(3):
            Bc[:, 9 + i, 1 + i] = 1.0 / J[i]
        Cm = np.tile(np.eye(12), (batch, 1, 1))
        dt = 0.02
        w3, q = block_weights(m, r, v, 1e-3, 1e-2, 10.0, all_elements=False)
    elif name == "long":
        Lm = d.mat(8, 8) / np.sqrt(8.0)
        S = d.mat(8, 8) / np.sqrt(8.0)
        Ac = -(Lm @ Lm.transpose(0, 2, 1) + 0.5 * np.eye(8)) + 0.3 * (S - S.transpose(0, 2, 1))
        Bc = d.mat(8, 2)
        Cm = d.mat(2, 8)
        dt = 0.05
        w3, q = block_weights(m, r, v, 1e-3, 1e-2, 10.0, all_elements=False)
    else:
        raise KeyError(name)
    A, B = c2d_backward_euler(Ac, Bc, dt)
    x0 = np.stack([d.u(-1.0, 1.0) for _ in range(n)], axis=1)
    ref = np.stack([d.u(0.0, 1.0) for _ in range(f * r)], axis=1) if f * r <= 64 else _ref_from_few(d, f, r)
    T = cfg["T"]
    return dict(name=name, A=A, B=B, C=Cm, x0=x0, ref=ref, W3=w3, Q=q, U=np.ones((m, T)), n=n, m=m, r=r, f=f, v=v, T=T,
                dt=dt, batch=batch)


def _ref_from_few(d: _Draws, f: int, r: int) -> np.ndarray:
    """Long references: a per-instance pulse (level, phase) instead of f*r independent draws."""
    level = d.u(0.5, 1.5)
    phase = d.u(0.0, 1.0)
    t = (np.arange(f * r) // r)[None, :] / float(f)
    return level[:, None] * (((t + phase[:, None]) % 1.0) < 0.5)
