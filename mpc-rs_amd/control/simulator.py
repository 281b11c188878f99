"""Mirror of src/control/simulator.rs; the rollout runs on the GPU through mpcx_rollout_f64."""
from __future__ import annotations

import math

import numpy as np

from .. import ffi
from .model import DiscreteStateSpaceModel

_ctx = None


def context() -> ffi.Context:
    """The process-wide default context (device 0); created on first use, fails when no GPU can run the library."""
    global _ctx
    if _ctx is None:
        _ctx = ffi.Context(0)
    return _ctx


def _system_simulate(model: DiscreteStateSpaceModel, mat_u: np.ndarray, x0: np.ndarray):
    """simulator.rs:5-35: returns (Y r x T, X n x (T+1))."""
    a, b, c = model.get_mat_a(), model.get_mat_b(), model.get_mat_c()
    mat_u = np.asarray(mat_u, dtype=np.float64)
    if mat_u.shape[0] != b.shape[1]:
        raise ValueError("Matrix multiplication dimensions mismatch.")  # nalgebra panics at simulator.rs:21
    out = context().rollout(a[None], b[None], c[None], np.asarray(x0, dtype=np.float64).reshape(1, -1), mat_u[None])
    return out["Y"][0], out["X"][0]


def compute_system_response(model: DiscreteStateSpaceModel, input: np.ndarray, initial_state: np.ndarray) -> list:  # noqa: A002
    """simulator.rs:37-48: row 0 of Y as a list."""
    y, _ = _system_simulate(model, input, initial_state)
    return [float(v) for v in y[0]]


def step(model: DiscreteStateSpaceModel, duration: float):
    """simulator.rs:50-63: zero initial state, floor(duration / dt) samples of a unit input (one input assumed).
    Returns (response, input, states)."""
    initial_state = np.zeros(model.get_mat_a().shape[0])
    n_samples = int(math.floor(duration / model.get_sampling_dt()))
    u = np.ones((1, n_samples))
    response, states = _system_simulate(model, u, initial_state)
    return response, u, states
