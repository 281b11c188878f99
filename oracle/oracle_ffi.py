"""ctypes binding of the C oracle (``mpc_oracle.c``).  TEST INFRASTRUCTURE ONLY.

Loads ``oracle/_build/libmpc_oracle.so`` (or the ``_fma`` variant) built by ``oracle/Makefile``.  Arrays cross
the boundary column-major, as in nalgebra; this module converts from and to ordinary (row, col) numpy arrays.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_P = C.POINTER(C.c_double)
_LIBS: dict[str, C.CDLL] = {}

ORC_OK, ORC_ERR_INVERSION, ORC_ERR_SHAPE = 0, 1, 2


def lib(variant: str = "") -> C.CDLL:
    """variant "" = separate multiply/add, "fma" = fused contraction allowed."""
    if variant in _LIBS:
        return _LIBS[variant]
    name = "libmpc_oracle%s.so" % ("_" + variant if variant else "")
    path = os.path.join(_HERE, "_build", name)
    if not variant and os.environ.get("MPC_ORACLE_LIB"):  # e.g. the sanitizer build of `make -C oracle asan`
        path = os.environ["MPC_ORACLE_LIB"]
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
    L = C.CDLL(path)
    i, d, i64 = C.c_int, C.c_double, C.c_int64
    L.orc_controller_weights.argtypes = [i, i, i, i, d, d, d, _P, _P, _P, _P]
    L.orc_controller_weights.restype = None
    L.orc_form_lifted_matrices.argtypes = [_P, _P, _P, i, i, i, i, i, _P, _P, _P, _P, _P, _P]
    L.orc_form_lifted_matrices.restype = i
    L.orc_gradient.argtypes = [_P, _P, _P, _P, _P, i, i, i, i, i, _P]
    L.orc_gradient.restype = None
    L.orc_compute_control_input.argtypes = [_P, _P, _P, i, _P, i, i, i, i, i, _P]
    L.orc_compute_control_input.restype = d
    L.orc_propagate_dynamics.argtypes = [_P, _P, _P, i, i, i, _P, _P, _P, _P]
    L.orc_propagate_dynamics.restype = None
    L.orc_system_simulate.argtypes = [_P, _P, _P, i, i, i, _P, i, _P, _P, _P]
    L.orc_system_simulate.restype = None
    L.orc_step_n_samples.argtypes = [d, d]
    L.orc_step_n_samples.restype = i
    L.orc_c2d_backward_euler.argtypes = [_P, _P, i, i, d, _P, _P]
    L.orc_c2d_backward_euler.restype = i
    L.orc_dc_motor_continuous.argtypes = [d] * 5 + [_P] * 3
    L.orc_dc_motor_continuous.restype = None
    L.orc_two_mass_continuous.argtypes = [d] * 6 + [_P] * 3
    L.orc_two_mass_continuous.restype = None
    L.orc_pulse_trajectory.argtypes = [i, _P]
    L.orc_pulse_trajectory.restype = i
    L.orc_closed_loop.argtypes = [_P, _P, _P, i, i, i, i, i, _P, _P, _P, i, _P, _P, _P, _P]
    L.orc_closed_loop.restype = None
    L.orc_closed_loop_mimo.argtypes = [_P, _P, _P, i, i, i, i, i, _P, _P, _P, i, _P, _P, _P, _P]
    L.orc_closed_loop_mimo.restype = None
    L.orc_try_inverse.argtypes = [_P, i]
    L.orc_try_inverse.restype = i
    L.orc_qp_build_batch.argtypes = [i64, i, i, i, i, i, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, i]
    L.orc_qp_build_batch.restype = i
    L.orc_qp_build_batch_fast.argtypes = [i64, i, i, i, i, i, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, i]
    L.orc_qp_build_batch_fast.restype = i
    L.orc_rollout_batch.argtypes = [i64, i, i, i, i64, _P, _P, _P, _P, _P, i, _P, _P, i]
    L.orc_rollout_batch.restype = i
    _LIBS[variant] = L
    return L


def _cm(a) -> np.ndarray:
    """Column-major flat copy of a (row, col) array."""
    return np.array(np.asarray(a, dtype=np.float64).T, dtype=np.float64, order="C", copy=True).reshape(-1)


def _p(a):
    return None if a is None else a.ctypes.data_as(_P)


def _from_cm(flat: np.ndarray, rows: int, cols: int) -> np.ndarray:
    return flat.reshape(cols, rows).T.copy()


def controller_weights(m, r, f, v, q0, q_other, pred_weight, variant=""):
    vm, fr = v * m, f * r
    w1, w2, w3 = (np.zeros(vm * vm) for _ in range(3))
    w4 = np.zeros(fr * fr)
    lib(variant).orc_controller_weights(m, r, f, v, q0, q_other, pred_weight, _p(w1), _p(w2), _p(w3), _p(w4))
    return _from_cm(w1, vm, vm), _from_cm(w2, vm, vm), _from_cm(w3, vm, vm), _from_cm(w4, fr, fr)


def form_lifted_matrices(A, B, Cm, f, v, W3, W4, want_gain=True, variant=""):
    """Returns (status, O, M, H, gain); gain is None when not requested or when the inversion failed."""
    A, B, Cm = (np.asarray(x, dtype=np.float64) for x in (A, B, Cm))
    n, m, r = A.shape[0], B.shape[1], Cm.shape[0]
    o = np.zeros(f * r * n)
    mm = np.zeros(f * r * v * m)
    h = np.zeros(v * m * v * m)
    g = np.zeros(v * m * f * r) if want_gain else None
    a_, b_, c_, w3_, w4_ = _cm(A), _cm(B), _cm(Cm), _cm(W3), _cm(W4)
    st = lib(variant).orc_form_lifted_matrices(_p(a_), _p(b_), _p(c_), n, m, r, f, v, _p(w3_), _p(w4_), _p(o), _p(mm),
                                               _p(h), _p(g))
    gain = _from_cm(g, v * m, f * r) if (want_gain and st == ORC_OK) else None
    return st, _from_cm(o, f * r, n), _from_cm(mm, f * r, v * m), _from_cm(h, v * m, v * m), gain


def gradient(O, M, W4, x0, ref, variant=""):
    fr, n = O.shape
    vm = M.shape[1]
    # m, r enter only through the products f*r and v*m: pass them factored as (m=vm, v=1), (r=fr, f=1)
    out = np.zeros(vm)
    o_, m_, w_, x_, r_ = _cm(O), _cm(M), _cm(W4), _cm(np.asarray(x0).reshape(-1, 1)), _cm(np.asarray(ref).reshape(-1, 1))
    lib(variant).orc_gradient(_p(o_), _p(m_), _p(w_), _p(x_), _p(r_), n, vm, fr, 1, 1, _p(out))
    return out


def system_simulate(A, B, Cm, U, x0, variant=""):
    A, B, Cm, U = (np.asarray(x, dtype=np.float64) for x in (A, B, Cm, U))
    n, m, r, T = A.shape[0], B.shape[1], Cm.shape[0], U.shape[1]
    X = np.zeros(n * (T + 1))
    Y = np.zeros(max(r * T, 1))
    a_, b_, c_, u_, x_ = _cm(A), _cm(B), _cm(Cm), _cm(U), _cm(np.asarray(x0).reshape(-1, 1))
    lib(variant).orc_system_simulate(_p(a_), _p(b_), _p(c_), n, m, r, _p(u_), T, _p(x_), _p(Y), _p(X))
    return _from_cm(Y[:r * T], r, T), _from_cm(X, n, T + 1)


def propagate_dynamics(A, B, Cm, u, x, variant=""):
    A, B, Cm = (np.asarray(q, dtype=np.float64) for q in (A, B, Cm))
    n, m, r = A.shape[0], B.shape[1], Cm.shape[0]
    xn, y = np.zeros(n), np.zeros(r)
    a_, b_, c_ = _cm(A), _cm(B), _cm(Cm)
    u_ = np.ascontiguousarray(np.asarray(u, dtype=np.float64).reshape(-1))
    x_ = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))
    lib(variant).orc_propagate_dynamics(_p(a_), _p(b_), _p(c_), n, m, r, _p(u_), _p(x_), _p(xn), _p(y))
    return xn, y


def c2d_backward_euler(Ac, Bc, dt, variant=""):
    Ac, Bc = np.asarray(Ac, dtype=np.float64), np.asarray(Bc, dtype=np.float64)
    n, m = Ac.shape[0], Bc.shape[1]
    A, B = np.zeros(n * n), np.zeros(n * m)
    ac_, bc_ = _cm(Ac), _cm(Bc)
    st = lib(variant).orc_c2d_backward_euler(_p(ac_), _p(bc_), n, m, dt, _p(A), _p(B))
    return st, _from_cm(A, n, n), _from_cm(B, n, m)


def dc_motor_continuous(b=0.1, j=0.01, k=0.01, l=0.5, r=1.0, variant=""):
    Ac, Bc, Cc = np.zeros(4), np.zeros(2), np.zeros(2)
    lib(variant).orc_dc_motor_continuous(b, j, k, l, r, _p(Ac), _p(Bc), _p(Cc))
    return _from_cm(Ac, 2, 2), _from_cm(Bc, 2, 1), _from_cm(Cc, 1, 2)


def two_mass_continuous(m1=2.0, m2=2.0, k1=100.0, k2=200.0, d1=1.0, d2=5.0, variant=""):
    Ac, Bc, Cc = np.zeros(16), np.zeros(4), np.zeros(4)
    lib(variant).orc_two_mass_continuous(m1, m2, k1, k2, d1, d2, _p(Ac), _p(Bc), _p(Cc))
    return _from_cm(Ac, 4, 4), _from_cm(Bc, 4, 1), _from_cm(Cc, 1, 4)


def pulse_trajectory(time_steps, variant=""):
    t = np.zeros(max(time_steps, 1))
    st = lib(variant).orc_pulse_trajectory(time_steps, _p(t))
    return st, t[:time_steps].reshape(-1, 1)


def closed_loop(A, B, Cm, f, v, O, gain, traj_total, steps, x0, variant=""):
    A, B, Cm = (np.asarray(q, dtype=np.float64) for q in (A, B, Cm))
    n, m, r = A.shape[0], B.shape[1], Cm.shape[0]
    states, outputs, inputs = np.zeros(n * (steps + 1)), np.zeros(r * (steps + 1)), np.zeros(max(steps, 1))
    a_, b_, c_, o_, g_ = _cm(A), _cm(B), _cm(Cm), _cm(O), _cm(gain)
    t_ = np.ascontiguousarray(np.asarray(traj_total, dtype=np.float64).reshape(-1))
    x_ = np.ascontiguousarray(np.asarray(x0, dtype=np.float64).reshape(-1))
    lib(variant).orc_closed_loop(_p(a_), _p(b_), _p(c_), n, m, r, f, v, _p(o_), _p(g_), _p(t_), steps, _p(x_),
                                 _p(states), _p(outputs), _p(inputs))
    return _from_cm(states, n, steps + 1), _from_cm(outputs, r, steps + 1), inputs[:steps]


def closed_loop_mimo(A, B, Cm, f, v, O, gain, ref, steps, x0, variant=""):
    """ref (r, steps + f).  Returns X (n, steps+1), U (m, steps), Y (r, steps)."""
    A, B, Cm = (np.asarray(q, dtype=np.float64) for q in (A, B, Cm))
    n, m, r = A.shape[0], B.shape[1], Cm.shape[0]
    X, U, Y = np.zeros(n * (steps + 1)), np.zeros(max(m * steps, 1)), np.zeros(max(r * steps, 1))
    a_, b_, c_, o_, g_ = _cm(A), _cm(B), _cm(Cm), _cm(O), _cm(gain)
    ref = np.asarray(ref, dtype=np.float64)
    assert ref.shape == (r, steps + f)
    r_ = _cm(ref)
    x_ = np.ascontiguousarray(np.asarray(x0, dtype=np.float64).reshape(-1))
    lib(variant).orc_closed_loop_mimo(_p(a_), _p(b_), _p(c_), n, m, r, f, v, _p(o_), _p(g_), _p(r_), steps, _p(x_),
                                      _p(X), _p(U), _p(Y))
    return _from_cm(X, n, steps + 1), _from_cm(U[:m * steps], m, steps), _from_cm(Y[:r * steps], r, steps)


def try_inverse(a, variant=""):
    a = np.asarray(a, dtype=np.float64)
    n = a.shape[0]
    flat = _cm(a)
    ok = lib(variant).orc_try_inverse(_p(flat), n)
    return bool(ok), _from_cm(flat, n, n)


def qp_build_batch(A, B, Cm, W3, W4, x0, ref, f, v, threads=1, variant="", want=("Phi", "Gamma", "H", "fvec"),
                   timing=None, fast=False):
    """Instance-major batch: A (batch, n, n) etc. as (row, col) arrays per instance; W3/W4 shared.

    Returns dict of (batch, rows, cols) arrays and the thread count used.  When `timing` is a dict, the wall
    time of the C call alone (buffers already laid out) is stored in timing["seconds"].
    """
    import time as _time
    A, B, Cm = (np.asarray(q, dtype=np.float64) for q in (A, B, Cm))
    batch, n = A.shape[0], A.shape[1]
    m, r = B.shape[2], Cm.shape[1]
    tr = lambda q: np.ascontiguousarray(np.transpose(q, (0, 2, 1))).reshape(-1)
    a_, b_, c_ = tr(A), tr(B), tr(Cm)
    w3_, w4_ = _cm(W3), _cm(W4)
    x_ = np.ascontiguousarray(np.asarray(x0, dtype=np.float64).reshape(batch, n)).reshape(-1)
    r_ = np.ascontiguousarray(np.asarray(ref, dtype=np.float64).reshape(batch, f * r)).reshape(-1)
    bufs = {
        "Phi": np.zeros(batch * f * r * n) if "Phi" in want else None,
        "Gamma": np.zeros(batch * f * r * v * m) if "Gamma" in want else None,
        "H": np.zeros(batch * v * m * v * m) if "H" in want else None,
        "fvec": np.zeros(batch * v * m) if "fvec" in want else None,
    }
    if fast:  # the speed-oriented CPU implementation; takes the r x r block of the (block-diagonal) W4
        fn = lib(variant).orc_qp_build_batch_fast
        w4_ = _cm(np.asarray(W4, dtype=np.float64)[:r, :r])
    else:
        fn = lib(variant).orc_qp_build_batch
    t0 = _time.perf_counter()
    used = fn(batch, n, m, r, f, v, _p(a_), _p(b_), _p(c_), _p(w3_), _p(w4_), _p(x_), _p(r_), _p(bufs["Phi"]),
              _p(bufs["Gamma"]), _p(bufs["H"]), _p(bufs["fvec"]), threads)
    if timing is not None:
        timing["seconds"] = _time.perf_counter() - t0
    shapes = {"Phi": (f * r, n), "Gamma": (f * r, v * m), "H": (v * m, v * m), "fvec": (v * m, 1)}
    out = {}
    for k, buf in bufs.items():
        if buf is not None:
            rows, cols = shapes[k]
            out[k] = np.transpose(buf.reshape(batch, cols, rows), (0, 2, 1)).copy()
    return out, used


def rollout_batch(A, B, Cm, x0, U, threads=1, variant="", timing=None):
    """A (batch,n,n), B (batch,n,m), C (batch,r,n), x0 (batch,n), U (batch,m,T) or shared (m,T).

    Returns X (batch,n,T+1), Y (batch,r,T), threads used.
    """
    A, B, Cm, U = (np.asarray(q, dtype=np.float64) for q in (A, B, Cm, U))
    batch, n = A.shape[0], A.shape[1]
    m, r = B.shape[2], Cm.shape[1]
    shared = U.ndim == 2
    T = U.shape[-1]
    tr = lambda q: np.ascontiguousarray(np.transpose(q, (0, 2, 1))).reshape(-1)
    a_, b_, c_ = tr(A), tr(B), tr(Cm)
    u_ = _cm(U) if shared else tr(U)
    x_ = np.ascontiguousarray(np.asarray(x0, dtype=np.float64).reshape(batch, n)).reshape(-1)
    X = np.zeros(batch * n * (T + 1))
    Y = np.zeros(max(batch * r * T, 1))
    import time as _time
    fn = lib(variant).orc_rollout_batch
    t0 = _time.perf_counter()
    used = fn(batch, n, m, r, T, _p(a_), _p(b_), _p(c_), _p(x_), _p(u_), int(shared), _p(X), _p(Y), threads)
    if timing is not None:
        timing["seconds"] = _time.perf_counter() - t0
    Xo = np.transpose(X.reshape(batch, T + 1, n), (0, 2, 1)).copy()
    Yo = np.transpose(Y[:batch * r * T].reshape(batch, T, r), (0, 2, 1)).copy()
    return Xo, Yo, used
