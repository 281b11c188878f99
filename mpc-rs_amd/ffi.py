"""ctypes binding of libmpcx.so, the C ABI declared in include/mpcx.h.

The library is the hand-written HIP (gfx950) implementation; there is no other compute path.  Loading fails
with an explicit error when the shared object is missing, and every call raises :class:`MpcxError` on a
non-zero status (the message comes from ``mpcx_last_error``).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import layout as L

_HERE = os.path.dirname(os.path.abspath(__file__))
# MPCX_LIB: developer override used to A/B differently built copies of the library; the default is the in-tree build
LIB_PATH = os.environ.get("MPCX_LIB") or os.path.join(_HERE, "csrc", "libmpcx.so")

OK = 0
ERR_NULL, ERR_DIMS, ERR_LAYOUT, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_SINGULAR = -1, -2, -3, -4, -5, -6
WEIGHTS_BLOCK, WEIGHTS_DENSE = 0, 1
HESSIAN_AUTO, HESSIAN_CONTRACT, HESSIAN_STRUCTURED = 0, 1, 2
PLANT_DC_MOTOR, PLANT_TWO_MASS = 0, 1
OPT_RELAY_CAP_BYTES = 1
OPT_QP_STREAMING = 2

# every symbol include/mpcx.h declares
EXPORTS = (
    "mpcx_version", "mpcx_array_elems", "mpcx_create", "mpcx_destroy", "mpcx_last_error", "mpcx_reserve", "mpcx_set_option",
    "mpcx_qp_build_f64", "mpcx_qp_build_f64_dev", "mpcx_rollout_f64", "mpcx_rollout_f64_dev",
    "mpcx_qp_solve_f64", "mpcx_qp_solve_f64_dev", "mpcx_qp_gain_f64", "mpcx_qp_gain_f64_dev",
    "mpcx_qp_gain_rows_f64", "mpcx_qp_gain_rows_f64_dev",
    "mpcx_closed_loop_f64", "mpcx_closed_loop_f64_dev",
    "mpcx_c2d_backward_euler_f64", "mpcx_c2d_backward_euler_f64_dev", "mpcx_c2d_backward_euler_f64_noalloc_dev",
    "mpcx_plants_f64", "mpcx_plants_f64_dev",
    "mpcx_stats_f64_dev", "mpcx_stats_f64", "mpcx_relayout_f64_dev",
    "mpcx_comm_get_id", "mpcx_comm_create", "mpcx_comm_destroy", "mpcx_comm_last_error", "mpcx_comm_rank", "mpcx_comm_world",
    "mpcx_gather_stats", "mpcx_gather_f64_dev", "mpcx_comm_synchronize",
)
COMM_ID_BYTES = 128
ERR_COMM = -7


class MpcxError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"mpcx error {code}: {message}")
        self.code = code
        self.message = message


class Dims(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("m", C.c_int32), ("r", C.c_int32), ("f", C.c_int32), ("v", C.c_int32),
        ("weights_mode", C.c_int32), ("hessian_algo", C.c_int32), ("u_shared", C.c_int32),
        ("batch", C.c_int64), ("T", C.c_int64), ("interleave", C.c_int64),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("count", C.c_int64), ("non_finite", C.c_int64), ("sum", C.c_double), ("sum_sq", C.c_double),
        ("max_abs", C.c_double), ("checksum", C.c_uint64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C mpc-rs_amd/csrc` (or __graft_entry__.build()). "
            "There is no fallback compute path.")
    lb = C.CDLL(LIB_PATH)
    vp, dp, i64 = C.c_void_p, C.c_void_p, C.c_int64
    lb.mpcx_version.restype = C.c_int
    lb.mpcx_array_elems.argtypes = [i64, i64, i64]
    lb.mpcx_array_elems.restype = i64
    lb.mpcx_create.argtypes = [C.POINTER(vp), C.c_int]
    lb.mpcx_create.restype = C.c_int
    lb.mpcx_destroy.argtypes = [vp]
    lb.mpcx_destroy.restype = None
    lb.mpcx_last_error.argtypes = [vp]
    lb.mpcx_last_error.restype = C.c_char_p
    lb.mpcx_reserve.argtypes = [vp, C.POINTER(Dims)]
    lb.mpcx_reserve.restype = C.c_int
    lb.mpcx_set_option.argtypes = [vp, C.c_int, i64]
    lb.mpcx_set_option.restype = C.c_int
    lb.mpcx_qp_build_f64.argtypes = [vp, C.POINTER(Dims)] + [dp] * 11
    lb.mpcx_qp_build_f64.restype = C.c_int
    lb.mpcx_qp_build_f64_dev.argtypes = [vp, C.POINTER(Dims)] + [dp] * 11 + [vp]
    lb.mpcx_qp_build_f64_dev.restype = C.c_int
    lb.mpcx_rollout_f64.argtypes = [vp, C.POINTER(Dims)] + [dp] * 7
    lb.mpcx_rollout_f64.restype = C.c_int
    lb.mpcx_rollout_f64_dev.argtypes = [vp, C.POINTER(Dims)] + [dp] * 7 + [vp]
    lb.mpcx_rollout_f64_dev.restype = C.c_int
    lb.mpcx_qp_solve_f64.argtypes = [vp, C.POINTER(Dims), dp, dp, dp, vp]
    lb.mpcx_qp_solve_f64.restype = C.c_int
    lb.mpcx_qp_solve_f64_dev.argtypes = [vp, C.POINTER(Dims), dp, dp, dp, vp, vp]
    lb.mpcx_qp_solve_f64_dev.restype = C.c_int
    lb.mpcx_qp_gain_f64.argtypes = [vp, C.POINTER(Dims), dp, dp, dp, dp, vp]
    lb.mpcx_qp_gain_f64.restype = C.c_int
    lb.mpcx_qp_gain_f64_dev.argtypes = [vp, C.POINTER(Dims), dp, dp, dp, dp, vp, vp]
    lb.mpcx_qp_gain_f64_dev.restype = C.c_int
    lb.mpcx_qp_gain_rows_f64.argtypes = [vp, C.POINTER(Dims), dp, dp, dp, dp, vp]
    lb.mpcx_qp_gain_rows_f64.restype = C.c_int
    lb.mpcx_qp_gain_rows_f64_dev.argtypes = [vp, C.POINTER(Dims), dp, dp, dp, dp, vp, vp]
    lb.mpcx_qp_gain_rows_f64_dev.restype = C.c_int
    lb.mpcx_closed_loop_f64.argtypes = [vp, C.POINTER(Dims)] + [dp] * 10
    lb.mpcx_closed_loop_f64.restype = C.c_int
    lb.mpcx_closed_loop_f64_dev.argtypes = [vp, C.POINTER(Dims)] + [dp] * 10 + [vp]
    lb.mpcx_closed_loop_f64_dev.restype = C.c_int
    lb.mpcx_c2d_backward_euler_f64.argtypes = [vp, C.POINTER(Dims), dp, dp, C.c_double, dp, dp, vp]
    lb.mpcx_c2d_backward_euler_f64.restype = C.c_int
    lb.mpcx_c2d_backward_euler_f64_dev.argtypes = [vp, C.POINTER(Dims), dp, dp, C.c_double, dp, dp, vp, vp]
    lb.mpcx_c2d_backward_euler_f64_dev.restype = C.c_int
    lb.mpcx_c2d_backward_euler_f64_noalloc_dev.argtypes = [vp, C.POINTER(Dims), dp, dp, C.c_double, dp, dp, vp, vp]
    lb.mpcx_c2d_backward_euler_f64_noalloc_dev.restype = C.c_int
    lb.mpcx_plants_f64.argtypes = [vp, C.c_int, i64, i64, dp, C.c_double, dp, dp, dp, vp]
    lb.mpcx_plants_f64.restype = C.c_int
    lb.mpcx_plants_f64_dev.argtypes = [vp, C.c_int, i64, i64, dp, C.c_double, dp, dp, dp, vp, vp]
    lb.mpcx_plants_f64_dev.restype = C.c_int
    lb.mpcx_stats_f64_dev.argtypes = [vp, dp, i64, vp, vp]
    lb.mpcx_stats_f64_dev.restype = C.c_int
    lb.mpcx_stats_f64.argtypes = [vp, dp, i64, C.POINTER(Stats)]
    lb.mpcx_stats_f64.restype = C.c_int
    lb.mpcx_relayout_f64_dev.argtypes = [vp, i64, i64, i64, dp, i64, dp, vp]
    lb.mpcx_relayout_f64_dev.restype = C.c_int
    lb.mpcx_comm_get_id.argtypes = [vp]
    lb.mpcx_comm_get_id.restype = C.c_int
    lb.mpcx_comm_create.argtypes = [C.POINTER(vp), vp, vp, C.c_int, C.c_int]
    lb.mpcx_comm_create.restype = C.c_int
    lb.mpcx_comm_destroy.argtypes = [vp]
    lb.mpcx_comm_destroy.restype = None
    lb.mpcx_comm_last_error.argtypes = [vp]
    lb.mpcx_comm_last_error.restype = C.c_char_p
    lb.mpcx_comm_rank.argtypes = [vp]
    lb.mpcx_comm_rank.restype = C.c_int
    lb.mpcx_comm_world.argtypes = [vp]
    lb.mpcx_comm_world.restype = C.c_int
    lb.mpcx_gather_stats.argtypes = [vp, dp, i64, vp, C.POINTER(Stats)]
    lb.mpcx_gather_stats.restype = C.c_int
    lb.mpcx_gather_f64_dev.argtypes = [vp, dp, i64, dp, C.c_int, vp]
    lb.mpcx_gather_f64_dev.restype = C.c_int
    lb.mpcx_comm_synchronize.argtypes = [vp]
    lb.mpcx_comm_synchronize.restype = C.c_int
    _lib = lb
    return lb


def comm_get_id() -> bytes:
    """MPCX_COMM_ID_BYTES opaque bytes that one rank creates and every rank passes to Comm()."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = lib().mpcx_comm_get_id(buf)
    if rc != OK:
        raise MpcxError(rc, lib().mpcx_comm_last_error(None).decode())
    return buf.raw


class Comm:
    """One rank of the library's RCCL communicator, bound to a Context (its device)."""

    def __init__(self, ctx: "Context", comm_id: bytes, world: int, rank: int):
        assert len(comm_id) == COMM_ID_BYTES
        self._lib = lib()
        h = C.c_void_p()
        rc = self._lib.mpcx_comm_create(C.byref(h), ctx._h, comm_id, world, rank)
        if rc != OK:
            raise MpcxError(rc, self._lib.mpcx_comm_last_error(None).decode() or "mpcx_comm_create failed")
        self._h, self.world, self.rank = h, world, rank

    def _check(self, rc: int):
        if rc != OK:
            raise MpcxError(rc, self._lib.mpcx_comm_last_error(self._h).decode())

    def gather_stats(self, buf_dev, count: int, stream=0) -> list:
        recs = (Stats * self.world)()
        self._check(self._lib.mpcx_gather_stats(self._h, buf_dev, count, stream, recs))
        return [r.as_dict() for r in recs]

    def gather_dev(self, src_dev, count: int, dst_dev, root=0, stream=0):
        self._check(self._lib.mpcx_gather_f64_dev(self._h, src_dev, count, dst_dev, root, stream))

    def synchronize(self):
        self._check(self._lib.mpcx_comm_synchronize(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mpcx_comm_destroy(self._h)
            self._h = None


def _hp(a):
    """Host pointer of a contiguous float64 numpy array (or None)."""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def make_dims(n, m, r, f=1, v=1, batch=1, T=0, interleave=L.INTERLEAVE_AOS, weights_mode=WEIGHTS_BLOCK,
              hessian_algo=HESSIAN_AUTO, u_shared=0) -> Dims:
    return Dims(n=n, m=m, r=r, f=f, v=v, weights_mode=weights_mode, hessian_algo=hessian_algo, u_shared=u_shared,
                batch=batch, T=T, interleave=interleave)


class Context:
    """One mpcx_ctx: bound to one device; not thread-safe."""

    def __init__(self, device: int = 0):
        self._lib = lib()
        h = C.c_void_p()
        rc = self._lib.mpcx_create(C.byref(h), device)
        if rc != OK:
            raise MpcxError(rc, "mpcx_create failed (no usable HIP device?)" if rc == ERR_NO_DEVICE else "mpcx_create failed")
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mpcx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int):
        if rc != OK:
            raise MpcxError(rc, self._lib.mpcx_last_error(self._h).decode())

    def reserve(self, dims: Dims):
        """Grow every scratch buffer the `_dev` calls of `dims` need, so that they do not allocate afterwards."""
        self._check(self._lib.mpcx_reserve(self._h, C.byref(dims)))

    def set_option(self, option: int, value: int):
        self._check(self._lib.mpcx_set_option(self._h, option, value))

    # ---- raw calls: pointers are ints / c_void_p (device) or numpy arrays (host) ----
    def qp_build_dev(self, dims: Dims, A, B, Cm, W3, W4, x0, ref, Phi, Gamma, H, fvec, stream=0):
        self._check(self._lib.mpcx_qp_build_f64_dev(self._h, C.byref(dims), A, B, Cm, W3, W4, x0, ref, Phi, Gamma, H,
                                                    fvec, stream))

    def rollout_dev(self, dims: Dims, A, B, Cm, x0, U, X, Y, stream=0):
        self._check(self._lib.mpcx_rollout_f64_dev(self._h, C.byref(dims), A, B, Cm, x0, U, X, Y, stream))

    def stats_dev(self, buf, count: int, out_dev, stream=0):
        self._check(self._lib.mpcx_stats_f64_dev(self._h, buf, count, out_dev, stream))

    def stats(self, buf_dev, count: int) -> dict:
        st = Stats()
        self._check(self._lib.mpcx_stats_f64(self._h, buf_dev, count, C.byref(st)))
        return st.as_dict()

    def relayout_dev(self, batch, elems, src_il, src, dst_il, dst, stream=0):
        self._check(self._lib.mpcx_relayout_f64_dev(self._h, batch, elems, src_il, src, dst_il, dst, stream))

    # ---- host-array convenience: (batch, rows, cols) numpy in, numpy out, through the host-pointer ABI ----
    def qp_build(self, A, B, Cm, f, v, W3=None, W4=None, x0=None, ref=None, interleave=L.INTERLEAVE_AOS,
                 weights_mode=WEIGHTS_BLOCK, hessian_algo=HESSIAN_AUTO, want=("Phi", "Gamma", "H", "fvec")):
        A = np.asarray(A, dtype=np.float64)
        B = np.asarray(B, dtype=np.float64)
        Cm = np.asarray(Cm, dtype=np.float64)
        batch, n = A.shape[0], A.shape[1]
        m, r = B.shape[2], Cm.shape[1]
        d = make_dims(n, m, r, f, v, batch, 0, interleave, weights_mode, hessian_algo)
        shapes = {"Phi": (f * r, n), "Gamma": (f * r, v * m), "H": (v * m, v * m), "fvec": (v * m, 1)}
        ins = [L.pack(A, interleave), L.pack(B, interleave), L.pack(Cm, interleave),
               None if W3 is None else L.colmajor(W3), None if W4 is None else L.colmajor(W4),
               None if x0 is None else L.pack(np.asarray(x0, dtype=np.float64).reshape(batch, n, 1), interleave),
               None if ref is None else L.pack(np.asarray(ref, dtype=np.float64).reshape(batch, f * r, 1), interleave)]
        outs = {}
        for k, (rows, cols) in shapes.items():
            outs[k] = np.full(max(L.array_elems(batch, interleave, rows * cols), 1), np.nan) if k in want else None
        self._check(self._lib.mpcx_qp_build_f64(self._h, C.byref(d), *[_hp(x) for x in ins],
                                                *[_hp(outs[k]) for k in ("Phi", "Gamma", "H", "fvec")]))
        res = {}
        for k, (rows, cols) in shapes.items():
            if outs[k] is not None:
                res[k] = L.unpack(outs[k], batch, rows, cols, interleave)
        return res

    def qp_solve_dev(self, dims: Dims, H, fvec, u_seq, info, stream=0):
        self._check(self._lib.mpcx_qp_solve_f64_dev(self._h, C.byref(dims), H, fvec, u_seq, info, stream))

    def qp_gain_dev(self, dims: Dims, H, Gamma, W4, gain, info, stream=0):
        self._check(self._lib.mpcx_qp_gain_f64_dev(self._h, C.byref(dims), H, Gamma, W4, gain, info, stream))

    def qp_solve(self, H, fvec, m, interleave=L.INTERLEAVE_AOS):
        """H (batch, vm, vm), fvec (batch, vm[, 1]) -> (u_seq (batch, vm), info (batch,) int32): u_seq = -H^-1 fvec."""
        H = np.asarray(H, dtype=np.float64)
        batch, vm = H.shape[0], H.shape[1]
        d = make_dims(1, m, 1, vm // m, vm // m, batch, 0, interleave)
        hb = L.pack(H, interleave)
        fb = L.pack(np.asarray(fvec, dtype=np.float64).reshape(batch, vm, 1), interleave)
        out = np.full(max(L.array_elems(batch, interleave, vm), 1), np.nan)
        info = np.full(max(batch, 1), -1, dtype=np.int32)
        self._check(self._lib.mpcx_qp_solve_f64(self._h, C.byref(d), _hp(hb), _hp(fb), _hp(out), info.ctypes.data_as(C.c_void_p)))
        return L.unpack(out, batch, vm, 1, interleave)[:, :, 0], info[:batch]

    def qp_gain(self, H, Gamma, Q, m, r, f, v, interleave=L.INTERLEAVE_AOS, weights_mode=WEIGHTS_BLOCK):
        """gain (batch, vm, fr) = H^-1 Gamma^T W4 and info (batch,); Q is the r x r block of W4 (WEIGHTS_BLOCK) or the
        dense (f r) x (f r) W4 itself (WEIGHTS_DENSE)."""
        H = np.asarray(H, dtype=np.float64)
        batch = H.shape[0]
        d = make_dims(1, m, r, f, v, batch, 0, interleave, weights_mode)
        hb, gb, qb = L.pack(H, interleave), L.pack(np.asarray(Gamma, dtype=np.float64), interleave), L.colmajor(Q)
        out = np.full(max(L.array_elems(batch, interleave, v * m * f * r), 1), np.nan)
        info = np.full(max(batch, 1), -1, dtype=np.int32)
        self._check(self._lib.mpcx_qp_gain_f64(self._h, C.byref(d), _hp(hb), _hp(gb), _hp(qb), _hp(out), info.ctypes.data_as(C.c_void_p)))
        return L.unpack(out, batch, v * m, f * r, interleave), info[:batch]

    def qp_gain_rows_dev(self, dims: Dims, H, Gamma, W4, gain_rows, info, stream=0):
        self._check(self._lib.mpcx_qp_gain_rows_f64_dev(self._h, C.byref(dims), H, Gamma, W4, gain_rows, info, stream))

    def qp_gain_rows(self, H, Gamma, Q, m, r, f, v, interleave=L.INTERLEAVE_AOS, weights_mode=WEIGHTS_BLOCK):
        """The first m rows of the gain, (batch, m, fr), and info (batch,): what the control move uses
        (controller.rs:130-132); pass it to closed_loop with v = 1."""
        H = np.asarray(H, dtype=np.float64)
        batch = H.shape[0]
        d = make_dims(1, m, r, f, v, batch, 0, interleave, weights_mode)
        hb, gb, qb = L.pack(H, interleave), L.pack(np.asarray(Gamma, dtype=np.float64), interleave), L.colmajor(Q)
        out = np.full(max(L.array_elems(batch, interleave, m * f * r), 1), np.nan)
        info = np.full(max(batch, 1), -1, dtype=np.int32)
        self._check(self._lib.mpcx_qp_gain_rows_f64(self._h, C.byref(d), _hp(hb), _hp(gb), _hp(qb), _hp(out), info.ctypes.data_as(C.c_void_p)))
        return L.unpack(out, batch, m, f * r, interleave), info[:batch]

    def c2d_dev(self, dims: Dims, Ac, Bc, dt, A, B, info, stream=0):
        self._check(self._lib.mpcx_c2d_backward_euler_f64_dev(self._h, C.byref(dims), Ac, Bc, dt, A, B, info, stream))

    def plants_dev(self, kind, batch, interleave, params, dt, A, B, Cm, info, stream=0):
        self._check(self._lib.mpcx_plants_f64_dev(self._h, kind, batch, interleave, params, dt, A, B, Cm, info, stream))

    def c2d_noalloc_dev(self, dims: Dims, Ac, Bc, dt, A, B, info, stream=0):
        """Same as c2d_dev, through the entry point that allocates no device memory (static device scratch)."""
        self._check(self._lib.mpcx_c2d_backward_euler_f64_noalloc_dev(self._h, C.byref(dims), Ac, Bc, dt, A, B, info, stream))

    def c2d(self, Ac, Bc, dt, interleave=L.INTERLEAVE_AOS):
        """Ac (batch, n, n), Bc (batch, n, m) -> A, B, info: backward Euler, A = (I - Ac dt)^-1, B = A Bc dt."""
        Ac = np.asarray(Ac, dtype=np.float64)
        Bc = np.asarray(Bc, dtype=np.float64)
        batch, n, m = Ac.shape[0], Ac.shape[1], Bc.shape[2]
        d = make_dims(n, m, 1, 1, 1, batch, 0, interleave)
        a = np.full(max(L.array_elems(batch, interleave, n * n), 1), np.nan)
        b = np.full(max(L.array_elems(batch, interleave, n * m), 1), np.nan)
        info = np.full(max(batch, 1), -1, dtype=np.int32)
        self._check(self._lib.mpcx_c2d_backward_euler_f64(self._h, C.byref(d), _hp(L.pack(Ac, interleave)), _hp(L.pack(Bc, interleave)),
                                                          dt, _hp(a), _hp(b), info.ctypes.data_as(C.c_void_p)))
        return L.unpack(a, batch, n, n, interleave), L.unpack(b, batch, n, m, interleave), info[:batch]

    def plants(self, kind, params, dt, interleave=L.INTERLEAVE_AOS):
        """params (batch, 5) for PLANT_DC_MOTOR (b, j, k, l, r) or (batch, 6) for PLANT_TWO_MASS (m1, m2, k1, k2, d1, d2)
        -> A (batch, n, n), B (batch, n, 1), C (batch, 1, n), info."""
        params = np.asarray(params, dtype=np.float64)
        batch, n_par = params.shape
        n = 2 if kind == PLANT_DC_MOTOR else 4
        a = np.full(max(L.array_elems(batch, interleave, n * n), 1), np.nan)
        b = np.full(max(L.array_elems(batch, interleave, n), 1), np.nan)
        c = np.full(max(L.array_elems(batch, interleave, n), 1), np.nan)
        info = np.full(max(batch, 1), -1, dtype=np.int32)
        self._check(self._lib.mpcx_plants_f64(self._h, kind, batch, interleave, _hp(L.pack(params.reshape(batch, n_par, 1), interleave)),
                                              dt, _hp(a), _hp(b), _hp(c), info.ctypes.data_as(C.c_void_p)))
        return (L.unpack(a, batch, n, n, interleave), L.unpack(b, batch, n, 1, interleave),
                L.unpack(c, batch, 1, n, interleave), info[:batch])

    def closed_loop_dev(self, dims: Dims, A, B, Cm, x0, Phi, gain, ref, X, U, Y, stream=0):
        self._check(self._lib.mpcx_closed_loop_f64_dev(self._h, C.byref(dims), A, B, Cm, x0, Phi, gain, ref, X, U, Y, stream))

    def closed_loop(self, A, B, Cm, x0, Phi, gain, ref, T, f, v, interleave=L.INTERLEAVE_AOS, want=("X", "U", "Y")):
        """Receding-horizon loop for a batch.  ref: (r, T+f) shared by the batch, or (batch, r, T+f) per instance.
        Returns X (batch, n, T+1), U (batch, m, T), Y (batch, r, T)."""
        A = np.asarray(A, dtype=np.float64)
        B = np.asarray(B, dtype=np.float64)
        Cm = np.asarray(Cm, dtype=np.float64)
        ref = np.asarray(ref, dtype=np.float64)
        batch, n, m, r = A.shape[0], A.shape[1], B.shape[2], Cm.shape[1]
        shared = ref.ndim == 2
        assert ref.shape[-2:] == (r, T + f), "ref must be r x (T + f)"
        d = make_dims(n, m, r, f, v, batch, T, interleave, u_shared=1 if shared else 0)
        ins = [L.pack(A, interleave), L.pack(B, interleave), L.pack(Cm, interleave),
               L.pack(np.asarray(x0, dtype=np.float64).reshape(batch, n, 1), interleave),
               L.pack(np.asarray(Phi, dtype=np.float64), interleave), L.pack(np.asarray(gain, dtype=np.float64), interleave),
               L.colmajor(ref) if shared else L.pack(ref, interleave)]
        shapes = {"X": (n, T + 1), "U": (m, T), "Y": (r, T)}
        outs = {k: (np.full(max(L.array_elems(batch, interleave, rr * cc), 1), np.nan) if k in want else None)
                for k, (rr, cc) in shapes.items()}
        self._check(self._lib.mpcx_closed_loop_f64(self._h, C.byref(d), *[_hp(a) for a in ins],
                                                   _hp(outs["X"]), _hp(outs["U"]), _hp(outs["Y"])))
        return {k: L.unpack(outs[k], batch, *shapes[k], interleave) for k in want}

    def rollout(self, A, B, Cm, x0, U, interleave=L.INTERLEAVE_AOS, want=("X", "Y")):
        """U is (batch, m, T) or, shared by all instances, (m, T).  Returns X (batch, n, T+1), Y (batch, r, T)."""
        A = np.asarray(A, dtype=np.float64)
        B = np.asarray(B, dtype=np.float64)
        Cm = np.asarray(Cm, dtype=np.float64)
        U = np.asarray(U, dtype=np.float64)
        batch, n = A.shape[0], A.shape[1]
        m, r = B.shape[2], Cm.shape[1]
        shared = U.ndim == 2
        T = U.shape[-1]
        d = make_dims(n, m, r, 1, 1, batch, T, interleave, u_shared=int(shared))
        u_buf = L.colmajor(U) if shared else L.pack(U, interleave)
        if u_buf.size == 0:
            u_buf = np.zeros(1)
        ins = [L.pack(A, interleave), L.pack(B, interleave), L.pack(Cm, interleave),
               L.pack(np.asarray(x0, dtype=np.float64).reshape(batch, n, 1), interleave), u_buf]
        X = np.full(max(L.array_elems(batch, interleave, n * (T + 1)), 1), np.nan) if "X" in want else None
        Y = np.full(max(L.array_elems(batch, interleave, r * T), 1), np.nan) if "Y" in want else None
        self._check(self._lib.mpcx_rollout_f64(self._h, C.byref(d), *[_hp(x) for x in ins], _hp(X), _hp(Y)))
        res = {}
        if X is not None:
            res["X"] = L.unpack(X, batch, n, T + 1, interleave)
        if Y is not None:
            res["Y"] = L.unpack(Y, batch, r, T, interleave)
        return res
