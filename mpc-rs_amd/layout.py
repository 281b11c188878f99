"""Batch layouts of the C ABI (include/mpcx.h).

An array holds, for each of ``batch`` instances, one matrix of ``rows x cols`` f64 stored column-major (nalgebra's
``DMatrix::as_slice`` order): element index ``e = col * rows + row``.  With interleave width ``W`` the element
``e`` of instance ``b`` lives at ``(b // W) * (E * W) + e * W + (b % W)``.  ``W = 1`` is instance-major, ``W = 0``
stands for ``W = batch`` (fully batch-interleaved); other widths tile the batch and pad the last tile.
"""
from __future__ import annotations

import numpy as np

INTERLEAVE_FULL = 0
INTERLEAVE_AOS = 1


def resolve_width(batch: int, interleave: int) -> int:
    return max(batch, 1) if interleave == 0 else interleave


def array_elems(batch: int, interleave: int, elems_per_instance: int) -> int:
    if batch <= 0 or elems_per_instance <= 0:
        return 0
    w = resolve_width(batch, interleave)
    return -(-batch // w) * w * elems_per_instance


def pack(mats: np.ndarray, interleave: int) -> np.ndarray:
    """(batch, rows, cols) array of ordinary row/col matrices -> flat f64 buffer in the ABI layout."""
    mats = np.asarray(mats, dtype=np.float64)
    if mats.ndim == 2:  # (batch, rows): vectors
        mats = mats[:, :, None]
    batch, rows, cols = mats.shape
    E = rows * cols
    if batch == 0 or E == 0:
        return np.zeros(0)
    cm = np.ascontiguousarray(mats.transpose(0, 2, 1)).reshape(batch, E)  # column-major per instance
    w = resolve_width(batch, interleave)
    tiles = -(-batch // w)
    if tiles * w != batch:
        cm = np.concatenate([cm, np.zeros((tiles * w - batch, E))], axis=0)
    return np.ascontiguousarray(cm.reshape(tiles, w, E).transpose(0, 2, 1)).reshape(-1)


def unpack(flat: np.ndarray, batch: int, rows: int, cols: int, interleave: int) -> np.ndarray:
    """Inverse of :func:`pack`: flat ABI buffer -> (batch, rows, cols)."""
    E = rows * cols
    if batch == 0 or E == 0:
        return np.zeros((batch, rows, cols))
    w = resolve_width(batch, interleave)
    tiles = -(-batch // w)
    cm = np.asarray(flat, dtype=np.float64)[: tiles * w * E].reshape(tiles, E, w).transpose(0, 2, 1).reshape(tiles * w, E)
    return np.ascontiguousarray(cm[:batch].reshape(batch, cols, rows).transpose(0, 2, 1))


def colmajor(mat: np.ndarray) -> np.ndarray:
    """One shared (rows, cols) matrix -> flat column-major buffer."""
    return np.ascontiguousarray(np.asarray(mat, dtype=np.float64).T).reshape(-1)
