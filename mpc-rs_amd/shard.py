"""Batch sharding by instance index and the fixed-size statistics record that ranks exchange.

Instances are independent, so a batch of B plants is cut into contiguous index ranges, one per rank (one process
per GPU).  There is no collective on the data path; after a run each rank reduces its outputs to one
``mpcx_stats`` record on its GPU (mpcx_stats_f64_dev) and the ranks all-gather those 48-byte records.
"""
from __future__ import annotations

import numpy as np


def shard_range(total: int, world: int, rank: int) -> tuple[int, int]:
    """(start, count) of rank's contiguous share; the first total % world ranks take one extra instance."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("need world >= 1 and 0 <= rank < world")
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def stats_from_words(words: np.ndarray) -> dict:
    """Decode the six 64-bit words of an mpcx_stats record (as read back from device memory)."""
    w = np.ascontiguousarray(np.asarray(words, dtype=np.int64).reshape(6))
    fl = w.view(np.float64)
    return {"count": int(w[0]), "non_finite": int(w[1]), "sum": float(fl[2]), "sum_sq": float(fl[3]),
            "max_abs": float(fl[4]), "checksum": int(w.view(np.uint64)[5])}


def merge_stats(records: list[dict]) -> dict:
    """Combine per-shard records into the record of the whole batch (checksum wraps modulo 2^64)."""
    out = {"count": 0, "non_finite": 0, "sum": 0.0, "sum_sq": 0.0, "max_abs": 0.0, "checksum": 0}
    for r in records:
        out["count"] += r["count"]
        out["non_finite"] += r["non_finite"]
        out["sum"] += r["sum"]
        out["sum_sq"] += r["sum_sq"]
        out["max_abs"] = max(out["max_abs"], r["max_abs"])
        out["checksum"] = (out["checksum"] + r["checksum"]) & 0xFFFFFFFFFFFFFFFF
    return out


def host_stats(values: np.ndarray) -> dict:
    """The same record computed on the host (for tests of the device reduction)."""
    v = np.ascontiguousarray(np.asarray(values, dtype=np.float64).reshape(-1))
    fin = np.isfinite(v)
    x = v.view(np.uint64).copy()
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
        chk = int(np.sum(x, dtype=np.uint64))
    vf = v[fin]
    return {"count": int(v.size), "non_finite": int((~fin).sum()), "sum": float(vf.sum()), "sum_sq": float((vf * vf).sum()),
            "max_abs": float(np.abs(vf).max()) if vf.size else 0.0, "checksum": chk}
