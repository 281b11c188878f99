"""Shared helpers of the test-suite: tolerances, norms, random stable plants, oracle batch calls."""
import numpy as np

# f64 tolerance of the parity tests: relative error in Frobenius norm, per output, over the whole batch
# and per instance (the bar the north-star states: <= 1e-10).
RTOL = 1e-10


def rel_fro(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    den = np.linalg.norm(want.ravel())
    num = np.linalg.norm((got - want).ravel())
    return num / den if den > 0 else num


def worst_instance_rel(got, want):
    got, want = np.asarray(got), np.asarray(want)
    b = got.shape[0]
    num = np.linalg.norm((got - want).reshape(b, -1), axis=1)
    den = np.linalg.norm(want.reshape(b, -1), axis=1)
    den = np.where(den > 0, den, 1.0)
    return float(np.max(num / den)) if b else 0.0


def assert_close(got, want, what, tol=RTOL):
    assert np.all(np.isfinite(got)), f"{what}: non-finite values"
    e = rel_fro(got, want)
    w = worst_instance_rel(got, want)
    assert e <= tol and w <= tol, f"{what}: rel Frobenius error {e:.3e} (worst instance {w:.3e}) > {tol:.1e}"


def random_stable_plants(rng, batch, n, m, r, rho=0.9):
    """Random discrete plants with spectral radius `rho` (so powers stay bounded over long horizons)."""
    A = rng.standard_normal((batch, n, n))
    for b in range(batch):
        A[b] *= rho / max(np.max(np.abs(np.linalg.eigvals(A[b]))), 1e-12)
    B = rng.standard_normal((batch, n, m))
    C = rng.standard_normal((batch, r, n))
    return A, B, C


def oracle_qp(orc, data, want=("Phi", "Gamma", "H", "fvec"), threads=4):
    from mpc_rs_amd import synth
    w4 = synth.dense_w4(data["Q"], data["f"])
    out, _ = orc.qp_build_batch(data["A"], data["B"], data["C"], data["W3"], w4, data["x0"], data["ref"], data["f"],
                                data["v"], threads=threads, want=want)
    return out
