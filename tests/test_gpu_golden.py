"""The committed fixtures tests/golden/qp_*.npz (inputs and expected Phi, Gamma, H, fvec, X, Y of all four workloads,
including the held-input branch f > v of controller.rs:63-88; written by tools/gen_golden.py from the numpy
restatement) against what the C ABI returns on the GPU.  Tolerance: 1e-10 relative Frobenius, per output and for the
worst instance."""
import glob
import os

import numpy as np
import pytest

from helpers import assert_close

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PATHS = sorted(glob.glob(os.path.join(GOLDEN, "qp_*.npz")))


def test_fixtures_are_present():
    names = {os.path.basename(p) for p in PATHS}
    assert {"qp_msd.npz", "qp_two_mass.npz", "qp_quadrotor.npz", "qp_long.npz", "qp_two_mass_held.npz",
            "qp_quadrotor_held.npz"} <= names


@pytest.mark.parametrize("interleave", [1, 0])
@pytest.mark.parametrize("path", PATHS, ids=[os.path.basename(p)[:-4] for p in PATHS])
def test_gpu_matches_golden_vectors(ctx, path, interleave):
    g = np.load(path, allow_pickle=False)
    f, v = int(g["f"]), int(g["v"])
    got = ctx.qp_build(g["A"], g["B"], g["C"], f, v, W3=g["W3"], W4=g["Q"], x0=g["x0"], ref=g["ref"], interleave=interleave)
    for k in ("Phi", "Gamma", "H", "fvec"):
        assert_close(got[k].reshape(g[k].shape), g[k], f"{os.path.basename(path)} W={interleave} {k}")
    ro = ctx.rollout(g["A"], g["B"], g["C"], g["x0"], g["U"], interleave=interleave)
    assert_close(ro["X"], g["X"], f"{os.path.basename(path)} W={interleave} X")
    assert_close(ro["Y"], g["Y"], f"{os.path.basename(path)} W={interleave} Y")
