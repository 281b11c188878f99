"""Phases of the closed-loop kernel that takes the window products out of the loop (needs a library built with
-DMPCX_LOOP_STAMPS: tools/build_variant.sh loopstamps closed_loop.hip -DMPCX_LOOP_STAMPS, MPCX_LIB pointing at it;
diagnostic, GPU only; the first entries of X are overwritten by the stamps):
    python tools/probe_loop.py <n> <m> <r> <f> <T> <batch>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_rs_amd import ffi
n, m, r, f, T, batch = (int(x) for x in sys.argv[1:7])
rng = np.random.default_rng(1)
A = rng.standard_normal((batch, n, n)) * (0.5 / np.sqrt(n)); B = rng.standard_normal((batch, n, m)); C = rng.standard_normal((batch, r, n))
Phi = rng.standard_normal((batch, f * r, n)) * 0.1
rows = rng.standard_normal((batch, m, f * r)) * (0.1 / np.sqrt(f * r))
ref = rng.uniform(0, 1, (r, T + f))
ctx = ffi.Context(0)
for _ in range(2):
    out = ctx.closed_loop(A, B, C, np.zeros((batch, n)), Phi, rows, ref, T, f, 1, interleave=1)
st = out["X"][:, :6, 0].mean(axis=0) / 100.0  # device elements 0..5 of X: x_0[0..5]
names = ["K Phi", "reference columns", "window products", "their sum + requests", "recursion", "flush"]
print("us per instance: " + "  ".join(f"{nm} {x:.1f}" for nm, x in zip(names, st)) + f"   total {st.sum():.1f}")
