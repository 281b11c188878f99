"""Generates tests/golden/qp_*.npz and rollout_*.npz: small batches of every workload with the outputs of the numpy
restatement (oracle/np_oracle.py).  The vectors are NOT outputs of the reference (it cannot be built here); they pin
the two independent restatements, the C oracle and the GPU kernels to one another across machines and compilers.

    python tools/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpc_rs_amd import synth  # noqa: E402
from oracle import np_oracle as npo  # noqa: E402

CASES = [  # name, workload, batch, f, v
    ("two_mass", "two_mass", 6, 20, 20),
    ("msd", "msd", 6, 20, 20),
    ("two_mass_held", "two_mass", 4, 12, 5),
    ("quadrotor", "quadrotor", 2, 8, 8),
    ("quadrotor_held", "quadrotor", 2, 7, 3),
    ("long", "long", 2, 30, 30),
]


def main():
    dst = os.path.join(ROOT, "tests", "golden")
    for name, wl, batch, f, v in CASES:
        d = synth.make_batch(wl, batch, f=f, v=v)
        w4 = synth.dense_w4(d["Q"], f)
        Phi, Gamma, H, fv = [], [], [], []
        for b in range(batch):
            O, M = npo.lifted_matrices(d["A"][b], d["B"][b], d["C"][b], f, v)
            Phi.append(O)
            Gamma.append(M)
            H.append(npo.hessian(M, d["W3"], w4))
            fv.append(npo.gradient(O, M, w4, d["x0"][b], d["ref"][b]).reshape(-1, 1))
        T = 25
        U = np.random.default_rng(7).standard_normal((batch, d["m"], T))
        X, Y = [], []
        for b in range(batch):
            y, x = npo.system_simulate(d["A"][b], d["B"][b], d["C"][b], U[b], d["x0"][b])
            X.append(x)
            Y.append(y)
        np.savez_compressed(os.path.join(dst, f"qp_{name}.npz"), A=d["A"], B=d["B"], C=d["C"], x0=d["x0"], ref=d["ref"],
                            W3=d["W3"], Q=d["Q"], f=f, v=v, Phi=np.stack(Phi), Gamma=np.stack(Gamma), H=np.stack(H),
                            fvec=np.stack(fv), U=U, X=np.stack(X), Y=np.stack(Y))
        print(name, os.path.getsize(os.path.join(dst, f"qp_{name}.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
