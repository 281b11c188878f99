"""The C ABI called from plain C (tests/c_abi/abi_check.c, gcc, no HIP headers): the reference scenario of
src/main.rs:14-71 through host pointers only.  On a machine without a GPU the same binary must report
MPCX_ERR_NO_DEVICE from mpcx_create instead of computing anything."""
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "c_abi", "abi_check")


def _run():
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", os.path.join(HERE, "c_abi")], check=True, capture_output=True)
    return subprocess.run([EXE], capture_output=True, text=True, timeout=120)


def test_plain_c_caller_without_gpu_gets_no_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    p = _run()
    assert p.returncode == 3 and "create rc=-5" in p.stdout, (p.returncode, p.stdout, p.stderr)


@pytest.mark.gpu
def test_plain_c_caller_reproduces_the_reference_scenario():
    from oracle import oracle_ffi as orc
    p = _run()
    assert p.returncode == 0, (p.stdout, p.stderr)
    vals = {}
    for line in p.stdout.splitlines():
        toks = line.split()
        if toks[0] == "A":
            vals["A"] = [float(t) for t in toks[1:]]
        elif toks[0] in ("H00", "gain00"):
            vals[toks[0]] = float(toks[1])
        elif toks[0] in ("y50", "u0"):
            for k, x in zip(toks[0::2], toks[1::2]):
                vals[k] = float(x)
        elif toks[0] == "bad_dims":
            vals["bad_dims"] = int(toks[1].split("=")[1])
    assert vals["bad_dims"] == -2
    f = v = 20
    Ac, Bc, Cc = orc.dc_motor_continuous()
    _, Ad, Bd = orc.c2d_backward_euler(Ac, Bc, 0.05)
    _, _, w3, w4 = orc.controller_weights(1, 1, f, v, 1.1e-9, 1e-4, 10.0)
    _, O, M, H, G = orc.form_lifted_matrices(Ad, Bd, Cc, f, v, w3, w4)
    traj = np.array([1.0 if (i < 100 or i >= 200) else 0.0 for i in range(300)])
    S, Out, In = orc.closed_loop(Ad, Bd, Cc, f, v, O, G, traj, 280, np.zeros(2))
    np.testing.assert_allclose(vals["A"], Ad.T.reshape(-1), rtol=1e-12)           # column-major print order
    np.testing.assert_allclose(vals["H00"], H[0, 0], rtol=1e-10)
    np.testing.assert_allclose(vals["gain00"], G[0, 0], rtol=1e-8)                # through H^-1, cond(H) ~ 268
    y = Out[0, 1:]                                                                # outputs[t+1] = y_t (main.rs:63)
    np.testing.assert_allclose([vals["y50"], vals["y150"], vals["y279"]], [y[50], y[150], y[279]], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose([vals["u0"], vals["u279"]], [In[0], In[279]], rtol=1e-8, atol=1e-10)
