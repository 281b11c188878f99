"""Time of the general-n discretisation (mpcx_c2d_backward_euler_f64: I - Ac dt inverted by the batched solver) for a few
state counts (diagnostic, GPU only; MPCX_LIB selects a library variant):  python tools/ab_c2d.py [n ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_rs_amd import ffi
ctx = ffi.Context(0); dev = torch.device("cuda", 0)
for n in [int(x) for x in sys.argv[1:]] or [24, 48, 64, 80, 128]:
    batch, m = 4096, 2
    rng = np.random.default_rng(n)
    Ac = torch.from_numpy(rng.standard_normal((batch, n, n)) * 0.3).to(dev); Bc = torch.from_numpy(rng.standard_normal((batch, m, n))).to(dev)
    A = torch.empty_like(Ac); B = torch.empty_like(Bc); info = torch.zeros(batch, dtype=torch.int32, device=dev)
    d = ffi.make_dims(n, m, 1, 1, 1, batch, 0, 1)
    run = lambda: ctx.c2d_dev(d, Ac.data_ptr(), Bc.data_ptr(), 0.01, A.data_ptr(), B.data_ptr(), info.data_ptr(), 0)
    for _ in range(2): run()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): run()
    torch.cuda.synchronize()
    print(f"c2d n={n} batch={batch}: {(time.perf_counter() - t) / 5 * 1e3:.3f} ms, failed {int(info.sum())}")
