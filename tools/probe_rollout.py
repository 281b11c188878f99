import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from mpc_rs_amd import ffi, layout as L, synth
dev = torch.device("cuda", 0); ctx = ffi.Context(0)
for batch, T in ((2048, 200), (4096, 200), (8192, 200), (16384, 200), (16384, 400), (32768, 200)):
    d = synth.make_batch("quadrotor", batch)
    n, m, r = 12, 4, 12
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    A, B, C = (to(L.pack(d[k], 1)) for k in ("A", "B", "C"))
    x0 = to(L.pack(d["x0"].reshape(batch, n, 1), 1)); U = to(np.ones((T, m)).reshape(-1))
    X = torch.empty(batch * n * (T + 1), dtype=torch.float64, device=dev); Y = torch.empty(batch * r * T, dtype=torch.float64, device=dev)
    dims = ffi.make_dims(n, m, r, 1, 1, batch, T, 1, u_shared=1)
    step = lambda: ctx.rollout_dev(dims, A.data_ptr(), B.data_ptr(), C.data_ptr(), x0.data_ptr(), U.data_ptr(), X.data_ptr(), Y.data_ptr(), 0)
    for _ in range(3): step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): step()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"batch {batch:6d} T {T:4d}: {ms:.4f} ms  {8*(n*(T+1)+r*T)*batch/ms/1e6:.0f} GB/s written", flush=True)
