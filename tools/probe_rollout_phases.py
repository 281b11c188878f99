import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from mpc_rs_amd import ffi, layout as L, synth
dev = torch.device("cuda", 0); ctx = ffi.Context(0)
for batch in (2048, 16384):
    T = 200
    d = synth.make_batch("quadrotor", batch)
    n, m, r = 12, 4, 12
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    A, B, C = (to(L.pack(d[k], 1)) for k in ("A", "B", "C"))
    x0 = to(L.pack(d["x0"].reshape(batch, n, 1), 1)); U = to(np.ones((T, m)).reshape(-1))
    X = torch.empty(batch * n * (T + 1), dtype=torch.float64, device=dev); Y = torch.empty(batch * r * T, dtype=torch.float64, device=dev)
    dims = ffi.make_dims(n, m, r, 1, 1, batch, T, 1, u_shared=1)
    for _ in range(3):
        ctx.rollout_dev(dims, A.data_ptr(), B.data_ptr(), C.data_ptr(), x0.data_ptr(), U.data_ptr(), X.data_ptr(), Y.data_ptr(), 0)
    torch.cuda.synchronize()
    y = Y.cpu().numpy()[: (batch // 4) * 8].reshape(-1, 8)[:, :4]
    print(batch, "cycles per chunk: stage-in %.0f  Bu %.0f  8 steps %.0f  flush %.0f" % tuple(y.mean(axis=0)))
