"""Rollout of the quadrotor batch with chosen T and outputs (diagnostic, GPU only): python tools/probe_rollout2.py T X|Y|XY [batch]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from mpc_rs_amd import ffi, layout as L, synth
dev = torch.device("cuda", 0); ctx = ffi.Context(0)
T = int(sys.argv[1]); want = sys.argv[2]; batch = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
d = synth.make_batch("quadrotor", batch)
n, m, r = 12, 4, 12
to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
A, B, C = (to(L.pack(d[k], 1)) for k in ("A", "B", "C"))
x0 = to(L.pack(d["x0"].reshape(batch, n, 1), 1)); U = to(np.ones((T, m)).reshape(-1))
X = torch.empty(batch * n * (T + 1) + 64, dtype=torch.float64, device=dev); Y = torch.empty(batch * r * T + 64, dtype=torch.float64, device=dev)
dims = ffi.make_dims(n, m, r, 1, 1, batch, T, 1, u_shared=1)
xp = X.data_ptr() if "X" in want else None; yp = Y.data_ptr() if "Y" in want else None
step = lambda: ctx.rollout_dev(dims, A.data_ptr(), B.data_ptr(), C.data_ptr(), x0.data_ptr(), U.data_ptr(), xp, yp, 0)
for _ in range(3): step()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): step()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
wb = 8 * ((n * (T + 1) if xp else 0) + (r * T if yp else 0)) * batch
print(f"T {T} {want} batch {batch}: {ms:.4f} ms  {wb/ms/1e6:.0f} GB/s written", flush=True)
