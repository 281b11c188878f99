"""Times the QP build for arbitrary (n, m, r, f, v) on random stable plants (device-resident, HIP events).  GPU only.
usage: probe_shapes.py batch interleave n,m,r,f,v [n,m,r,f,v ...]
Prints per shape: ms per launch, builds/s, fraction of 8 TB/s on the algorithmic bytes."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_rs_amd import ffi, layout as L, synth  # noqa: E402

batch, il = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
ctx = ffi.Context(0)
rng = np.random.default_rng(5)


def to(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


for spec in sys.argv[3:]:
    n, m, r, f, v = (int(x) for x in spec.split(","))
    A = rng.standard_normal((batch, n, n)) * (0.5 / np.sqrt(n))
    B = rng.standard_normal((batch, n, m))
    C = rng.standard_normal((batch, r, n))
    w3, q = synth.block_weights(m, r, v, 0.3, 0.7, 2.0, all_elements=False)
    bufs = {"A": to(L.pack(A, il)), "B": to(L.pack(B, il)), "C": to(L.pack(C, il)),
            "x0": to(L.pack(rng.standard_normal((batch, n, 1)), il)), "ref": to(L.pack(rng.standard_normal((batch, f * r, 1)), il)),
            "W3": to(L.colmajor(w3)), "Q": to(L.colmajor(q))}
    sizes = {"Phi": f * r * n, "Gamma": f * r * v * m, "H": v * m * v * m, "fvec": v * m}
    outs = {k: torch.empty(L.array_elems(batch, il, e), dtype=torch.float64, device=dev) for k, e in sizes.items()}
    dims = ffi.make_dims(n, m, r, f, v, batch, 0, il)
    sp = torch.cuda.current_stream().cuda_stream

    def step():
        ctx.qp_build_dev(dims, bufs["A"].data_ptr(), bufs["B"].data_ptr(), bufs["C"].data_ptr(), bufs["W3"].data_ptr(),
                         bufs["Q"].data_ptr(), bufs["x0"].data_ptr(), bufs["ref"].data_ptr(), outs["Phi"].data_ptr(),
                         outs["Gamma"].data_ptr(), outs["H"].data_ptr(), outs["fvec"].data_ptr(), sp)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record()
        step()
        b.record()
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    nbytes = 8 * (n * n + n * m + r * n + n + f * r + f * r * n + f * r * v * m + (v * m) ** 2 + v * m)
    print(f"n={n} m={m} r={r} f={f} v={v} batch={batch} W={il}: {ms:.4f} ms  {batch / ms * 1e3:.3e} builds/s  "
          f"{nbytes * batch / ms / 1e6 / 8000:.3f} of HBM peak", flush=True)
