"""Times the QP build with subsets of the outputs requested, per interleave width (diagnostic, GPU only)."""
import os, sys, itertools
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_rs_amd import ffi, layout as L, synth

name = sys.argv[1] if len(sys.argv) > 1 else "two_mass"
count = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
dev = torch.device("cuda", 0)
ctx = ffi.Context(0)
wl = synth.WORKLOADS[name]
n, m, r, f, v = wl["n"], wl["m"], wl["r"], wl["f"], wl["v"]
if len(sys.argv) > 5:  # optional horizon override: f v
    f, v = int(sys.argv[4]), int(sys.argv[5])
d = synth.make_batch(name, count, f=f, v=v)
sizes = {"Phi": f * r * n, "Gamma": f * r * v * m, "H": v * m * v * m, "fvec": v * m}
for il in [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["0", "4096"])]:
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    bufs = {k: to(L.pack(d[k], il)) for k in ("A", "B", "C")}
    bufs["x0"] = to(L.pack(d["x0"].reshape(count, n, 1), il)); bufs["ref"] = to(L.pack(d["ref"].reshape(count, f * r, 1), il))
    bufs["W3"] = to(L.colmajor(d["W3"])); bufs["Q"] = to(L.colmajor(d["Q"]))
    outs = {k: torch.empty(L.array_elems(count, il, e), dtype=torch.float64, device=dev) for k, e in sizes.items()}
    dims = ffi.make_dims(n, m, r, f, v, count, 0, il)
    for subset in [("Phi",), ("Gamma",), ("H",), ("fvec",), ("Gamma", "H"), ("Phi", "Gamma", "H"), ("Phi", "Gamma", "H", "fvec")]:
        ptr = lambda k: outs[k].data_ptr() if k in subset else None
        def step():
            ctx.qp_build_dev(dims, bufs["A"].data_ptr(), bufs["B"].data_ptr(), bufs["C"].data_ptr(), bufs["W3"].data_ptr(),
                             bufs["Q"].data_ptr(), bufs["x0"].data_ptr(), bufs["ref"].data_ptr(), ptr("Phi"), ptr("Gamma"),
                             ptr("H"), ptr("fvec"), 0)
        for _ in range(3): step()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): step()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        wbytes = 8 * sum(sizes[k] for k in subset) * count
        print(f"W={il:6d} {'+'.join(subset):22s} {ms:7.3f} ms  {wbytes / ms / 1e6:8.1f} GB/s written", flush=True)
