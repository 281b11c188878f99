"""Runs a few launches of one QP-build configuration (for rocprofv3 counter passes).  GPU only.
usage: probe_one.py workload count interleave outputs(e.g. H or Phi,Gamma,H,fvec) [launches]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_rs_amd import ffi, layout as L, synth
name, count, il, subset = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4].split(",")
launches = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = torch.device("cuda", 0); ctx = ffi.Context(0)
wl = synth.WORKLOADS[name]; n, m, r, f, v = wl["n"], wl["m"], wl["r"], wl["f"], wl["v"]
d = synth.make_batch(name, count)
to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bufs = {k: to(L.pack(d[k], il)) for k in ("A", "B", "C")}
bufs["x0"] = to(L.pack(d["x0"].reshape(count, n, 1), il)); bufs["ref"] = to(L.pack(d["ref"].reshape(count, f * r, 1), il))
bufs["W3"] = to(L.colmajor(d["W3"])); bufs["Q"] = to(L.colmajor(d["Q"]))
sizes = {"Phi": f * r * n, "Gamma": f * r * v * m, "H": v * m * v * m, "fvec": v * m}
outs = {k: torch.empty(L.array_elems(count, il, e), dtype=torch.float64, device=dev) for k, e in sizes.items()}
dims = ffi.make_dims(n, m, r, f, v, count, 0, il)
ptr = lambda k: outs[k].data_ptr() if k in subset else None
for _ in range(launches):
    ctx.qp_build_dev(dims, bufs["A"].data_ptr(), bufs["B"].data_ptr(), bufs["C"].data_ptr(), bufs["W3"].data_ptr(),
                     bufs["Q"].data_ptr(), bufs["x0"].data_ptr(), bufs["ref"].data_ptr(), ptr("Phi"), ptr("Gamma"), ptr("H"), ptr("fvec"), 0)
torch.cuda.synchronize()
print("done")
