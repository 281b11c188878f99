"""Store rate of the QP build per subset of its outputs (diagnostic, GPU only):
    python tools/probe_outputs.py <workload> <batch> [f v]
Times mpcx_qp_build_f64_dev with each listed subset of {Phi, Gamma, H, fvec} requested and prints the time and the
bytes of the requested outputs per second: which output's store stream is the slow one, and what the streams cost
each other when they share a launch."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_rs_amd import ffi, layout as L, synth
name = sys.argv[1]; count = int(sys.argv[2])
dev = torch.device("cuda", 0); ctx = ffi.Context(0)
wl = synth.WORKLOADS[name]; n, m, r, f, v = wl["n"], wl["m"], wl["r"], wl["f"], wl["v"]
if len(sys.argv) > 4:
    f, v = int(sys.argv[3]), int(sys.argv[4])
d = synth.make_batch(name, count, f=f, v=v); il = 1
to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bufs = {k: to(L.pack(d[k], il)) for k in ("A", "B", "C")}
bufs["x0"] = to(L.pack(d["x0"].reshape(count, n, 1), il)); bufs["ref"] = to(L.pack(d["ref"].reshape(count, f * r, 1), il))
bufs["W3"] = to(L.colmajor(d["W3"])); bufs["Q"] = to(L.colmajor(d["Q"]))
sizes = {"Phi": f * r * n, "Gamma": f * r * v * m, "H": v * m * v * m, "fvec": v * m}
outs = {k: torch.zeros(count * e, dtype=torch.float64, device=dev) for k, e in sizes.items()}
dims = ffi.make_dims(n, m, r, f, v, count, 0, il)
subsets = os.environ.get("SUBSETS", "Gamma,H,Gamma+H,Phi+Gamma+H+fvec").split(",")
for sub in subsets:
    want = sub.split("+")
    ptr = lambda k: outs[k].data_ptr() if k in want else None
    run = lambda: ctx.qp_build_dev(dims, bufs["A"].data_ptr(), bufs["B"].data_ptr(), bufs["C"].data_ptr(), bufs["W3"].data_ptr(),
                                   bufs["Q"].data_ptr(), bufs["x0"].data_ptr(), bufs["ref"].data_ptr(), ptr("Phi"), ptr("Gamma"), ptr("H"), ptr("fvec"), 0)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    steps = 5
    e0.record()
    for _ in range(steps):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    nbytes = 8 * count * sum(sizes[k] for k in want)
    print(f"{name} f={f} v={v} batch={count} {sub:>20}: {ms:7.3f} ms  {nbytes / ms / 1e9:6.2f} TB/s")
