"""Where a panel step of the blocked solver's right-hand-side kernel spends its time (needs a library built with
-DMPCX_RHS_STAMPS: tools/build_variant.sh rhsstamps solve_large.hip -DMPCX_RHS_STAMPS, MPCX_LIB pointing at it;
diagnostic, GPU only; the gain it computes is overwritten by the stamps):
    python tools/probe_rhs.py <workload> <count>"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_rs_amd import ffi, layout as L, synth
name = sys.argv[1]; count = int(sys.argv[2])
dev = torch.device("cuda", 0); ctx = ffi.Context(0)
d = synth.make_batch(name, count); n, m, r, f, v = d["n"], d["m"], d["r"], d["f"], d["v"]
FR, VM = f * r, v * m
to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bufs = {k: to(L.pack(d[k], 1)) for k in ("A", "B", "C")}
bufs["x0"] = to(L.pack(d["x0"].reshape(count, n, 1), 1)); bufs["ref"] = to(L.pack(d["ref"].reshape(count, FR, 1), 1))
bufs["W3"] = to(L.colmajor(d["W3"])); bufs["Q"] = to(L.colmajor(d["Q"]))
out = lambda e: torch.zeros(count * e, dtype=torch.float64, device=dev)
Gam, H, fv, gain = out(FR * VM), out(VM * VM), out(VM), out(VM * FR)
info = torch.zeros(count, dtype=torch.int32, device=dev)
dims = ffi.make_dims(n, m, r, f, v, count, 0, 1)
ctx.qp_build_dev(dims, bufs["A"].data_ptr(), bufs["B"].data_ptr(), bufs["C"].data_ptr(), bufs["W3"].data_ptr(), bufs["Q"].data_ptr(),
                 bufs["x0"].data_ptr(), bufs["ref"].data_ptr(), None, Gam.data_ptr(), H.data_ptr(), fv.data_ptr(), 0)
for _ in range(2):
    ctx.qp_gain_dev(dims, H.data_ptr(), Gam.data_ptr(), bufs["Q"].data_ptr(), gain.data_ptr(), info.data_ptr(), 0)
torch.cuda.synchronize()
npan = (VM + 15) // 16
nw = 8 if npan > 16 else 4
raw = gain.cpu().numpy().reshape(count, VM * FR)[:, : 8 * nw].reshape(count, nw, 8).mean(axis=0) / 100.0  # us
st = raw[:, :4] / npan  # per panel step
names = ["wait at barrier 1", "section + requests", "wait at barrier 2", "tile updates"]
for w in range(nw):
    print(f"wavefront {w}: " + "  ".join(f"{nm} {x:5.2f}" for nm, x in zip(names, st[w])) + f"   step {st[w].sum():5.2f} us")
print("whole block (wavefront 0): forming the block %.1f us, %d panel steps %.1f us, U11 step and result %.1f us" % (raw[0, 4], npan, raw[0, 5], raw[0, 6]))
