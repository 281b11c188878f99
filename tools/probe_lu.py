"""Per-block time stamps of the left-looking factorisation (needs a library built with -DMPCX_LU_STAMPS:
tools/build_variant.sh lustamps solve_large.hip -DMPCX_LU_STAMPS, MPCX_LIB pointing at it; diagnostic, GPU only)."""
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
if npan * 16 <= 256:  # right-looking kernel: three sums per instance
    st = gain.cpu().numpy().reshape(count, VM * FR)[:, :3] / 100.0
    print("per instance (us): panel loads %.1f  panel factorisations %.1f  trailing sweeps %.1f  total %.1f; per panel %.2f / %.2f / %.2f" %
          (*st.mean(axis=0), st.sum(axis=1).mean(), *(st.mean(axis=0) / npan)))
    sys.exit(0)
st = gain.cpu().numpy().reshape(count, VM * FR)[:, : 4 * npan].reshape(count, npan, 4) / 100.0  # us
load, apply_, fact = st[:, :, 1] - st[:, :, 0], st[:, :, 2] - st[:, :, 1], st[:, :, 3] - st[:, :, 2]
print("per instance: load %.0f us, apply %.0f us, factor+store %.0f us, total %.0f us" %
      (load.sum(1).mean(), apply_.sum(1).mean(), fact.sum(1).mean(), (st[:, -1, 3] - st[:, 0, 0]).mean()))
for j in (1, 2, 5, 10, 20, 30, 40, npan - 1):
    if j < npan:
        print("block %3d: load %6.2f  apply %7.2f (%.2f us per panel)  factor+store %6.2f" %
              (j, load[:, j].mean(), apply_[:, j].mean(), apply_[:, j].mean() / max(j, 1), fact[:, j].mean()))
fs = gain.cpu().numpy().reshape(count, VM * FR)[:, 4 * npan: 4 * npan + 20 * npan].reshape(count, npan, 20) / 100.0
d_ = np.diff(fs, axis=2).mean(axis=0)  # (npan, 19)
for j in (1, 10, 25, 40):
    if j < npan:
        print("block %3d columns: " % j + " ".join("%.1f" % x for x in d_[j, :16]) + "  | rows back %.1f  inverses+lists %.1f  store+desc %.1f" % tuple(d_[j, 16:19]))
