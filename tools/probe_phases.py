"""Phase stamps of the workgroup QP kernel (needs a library built with -DMPCX_WG_STAMPS, e.g. tools/build_variant.sh stamps qp_wg.hip -DMPCX_WG_STAMPS, and MPCX_LIB pointing at it; diagnostic, GPU only).
With STAMPS2=1 in the environment and a -DMPCX_WG_STAMPS2 build: fvec is computed and the first loader wave's stamps around fvec and the held-input sums are printed (pass an output list without Phi: the stamps live in its buffer).
With TIMELINE=1: mean time of each stamp since the kernel start per 256 consecutive workgroups.
With PER_XCD=1: mean start, end and duration of the workgroups of each XCD (workgroup i runs on XCD i % 8)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_rs_amd import ffi, layout as L, synth
name = sys.argv[1]; count = int(sys.argv[2]); want = sys.argv[3].split("+")
dev = torch.device("cuda", 0); ctx = ffi.Context(0)
wl = synth.WORKLOADS[name]; n, m, r, f, v = wl["n"], wl["m"], wl["r"], wl["f"], wl["v"]
if len(sys.argv) > 5:  # optional horizon override: f v
    f, v = int(sys.argv[4]), int(sys.argv[5])
d = synth.make_batch(name, count, f=f, v=v); il = 1
to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bufs = {k: to(L.pack(d[k], il)) for k in ("A", "B", "C")}
bufs["x0"] = to(L.pack(d["x0"].reshape(count, n, 1), il)); bufs["ref"] = to(L.pack(d["ref"].reshape(count, f * r, 1), il))
bufs["W3"] = to(L.colmajor(d["W3"])); bufs["Q"] = to(L.colmajor(d["Q"]))
sizes = {"Phi": f * r * n, "Gamma": f * r * v * m, "H": v * m * v * m, "fvec": v * m}
outs = {k: torch.zeros(count * e, dtype=torch.float64, device=dev) for k, e in sizes.items()}
dims = ffi.make_dims(n, m, r, f, v, count, 0, il)
ptr = lambda k: outs[k].data_ptr() if (k in want or k == "fvec" or (k == "Phi" and os.environ.get("STAMPS2"))) else None
for _ in range(3):
    ctx.qp_build_dev(dims, bufs["A"].data_ptr(), bufs["B"].data_ptr(), bufs["C"].data_ptr(), bufs["W3"].data_ptr(),
                     bufs["Q"].data_ptr(), bufs["x0"].data_ptr(), bufs["ref"].data_ptr(), ptr("Phi"), ptr("Gamma"), ptr("H"), ptr("fvec"), 0)
torch.cuda.synchronize()
if os.environ.get("STAMPS2"):
    raw = outs["Phi"].cpu().numpy().reshape(count, f * r * n)[:, :13]
    mid2 = slice(count // 4, 3 * count // 4)
    for half in (0, 1):
        q = raw[mid2][half::2]
        print("half", half, "loader: S,QG,Z end->fvec start %.1f  fvec %.1f  sums(loader) %.1f ; compute t0->3 %.1f, 3->4 %.1f total %.1f" % tuple(x / 100 for x in ((q[:, 10] - q[:, 2]).mean(), (q[:, 11] - q[:, 10]).mean(), (q[:, 12] - q[:, 11]).mean(), (q[:, 3] - q[:, 0]).mean(), (q[:, 4] - q[:, 3]).mean(), (q[:, 9] - q[:, 0]).mean())))
    print("kernel span %.3f ms" % ((raw[:, 9].max() - raw[:, 0].min()) / 1e5))
    sys.exit(0)
st = outs["fvec"].cpu().numpy().reshape(count, v * m)[:, :10]
names = ["load+recur", "S,QG,Z", "gamma(first)", "fvec", "sweep prep", "ga load", "sweep 1st half", "sweep 2nd half", "gamma(last)"]
dt = np.diff(st, axis=1) / 100.0  # 100 MHz -> us
mid = slice(count // 4, 3 * count // 4)
for half in (0, 1):
    sel = dt[mid][half::2]
    print(("even" if half == 0 else "odd ") + " blocks: " + "  ".join(f"{nm} {x:6.1f}" for nm, x in zip(names, sel.mean(axis=0))) + f"  total {(st[mid][half::2, 9] - st[mid][half::2, 0]).mean() / 100:.1f} us")
print("kernel span %.3f ms" % ((st[:, 9].max() - st[:, 0].min()) / 1e5))
if os.environ.get("TIMELINE"):  # mean time of every stamp since the kernel's start, per group of 256 consecutive workgroups
    t0 = st[:, 0].min()
    for g0 in range(0, count, 256):
        sel = (st[g0:g0 + 256] - t0) / 100.0
        print(f"blocks {g0:5d}+: " + "  ".join(f"s{k} {np.nanmean(sel[:, k]):7.1f}" for k in (0, 1, 2, 3, 4, 6, 8, 9)))
if os.environ.get("PER_XCD"):  # workgroup i runs on XCD i % 8: mean start, end and duration per XCD and dispatch round
    t0 = st[:, 0].min()
    for x in range(8):
        sel = st[x::8]
        print(f"XCD {x}: start {(sel[:, 0] - t0).mean() / 100:7.1f}  end {(sel[:, 9] - t0).mean() / 100:7.1f}  duration {(sel[:, 9] - sel[:, 0]).mean() / 100:6.1f} us;"
              f" first 512: {(sel[:64, 9] - sel[:64, 0]).mean() / 100:6.1f}, rest: {(sel[64:, 9] - sel[64:, 0]).mean() / 100:6.1f}")
