"""Times the host-pointer entry point (H2D of the inputs, kernel, D2H of the outputs) on the headline workload.
The rate it prints is the PCIe-inclusive one; bench.py's `value` excludes transfers by construction."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_rs_amd import ffi, layout as L, synth
import ctypes as C

count = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
il = 4096
d = synth.make_batch("two_mass", count)
n, m, r, f, v = 4, 1, 1, 20, 20
ins = [L.pack(d["A"], il), L.pack(d["B"], il), L.pack(d["C"], il), L.colmajor(d["W3"]), L.colmajor(d["Q"]),
       L.pack(d["x0"].reshape(count, n, 1), il), L.pack(d["ref"].reshape(count, f * r, 1), il)]
outs = [np.empty(L.array_elems(count, il, e)) for e in (f * r * n, f * r * v * m, v * m * v * m, v * m)]
dims = ffi.make_dims(n, m, r, f, v, count, 0, il)
with ffi.Context(0) as ctx:
    lib = ffi.lib()
    for it in range(3):
        t0 = time.perf_counter()
        rc = lib.mpcx_qp_build_f64(ctx._h, C.byref(dims), *[x.ctypes.data_as(C.c_void_p) for x in ins],
                                   *[x.ctypes.data_as(C.c_void_p) for x in outs])
        dt = time.perf_counter() - t0
        assert rc == 0
        gb = (sum(x.nbytes for x in ins) + sum(x.nbytes for x in outs)) / 1e9
        print(f"host-pointer call {it}: {dt*1e3:.1f} ms  {count/dt:.3e} builds/s  {gb/dt:.1f} GB/s over PCIe (pageable host memory)")
