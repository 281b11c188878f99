"""The four batched workloads at their full benchmark sizes, through the device-pointer C ABI.

The CPU oracle cannot cover a million instances in seconds, so whole-batch correctness is established through
properties that do not depend on the batch size and that, together, determine the outputs:

  Phi     block 0 = C A and block i+1 = (block i) A                      (controller.rs:33-44)
  Gamma   block column 0 = [C B; Phi B] (G_i = C A^i B), columns 1..v-2 are column 0 shifted down one block
          per column with exact zeros above, the last column is the running sum of column 0
                                                                         (controller.rs:46-90)
  H       = Gamma^T blockdiag(Q) Gamma + W3                              (controller.rs:92-93)
  fvec    = Gamma^T blockdiag(Q) (Phi x0 - ref)                          (controller.rs:96, :125-127)
  rollout x_{t+1} = A x_t + B u_t, y_t = C x_t                           (simulator.rs:10-34)
  Phi x0 + Gamma u = outputs y_1..y_f of the rollout driven by u (held after the control horizon)

Every property is evaluated for EVERY instance on the device with torch in f64 (plumbing: batched products of the
library's own outputs), and a spread of instances (first, last, tile boundaries, random) is compared with the
CPU oracle element by element.  Tolerance: 1e-10 relative Frobenius norm per instance.
"""
import numpy as np
import pytest
import torch  # noqa: F401  (loaded at collection time, i.e. before the session's mpcx context brings in its HIP runtime)

from helpers import RTOL

pytestmark = pytest.mark.gpu

FULL = {"msd": 65536, "two_mass": 1 << 20, "quadrotor": 16384, "long": 1024}
CHUNK = {"msd": 65536, "two_mass": 131072, "quadrotor": 1024, "long": 128}
ORACLE_SAMPLES = {"msd": 96, "two_mass": 96, "quadrotor": 16, "long": 4}


def _view(flat, batch, rows, cols, il):
    """(batch, rows, cols) torch view/copy of a flat ABI buffer (column-major per instance, interleave il)."""
    E = rows * cols
    w = batch if il == 0 else il
    if w == 1:
        return flat[: batch * E].view(batch, cols, rows).transpose(1, 2)
    tiles = -(-batch // w)
    return flat[: tiles * w * E].view(tiles, E, w).permute(0, 2, 1).reshape(tiles * w, cols, rows)[:batch].transpose(1, 2)


def _rel(got, want):
    """Per-instance relative Frobenius error, worst over the batch (a float)."""
    import torch
    num = torch.linalg.vector_norm((got - want).reshape(got.shape[0], -1), dim=1)
    den = torch.linalg.vector_norm(want.reshape(want.shape[0], -1), dim=1).clamp_min(1e-300)
    return float((num / den).max())


def _sample_indices(batch, il, count, seed):
    w = batch if il == 0 else il
    fixed = [0, 1, 63, 64, w - 1, w, batch // 2, batch - w, batch - 65, batch - 64, batch - 1]
    rng = np.random.default_rng(seed)
    idx = [i for i in dict.fromkeys(fixed) if 0 <= i < batch][: max(count // 2, 3)]
    while len(idx) < count:
        i = int(rng.integers(0, batch))
        if i not in idx:
            idx.append(i)
    return np.array(sorted(idx[:count]))


@pytest.fixture(scope="module")
def torch_dev():
    import torch
    assert torch.cuda.is_available(), "the full-size tests need the GPU"
    torch.cuda.set_device(0)
    return torch, torch.device("cuda", 0)


def _resident(torch, dev, d, il, batch):
    from mpc_rs_amd import layout as L
    n, r, f = d["n"], d["r"], d["f"]

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    bufs = {k: to_dev(L.pack(d[k], il)) for k in ("A", "B", "C")}
    bufs["x0"] = to_dev(L.pack(d["x0"].reshape(batch, n, 1), il))
    bufs["ref"] = to_dev(L.pack(d["ref"].reshape(batch, f * r, 1), il))
    bufs["W3"] = to_dev(L.colmajor(d["W3"]))
    bufs["Q"] = to_dev(L.colmajor(d["Q"]))
    return bufs


@pytest.mark.parametrize("name", ["msd", "two_mass", "quadrotor", "long"])
def test_qp_build_full_size(ctx, torch_dev, name):
    torch, dev = torch_dev
    from mpc_rs_amd import ffi, layout as L, synth
    from oracle import oracle_ffi as orc

    batch = FULL[name]
    d = synth.make_batch(name, batch)
    n, m, r, f, v = d["n"], d["m"], d["r"], d["f"], d["v"]
    FR, VM = f * r, v * m
    il = 4096 if (m <= 2 and r <= 2 and n <= 6) else L.INTERLEAVE_AOS  # the layout each kernel family is native in
    bufs = _resident(torch, dev, d, il, batch)
    shapes = {"Phi": (FR, n), "Gamma": (FR, VM), "H": (VM, VM), "fvec": (VM, 1)}
    outs = {k: torch.full((L.array_elems(batch, il, rr * cc),), float("nan"), dtype=torch.float64, device=dev)
            for k, (rr, cc) in shapes.items()}
    sp = torch.cuda.current_stream().cuda_stream
    dims = ffi.make_dims(n, m, r, f, v, batch, 0, il)
    ctx.qp_build_dev(dims, bufs["A"].data_ptr(), bufs["B"].data_ptr(), bufs["C"].data_ptr(), bufs["W3"].data_ptr(),
                     bufs["Q"].data_ptr(), bufs["x0"].data_ptr(), bufs["ref"].data_ptr(), outs["Phi"].data_ptr(),
                     outs["Gamma"].data_ptr(), outs["H"].data_ptr(), outs["fvec"].data_ptr(), sp)

    # prediction consistency needs a rollout driven by a per-instance input sequence, held after the control horizon
    rng = np.random.default_rng(77)
    useq = torch.from_numpy(rng.uniform(-1.0, 1.0, (batch, VM))).to(dev)          # (batch, v m), block k = u_k
    T = f + 1
    steps = torch.clamp(torch.arange(T, device=dev), max=v - 1)
    U = useq.view(batch, v, m)[:, steps, :].transpose(1, 2).contiguous()            # (batch, m, T)
    if il == 1:
        u_flat = U.transpose(1, 2).reshape(-1).contiguous()
    else:
        tiles = batch // il
        assert tiles * il == batch
        u_flat = U.transpose(1, 2).reshape(tiles, il, m * T).permute(0, 2, 1).reshape(-1).contiguous()
    X = torch.full((L.array_elems(batch, il, n * (T + 1)),), float("nan"), dtype=torch.float64, device=dev)
    Y = torch.full((L.array_elems(batch, il, r * T),), float("nan"), dtype=torch.float64, device=dev)
    rdims = ffi.make_dims(n, m, r, 1, 1, batch, T, il, u_shared=0)
    ctx.rollout_dev(rdims, bufs["A"].data_ptr(), bufs["B"].data_ptr(), bufs["C"].data_ptr(), bufs["x0"].data_ptr(),
                    u_flat.data_ptr(), X.data_ptr(), Y.data_ptr(), sp)
    torch.cuda.synchronize()

    for k in outs:
        assert bool(torch.isfinite(outs[k]).all()), f"{name}: {k} holds non-finite values"

    W3 = bufs["W3"].view(VM, VM).t()
    Q = bufs["Q"].view(r, r).t()
    full = {k: _view(bufs[k], batch, *s, il) for k, s in
            {"A": (n, n), "B": (n, m), "C": (r, n), "x0": (n, 1), "ref": (FR, 1)}.items()}
    full.update({k: _view(outs[k], batch, *shapes[k], il) for k in outs})
    full["Y"] = _view(Y, batch, r, T, il)
    step = CHUNK[name]

    def sweep(first=0):
        worst = {}

        def note(key, e):
            worst[key] = max(worst.get(key, 0.0), e)

        for lo in range(first, batch, step):
            sl = slice(lo, min(lo + step, batch))
            A, B, Cm, x0, ref = (full[k][sl] for k in ("A", "B", "C", "x0", "ref"))
            Phi, Gam, H, fv = (full[k][sl].contiguous() for k in ("Phi", "Gamma", "H", "fvec"))
            b = Phi.shape[0]
            # Phi: first block and the block recurrence
            note("Phi[0] = C A", _rel(Phi[:, :r], Cm @ A))
            if f > 1:
                note("Phi[i+1] = Phi[i] A", _rel(Phi[:, r:], Phi[:, :-r] @ A))
            # Gamma: Markov blocks in block column 0 (or the running sum when v = 1)
            G = torch.cat([Cm @ B, Phi[:, :-r] @ B], dim=1)                              # (b, f r, m): G_0 .. G_{f-1}
            S = torch.cumsum(G.view(b, f, r, m), dim=1).reshape(b, FR, m)
            if v > 1:
                note("Gamma[:, 0] = Markov blocks", _rel(Gam[:, :, :m], G))
            for k in range(1, v - 1):   # shifted copies, exact zeros above
                col = Gam[:, :, k * m:(k + 1) * m]
                assert bool((col[:, :k * r] == 0).all()), f"{name}: Gamma block column {k} is not zero above the diagonal"
                note("Gamma Toeplitz shift", _rel(col[:, k * r:], Gam[:, :FR - k * r, :m]))
            last = Gam[:, :, (v - 1) * m:]
            assert bool((last[:, :(v - 1) * r] == 0).all()), f"{name}: last Gamma block column is not zero above the diagonal"
            note("Gamma last column = running sum", _rel(last[:, (v - 1) * r:], S[:, :FR - (v - 1) * r]))
            # H and fvec from the library's own Gamma and Phi
            QG = torch.einsum("pq,bfqc->bfpc", Q, Gam.view(b, f, r, VM)).reshape(b, FR, VM)
            note("H = Gamma^T Qbar Gamma + W3", _rel(H, Gam.transpose(1, 2) @ QG + W3))
            s = Phi @ x0 - ref
            z = torch.einsum("pq,bfq->bfp", Q, s.view(b, f, r)).reshape(b, FR, 1)
            note("fvec = Gamma^T Qbar (Phi x0 - ref)", _rel(fv, Gam.transpose(1, 2) @ z))
            # prediction consistency with the rollout kernel
            pred = Phi @ x0 + Gam @ useq[sl].unsqueeze(2)                                  # (b, f r, 1): y_1 .. y_f
            yy = full["Y"][sl][:, :, 1:].transpose(1, 2).reshape(b, FR, 1)
            note("Phi x0 + Gamma u = rollout outputs", _rel(pred, yy))
            del QG, G, S
        return worst

    worst = sweep()
    bad = {k: e for k, e in worst.items() if not e <= RTOL}
    assert not bad, f"{name} (batch {batch}): {bad}"

    if name == "msd":
        # negative control: one element of one instance off by 1e-6 (relative) must trip each property that covers it
        saved = {k: full[k][batch - 1].clone() for k in ("Phi", "Gamma", "H", "fvec")}
        full["Phi"][batch - 1, 5, 1] *= 1.0 + 1e-6
        full["Gamma"][batch - 1, 7, 2] *= 1.0 + 1e-6
        full["H"][batch - 1, 3, 5] *= 1.0 + 1e-6
        full["fvec"][batch - 1, 4, 0] *= 1.0 + 1e-6
        tripped = sweep(first=(batch - 1) // step * step)
        for key in ("Phi[i+1] = Phi[i] A", "Gamma Toeplitz shift", "H = Gamma^T Qbar Gamma + W3",
                    "fvec = Gamma^T Qbar (Phi x0 - ref)", "Phi x0 + Gamma u = rollout outputs"):
            assert tripped[key] > RTOL, f"the property '{key}' did not notice a perturbed element"
        for k, val in saved.items():
            full[k][batch - 1] = val

    # a spread of instances against the CPU oracle, element by element
    idx = _sample_indices(batch, il, ORACLE_SAMPLES[name], seed=5)
    w4 = synth.dense_w4(d["Q"], f)
    want, _ = orc.qp_build_batch(d["A"][idx], d["B"][idx], d["C"][idx], d["W3"], w4, d["x0"][idx], d["ref"][idx], f, v,
                                 threads=8)
    tidx = torch.from_numpy(idx).to(dev)
    for k, (rr, cc) in shapes.items():
        got = full[k][tidx].cpu().numpy()
        ref_k = want[k].reshape(len(idx), rr, cc)
        num = np.linalg.norm((got - ref_k).reshape(len(idx), -1), axis=1)
        den = np.maximum(np.linalg.norm(ref_k.reshape(len(idx), -1), axis=1), 1e-300)
        e = float((num / den).max())
        assert e <= RTOL, f"{name}: {k} differs from the oracle on instance {idx[int(np.argmax(num / den))]}: {e:.3e}"


@pytest.mark.parametrize("name", ["msd", "two_mass", "quadrotor", "long"])
def test_rollout_full_size(ctx, torch_dev, name):
    """The benchmark's rollout (shared unit step input, simulator.rs:54-58) for every instance of the full batch."""
    torch, dev = torch_dev
    from mpc_rs_amd import ffi, layout as L, synth
    from oracle import oracle_ffi as orc

    batch = FULL[name]
    d = synth.make_batch(name, batch)
    n, m, r, T = d["n"], d["m"], d["r"], d["T"]
    il = 4096 if (m <= 2 and r <= 2 and n <= 6) else L.INTERLEAVE_AOS
    bufs = _resident(torch, dev, d, il, batch)
    Uh = torch.from_numpy(L.colmajor(d["U"])).to(dev)
    X = torch.full((L.array_elems(batch, il, n * (T + 1)),), float("nan"), dtype=torch.float64, device=dev)
    Y = torch.full((L.array_elems(batch, il, r * T),), float("nan"), dtype=torch.float64, device=dev)
    dims = ffi.make_dims(n, m, r, 1, 1, batch, T, il, u_shared=1)
    ctx.rollout_dev(dims, bufs["A"].data_ptr(), bufs["B"].data_ptr(), bufs["C"].data_ptr(), bufs["x0"].data_ptr(),
                    Uh.data_ptr(), X.data_ptr(), Y.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(X).all()) and bool(torch.isfinite(Y).all())
    A, B, Cm, x0 = (_view(bufs[k], batch, *s, il) for k, s in
                    {"A": (n, n), "B": (n, m), "C": (r, n), "x0": (n, 1)}.items())
    Xv, Yv = _view(X, batch, n, T + 1, il), _view(Y, batch, r, T, il)
    U = Uh.view(T, m).t()                                                              # (m, T), shared
    worst = {}
    step = CHUNK[name]
    for lo in range(0, batch, step):
        sl = slice(lo, min(lo + step, batch))
        xs = Xv[sl].contiguous()
        assert bool((xs[:, :, :1] == x0[sl]).all()), f"{name}: X[:, 0] is not x0"
        nxt = A[sl] @ xs[:, :, :T] + B[sl] @ U
        # one-step residual relative to the trajectory itself (the state may pass through zero)
        num = torch.linalg.vector_norm((xs[:, :, 1:] - nxt).reshape(xs.shape[0], -1), dim=1)
        den = torch.linalg.vector_norm(xs.reshape(xs.shape[0], -1), dim=1).clamp_min(1e-300)
        worst["x+ = A x + B u"] = max(worst.get("x+ = A x + B u", 0.0), float((num / den).max()))
        worst["y = C x"] = max(worst.get("y = C x", 0.0), _rel(Yv[sl], Cm[sl] @ xs[:, :, :T]))
    bad = {k: e for k, e in worst.items() if not e <= RTOL}
    assert not bad, f"{name} (batch {batch}): {bad}"

    idx = _sample_indices(batch, il, 64, seed=6)
    Xw, Yw, _ = orc.rollout_batch(d["A"][idx], d["B"][idx], d["C"][idx], d["x0"][idx], d["U"], threads=8)
    tidx = torch.from_numpy(idx).to(dev)
    for k, got, want in (("X", Xv[tidx].cpu().numpy(), Xw.reshape(len(idx), n, T + 1)),
                         ("Y", Yv[tidx].cpu().numpy(), Yw.reshape(len(idx), r, T))):
        num = np.linalg.norm((got - want).reshape(len(idx), -1), axis=1)
        den = np.maximum(np.linalg.norm(want.reshape(len(idx), -1), axis=1), 1e-300)
        assert float((num / den).max()) <= RTOL, f"{name}: {k} differs from the oracle: {float((num / den).max()):.3e}"


@pytest.mark.parametrize("name", ["msd", "two_mass"])
def test_gain_solve_loop_full_size(ctx, torch_dev, name):
    """Device chain build -> gain / solve -> closed loop at the full batch (controller.rs:93-96, :117-135,
    main.rs:51-71).  The solves are held to a backward-error bound (independent of cond(H)):
    ||H X - R|| <= 1e-10 (||H|| ||X|| + ||R||); the loop to its own recurrences, evaluated for every instance."""
    torch, dev = torch_dev
    from mpc_rs_amd import ffi, layout as L, synth

    batch = FULL[name]
    d = synth.make_batch(name, batch)
    n, m, r, f, v, T = d["n"], d["m"], d["r"], d["f"], d["v"], d["T"]
    FR, VM = f * r, v * m
    il = 4096
    bufs = _resident(torch, dev, d, il, batch)
    sp = torch.cuda.current_stream().cuda_stream

    def out(elems):
        return torch.full((L.array_elems(batch, il, elems),), float("nan"), dtype=torch.float64, device=dev)
    qp = {"Phi": out(FR * n), "Gamma": out(FR * VM), "H": out(VM * VM), "fvec": out(VM)}
    dims = ffi.make_dims(n, m, r, f, v, batch, T, il, u_shared=1)
    ctx.qp_build_dev(dims, bufs["A"].data_ptr(), bufs["B"].data_ptr(), bufs["C"].data_ptr(), bufs["W3"].data_ptr(),
                     bufs["Q"].data_ptr(), bufs["x0"].data_ptr(), bufs["ref"].data_ptr(), qp["Phi"].data_ptr(),
                     qp["Gamma"].data_ptr(), qp["H"].data_ptr(), qp["fvec"].data_ptr(), sp)
    info_g = torch.full((batch,), -1, dtype=torch.int32, device=dev)
    info_s = torch.full((batch,), -1, dtype=torch.int32, device=dev)
    gain, useq = out(VM * FR), out(VM)
    ctx.qp_gain_dev(dims, qp["H"].data_ptr(), qp["Gamma"].data_ptr(), bufs["Q"].data_ptr(), gain.data_ptr(),
                    info_g.data_ptr(), sp)
    ctx.qp_solve_dev(dims, qp["H"].data_ptr(), qp["fvec"].data_ptr(), useq.data_ptr(), info_s.data_ptr(), sp)
    pulse = np.zeros((r, T + f))
    pulse[:, : (T + f) // 3] = 1.0
    pulse[:, 2 * (T + f) // 3:] = 1.0                                                   # the shape of trajectory.rs:4-8
    traj = torch.from_numpy(L.colmajor(pulse)).to(dev)
    X, U, Y = out(n * (T + 1)), out(m * T), out(r * T)
    ctx.closed_loop_dev(dims, bufs["A"].data_ptr(), bufs["B"].data_ptr(), bufs["C"].data_ptr(), bufs["x0"].data_ptr(),
                        qp["Phi"].data_ptr(), gain.data_ptr(), traj.data_ptr(), X.data_ptr(), U.data_ptr(), Y.data_ptr(), sp)
    torch.cuda.synchronize()
    assert int((info_g != 0).sum()) == 0 and int((info_s != 0).sum()) == 0, "an instance was flagged singular"

    Q = bufs["Q"].view(r, r).t()
    A, B, Cm, x0 = (_view(bufs[k], batch, *s, il) for k, s in
                    {"A": (n, n), "B": (n, m), "C": (r, n), "x0": (n, 1)}.items())
    Phi, Gam, H, fv = (_view(qp[k], batch, *s, il) for k, s in
                       {"Phi": (FR, n), "Gamma": (FR, VM), "H": (VM, VM), "fvec": (VM, 1)}.items())
    K, us = _view(gain, batch, VM, FR, il), _view(useq, batch, VM, 1, il)
    Xv, Uv, Yv = _view(X, batch, n, T + 1, il), _view(U, batch, m, T, il), _view(Y, batch, r, T, il)
    ref_t = traj.view(T + f, r)                                                          # row t = reference at step t
    win = torch.stack([ref_t[t:t + f].reshape(-1) for t in range(T)], dim=1)             # (f r, T): window of step t
    fro = lambda a: torch.linalg.vector_norm(a.reshape(a.shape[0], -1), dim=1)
    worst = {}

    def note(key, e):
        worst[key] = max(worst.get(key, 0.0), float(e))

    step = CHUNK[name]
    for lo in range(0, batch, step):
        sl = slice(lo, min(lo + step, batch))
        h, g, k_, u_ = H[sl].contiguous(), Gam[sl].contiguous(), K[sl].contiguous(), us[sl].contiguous()
        b = h.shape[0]
        rhs = torch.einsum("bfqc,qp->bcfp", g.view(b, f, r, VM), Q).reshape(b, VM, FR)   # Gamma^T blockdiag(Q)
        note("H gain = Gamma^T W4", ((fro(h @ k_ - rhs)) / (fro(h) * fro(k_) + fro(rhs))).max())
        note("H u = -fvec", ((fro(h @ u_ + fv[sl])) / (fro(h) * fro(u_) + fro(fv[sl]))).max())
        xs, uu = Xv[sl].contiguous(), Uv[sl].contiguous()
        assert bool((xs[:, :, :1] == x0[sl]).all())
        # u_t = first m entries of gain (window_t - Phi x_t)   (controller.rs:125-132)
        s = win.unsqueeze(0) - Phi[sl] @ xs[:, :, :T]                                     # (b, f r, T)
        u_want = k_[:, :m, :] @ s
        den = (fro(k_[:, :m, :]) * fro(s)).clamp_min(1e-300)                              # scale of the products summed
        note("u = gain (ref - Phi x)", (fro(uu - u_want) / den).max())
        nxt = A[sl] @ xs[:, :, :T] + B[sl] @ uu
        note("x+ = A x + B u", (fro(xs[:, :, 1:] - nxt) / fro(xs).clamp_min(1e-300)).max())
        note("y = C x", _rel(Yv[sl], Cm[sl] @ xs[:, :, :T]))
    bad = {k: e for k, e in worst.items() if not e <= RTOL}
    assert not bad, f"{name} (batch {batch}): {bad}"


@pytest.mark.parametrize("name", ["quadrotor", "long"])
def test_blocked_gain_solve_full_size(ctx, torch_dev, name):
    """Build -> gain / solve through the blocked solver (200 and 800 unknowns) at the full batch
    (controller.rs:93-96, :130-134): every instance is held to the backward-error bound
    ||H X - R|| <= 1e-10 (||H|| ||X|| + ||R||), which does not depend on cond(H), and a few instances to the CPU oracle."""
    torch, dev = torch_dev
    from mpc_rs_amd import ffi, layout as L, synth
    from oracle import oracle_ffi as orc

    batch = FULL[name]
    d = synth.make_batch(name, batch)
    n, m, r, f, v = d["n"], d["m"], d["r"], d["f"], d["v"]
    FR, VM = f * r, v * m
    il = 1
    bufs = _resident(torch, dev, d, il, batch)
    sp = torch.cuda.current_stream().cuda_stream

    def out(elems):
        return torch.full((L.array_elems(batch, il, elems),), float("nan"), dtype=torch.float64, device=dev)
    qp = {"Gamma": out(FR * VM), "H": out(VM * VM), "fvec": out(VM)}
    dims = ffi.make_dims(n, m, r, f, v, batch, 0, il)
    ctx.qp_build_dev(dims, bufs["A"].data_ptr(), bufs["B"].data_ptr(), bufs["C"].data_ptr(), bufs["W3"].data_ptr(),
                     bufs["Q"].data_ptr(), bufs["x0"].data_ptr(), bufs["ref"].data_ptr(), None,
                     qp["Gamma"].data_ptr(), qp["H"].data_ptr(), qp["fvec"].data_ptr(), sp)
    info_g = torch.full((batch,), -1, dtype=torch.int32, device=dev)
    info_s = torch.full((batch,), -1, dtype=torch.int32, device=dev)
    gain, useq = out(VM * FR), out(VM)
    ctx.qp_gain_dev(dims, qp["H"].data_ptr(), qp["Gamma"].data_ptr(), bufs["Q"].data_ptr(), gain.data_ptr(),
                    info_g.data_ptr(), sp)
    ctx.qp_solve_dev(dims, qp["H"].data_ptr(), qp["fvec"].data_ptr(), useq.data_ptr(), info_s.data_ptr(), sp)
    torch.cuda.synchronize()
    assert int((info_g != 0).sum()) == 0 and int((info_s != 0).sum()) == 0, "an instance was flagged singular"

    Q = bufs["Q"].view(r, r).t()
    Gam, H, fv = (_view(qp[k], batch, *s, il) for k, s in
                  {"Gamma": (FR, VM), "H": (VM, VM), "fvec": (VM, 1)}.items())
    K, us = _view(gain, batch, VM, FR, il), _view(useq, batch, VM, 1, il)
    fro = lambda a: torch.linalg.vector_norm(a.reshape(a.shape[0], -1), dim=1)
    worst = {"H gain = Gamma^T W4": 0.0, "H u = -fvec": 0.0}
    step = CHUNK[name]
    for lo in range(0, batch, step):
        sl = slice(lo, min(lo + step, batch))
        h, g, k_, u_ = H[sl].contiguous(), Gam[sl].contiguous(), K[sl].contiguous(), us[sl].contiguous()
        b = h.shape[0]
        rhs = torch.einsum("bfqc,qp->bcfp", g.view(b, f, r, VM), Q).reshape(b, VM, FR)   # Gamma^T blockdiag(Q)
        worst["H gain = Gamma^T W4"] = max(worst["H gain = Gamma^T W4"],
                                           float(((fro(h @ k_ - rhs)) / (fro(h) * fro(k_) + fro(rhs))).max()))
        worst["H u = -fvec"] = max(worst["H u = -fvec"],
                                   float(((fro(h @ u_ + fv[sl])) / (fro(h) * fro(u_) + fro(fv[sl]))).max()))
    bad = {k: e for k, e in worst.items() if not e <= RTOL}
    assert not bad, f"{name} (batch {batch}): {bad}"

    # a few instances against the oracle's gain, scaled by the conditioning of their Hessian
    for i in _sample_indices(batch, il, 3, 11):
        Hi = np.ascontiguousarray(H[i].cpu().numpy())
        ok, inv = orc.try_inverse(Hi)
        assert ok
        want = inv @ (Gam[i].cpu().numpy().T @ np.kron(np.eye(f), d["Q"]))                 # controller.rs:93-96
        got = K[i].cpu().numpy()
        cond = np.linalg.cond(Hi)
        err = np.linalg.norm(got - want) / np.linalg.norm(want)
        assert err <= 1e-11 * cond, f"{name} instance {i}: gain differs from the oracle by {err:.2e} (cond {cond:.1e})"
