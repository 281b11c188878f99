"""GPU parity of the condensed-QP build (Phi, Gamma, H, fvec) against the C oracle, through the C ABI.

Tolerance: relative Frobenius error <= 1e-10 per output, over the batch and for the worst single instance
(tests/helpers.py RTOL).  The GPU evaluates the same quantities with a different association (Markov recurrence
instead of re-multiplied powers, running sums along block diagonals of H), so equality is not bitwise.
"""
import os

import numpy as np
import pytest

from helpers import assert_close, oracle_qp, random_stable_plants
from mpc_rs_amd import ffi, synth
from oracle import oracle_ffi as orc

pytestmark = pytest.mark.gpu


def _run(ctx, data, interleave, want=("Phi", "Gamma", "H", "fvec"), **kw):
    return ctx.qp_build(data["A"], data["B"], data["C"], data["f"], data["v"], W3=data["W3"], W4=data["Q"],
                        x0=data["x0"], ref=data["ref"], interleave=interleave, want=want, **kw)


@pytest.mark.parametrize("name,batch", [("msd", 1000), ("two_mass", 777)])
@pytest.mark.parametrize("interleave", [0, 64, 256, 1])
def test_lane_kernel_workloads(ctx, name, batch, interleave):
    data = synth.make_batch(name, batch)
    want = oracle_qp(orc, data)
    got = _run(ctx, data, interleave)
    for k in ("Phi", "Gamma", "H", "fvec"):
        assert_close(got[k], want[k], f"{name} W={interleave} {k}")


@pytest.mark.parametrize("n,m,r", [(1, 1, 1), (3, 1, 1), (5, 1, 1), (6, 1, 1), (2, 2, 2), (4, 2, 2), (6, 2, 2),
                                   (3, 2, 1), (4, 2, 1), (3, 1, 2), (4, 1, 2)])
@pytest.mark.parametrize("f,v", [(8, 8), (6, 3), (9, 1), (7, 2), (1, 1)])
def test_lane_kernel_shapes_and_held_input(ctx, n, m, r, f, v):
    rng = np.random.default_rng(1000 * n + 100 * m + 10 * r + f + v)
    batch = 130
    A, B, C = random_stable_plants(rng, batch, n, m, r)
    w3 = rng.standard_normal((v * m, v * m))
    q = rng.standard_normal((r, r))  # not symmetric on purpose: the kernels may not assume symmetry
    data = dict(A=A, B=B, C=C, W3=w3, Q=q, x0=rng.standard_normal((batch, n)), ref=rng.standard_normal((batch, f * r)),
                f=f, v=v)
    want = oracle_qp(orc, data)
    for il in (0, 1):
        got = _run(ctx, data, il)
        for k in ("Phi", "Gamma", "H", "fvec"):
            assert_close(got[k], want[k], f"n{n} m{m} r{r} f{f} v{v} W={il} {k}")


@pytest.mark.parametrize("n,m,r", [(4, 1, 1), (2, 1, 1), (4, 2, 2), (6, 2, 2), (3, 2, 1), (4, 1, 2), (5, 2, 2), (6, 3, 3), (2, 3, 1)])
@pytest.mark.parametrize("f,v,rho", [(4, 3, 0.9), (30, 10, 0.9), (60, 5, 0.97), (40, 30, 1.0), (25, 12, 1.05), (33, 3, 0.5)])
def test_lane_kernel_held_inputs_border_blocks(ctx, n, m, r, f, v, rho):
    """Held inputs (3 <= v < f) on the one-lane kernel: the last two block rows and columns of H are correlations of the
    Markov sequence at a lag, which the kernel evaluates through accumulators formed once and an n x m recurrence per
    block instead of the direct sums of controller.rs:63-93.  Windows from 2 to 57 steps, slowly decaying, marginal and
    growing plants, a non-symmetric Q; H alone must equal H of the full call bit for bit."""
    rng = np.random.default_rng(7000 + 1000 * n + 100 * m + 10 * r + f + v)
    batch = 96
    A, B, C = random_stable_plants(rng, batch, n, m, r, rho=rho)
    data = dict(A=A, B=B, C=C, W3=rng.standard_normal((v * m, v * m)), Q=rng.standard_normal((r, r)),
                x0=rng.standard_normal((batch, n)), ref=rng.standard_normal((batch, f * r)), f=f, v=v)
    want = oracle_qp(orc, data)
    for il in (0, 1, 64):
        got = _run(ctx, data, il)
        for k in ("Phi", "Gamma", "H", "fvec"):
            assert_close(got[k], want[k], f"n{n} m{m} r{r} f{f} v{v} rho={rho} W={il} {k}")
    full = _run(ctx, data, 64)
    part = _run(ctx, data, 64, want=("H",))
    assert np.array_equal(part["H"], full["H"]), "H alone differs from the full call"


@pytest.mark.parametrize("n,m,r", [(5, 2, 2), (3, 2, 2), (2, 3, 1), (6, 1, 3), (1, 2, 2), (5, 1, 2), (2, 1, 2), (6, 3, 3),
                                   (1, 3, 3), (4, 3, 2)])
@pytest.mark.parametrize("f,v", [(20, 20), (9, 4), (5, 1)])
def test_small_plants_outside_the_exact_lane_shapes(ctx, n, m, r, f, v):
    """Plants of at most 6 states and 3 inputs / outputs whose (n, m, r) is not one of the exactly instantiated one-lane
    shapes: block weights as Controller::new builds them (controller.rs:149-185, all-elements blocks and q I blocks),
    every batch layout, ragged batch."""
    rng = np.random.default_rng(3000 + 100 * n + 10 * m + r + f + v)
    batch = 200
    A, B, C = random_stable_plants(rng, batch, n, m, r)
    for all_elements in (True, False):
        w3, q = synth.block_weights(m, r, v, 0.3, 0.7, 2.0, all_elements=all_elements)
        data = dict(A=A, B=B, C=C, W3=w3, Q=q, x0=rng.standard_normal((batch, n)), ref=rng.standard_normal((batch, f * r)),
                    f=f, v=v)
        want = oracle_qp(orc, data)
        for il in (0, 1, 64):
            got = _run(ctx, data, il)
            for k in ("Phi", "Gamma", "H", "fvec"):
                assert_close(got[k], want[k], f"n{n} m{m} r{r} f{f} v{v} W={il} all_elements={all_elements} {k}")
    # output subsets must equal the full call bit for bit
    full = _run(ctx, data, 64)
    for sub in (("Gamma",), ("H", "fvec"), ("Phi", "fvec")):
        part = _run(ctx, data, 64, want=sub)
        for k in sub:
            assert np.array_equal(part[k], full[k]), f"n{n} m{m} r{r}: {k} alone differs from the full call"


@pytest.mark.parametrize("n,m,r,f,v", [(12, 4, 12, 10, 10), (12, 4, 12, 9, 4), (8, 2, 2, 40, 40), (8, 2, 2, 30, 7),
                                       (7, 3, 5, 6, 6), (9, 1, 1, 12, 2), (16, 5, 3, 5, 1), (10, 2, 4, 3, 3),
                                       (8, 4, 4, 11, 11), (7, 3, 3, 9, 5), (9, 2, 4, 8, 8), (9, 1, 1, 300, 300),
                                       (8, 2, 2, 150, 150), (8, 2, 2, 140, 131), (20, 2, 2, 6, 6), (12, 4, 12, 3, 2)])
def test_workgroup_kernel_shapes(ctx, n, m, r, f, v):
    rng = np.random.default_rng(n * 7 + m * 5 + r * 3 + f + v)
    batch = 9
    A, B, C = random_stable_plants(rng, batch, n, m, r)
    data = dict(A=A, B=B, C=C, W3=rng.standard_normal((v * m, v * m)), Q=rng.standard_normal((r, r)),
                x0=rng.standard_normal((batch, n)), ref=rng.standard_normal((batch, f * r)), f=f, v=v)
    want = oracle_qp(orc, data)
    got = _run(ctx, data, 1)
    for k in ("Phi", "Gamma", "H", "fvec"):
        assert_close(got[k], want[k], f"wg n{n} m{m} r{r} f{f} v{v} {k}")


@pytest.mark.parametrize("n,m,r,f,v", [(8, 2, 2, 40, 40), (8, 2, 2, 30, 7), (12, 4, 12, 10, 10), (9, 1, 1, 300, 300),
                                       (7, 3, 3, 9, 5), (8, 4, 4, 11, 11), (9, 2, 4, 8, 8), (20, 2, 2, 6, 6)])
@pytest.mark.parametrize("pattern", ["tridiagonal", "zero", "spikes", "lower", "one_column", "nan"])
def test_workgroup_kernel_sparse_input_weight(ctx, n, m, r, f, v, pattern):
    """W3 with exact zeros: the workgroup kernels fetch, per column block, only the rows that hold non-zeros (of that
    block and of the block two to its right), so every zero pattern must give the same H as the dense path.
    "tridiagonal" is what Controller::new builds (controller.rs:149-177)."""
    rng = np.random.default_rng(n * 11 + m * 5 + r * 3 + f + v + len(pattern))
    batch, vm = 5, v * m
    A, B, C = random_stable_plants(rng, batch, n, m, r)
    dense = rng.standard_normal((vm, vm))
    if pattern == "tridiagonal":
        w3, _ = synth.block_weights(m, r, v, 0.3, 0.7, 1.0)
    elif pattern == "zero":
        w3 = np.zeros((vm, vm))
    elif pattern == "spikes":  # a few isolated entries anywhere, corners included
        w3 = np.zeros((vm, vm))
        w3[0, vm - 1] = 2.0
        w3[vm - 1, 0] = -3.0
        for _ in range(4):
            w3[rng.integers(vm), rng.integers(vm)] = rng.standard_normal()
    elif pattern == "lower":
        w3 = np.tril(dense)
    elif pattern == "one_column":
        w3 = np.zeros((vm, vm))
        w3[:, vm // 2] = dense[:, 0]
    else:  # a NaN counts as a non-zero and must reach H
        w3 = np.zeros((vm, vm))
        w3[vm // 3, vm // 2] = np.nan
    data = dict(A=A, B=B, C=C, W3=w3, Q=rng.standard_normal((r, r)), x0=rng.standard_normal((batch, n)),
                ref=rng.standard_normal((batch, f * r)), f=f, v=v)
    got = _run(ctx, data, 1, want=("H",))["H"]
    base = _run(ctx, dict(data, W3=np.zeros((vm, vm))), 1, want=("H",))["H"]  # Gamma^T Qbar Gamma alone
    want = base + w3[None]
    if pattern == "nan":
        assert np.array_equal(np.isnan(got), np.isnan(want))
        got, want = np.nan_to_num(got), np.nan_to_num(want)
    assert np.array_equal(got, want), f"{pattern}: H differs from (H with W3 = 0) + W3 in {np.count_nonzero(got != want)} entries"
    if pattern != "nan":
        assert_close(got, oracle_qp(orc, data, want=("H",))["H"], f"wg sparse W3 {pattern} n{n} m{m} r{r} f{f} v{v} H")


@pytest.mark.parametrize("n,m,r,f,v", [
    (10, 8, 2, 9, 9), (10, 8, 2, 12, 5),      # 2 m = 16: one shift per operand row of the held-input sums
    (10, 3, 4, 14, 6), (9, 5, 2, 11, 4),      # 16 / m and 16 / (2 m) leave dead operand columns
    (12, 9, 2, 7, 3), (18, 16, 2, 5, 2),      # 2 m > 16: held-input sums as dot products; fvec with one shift per row
    (18, 17, 1, 4, 2), (18, 17, 2, 3, 3),     # m > 16: fvec as dot products as well
    (8, 2, 3, 41, 17), (8, 2, 3, 41, 41),     # odd r: Gamma stored 8 bytes per lane
    (8, 2, 2, 333, 40), (9, 1, 2, 520, 3),    # more than 16 operand-row groups of fvec, long held-input sums
    (12, 4, 12, 50, 2), (12, 4, 12, 50, 49),  # shortest and longest swept part with held inputs
])
def test_workgroup_kernel_matrix_core_sums(ctx, n, m, r, f, v):
    """fvec as a Hankel product, the held-input sums (both products of a tile in one wavefront) and their fall-backs,
    over input counts that fill, under-fill and exceed the 16 columns of the matrix-core tile."""
    rng = np.random.default_rng(31 * n + 17 * m + 13 * r + 7 * f + v)
    batch = 5
    A, B, C = random_stable_plants(rng, batch, n, m, r)
    data = dict(A=A, B=B, C=C, W3=rng.standard_normal((v * m, v * m)), Q=rng.standard_normal((r, r)),
                x0=rng.standard_normal((batch, n)), ref=rng.standard_normal((batch, f * r)), f=f, v=v)
    want = oracle_qp(orc, data)
    got = _run(ctx, data, 1)
    for k in ("Phi", "Gamma", "H", "fvec"):
        assert_close(got[k], want[k], f"wg sums n{n} m{m} r{r} f{f} v{v} {k}")
    only = _run(ctx, data, 1, want=("fvec",))
    assert np.array_equal(only["fvec"], got["fvec"]), "fvec alone differs from the full call"
    hf = _run(ctx, data, 1, want=("H", "fvec"))  # no Gamma: every wavefront takes held-input tiles
    assert np.array_equal(hf["H"], got["H"]) and np.array_equal(hf["fvec"], got["fvec"])


@pytest.mark.parametrize("n,m,r,f,v,batch", [(8, 2, 2, 30, 7, 600), (12, 4, 12, 9, 4, 520), (10, 3, 4, 14, 6, 700)])
def test_workgroup_kernel_held_inputs_second_workgroup_per_cu(ctx, n, m, r, f, v, batch):
    """Held inputs with more instances than compute units but at most four per CU: the workgroups of the second CU
    slot sweep first and write Gamma last (and take part in the held-input sums); every instance against the oracle."""
    rng = np.random.default_rng(n + 3 * m + 5 * r + 7 * f + 11 * v + batch)
    A, B, C = random_stable_plants(rng, batch, n, m, r)
    data = dict(A=A, B=B, C=C, W3=rng.standard_normal((v * m, v * m)), Q=rng.standard_normal((r, r)),
                x0=rng.standard_normal((batch, n)), ref=rng.standard_normal((batch, f * r)), f=f, v=v)
    want = oracle_qp(orc, data)
    got = _run(ctx, data, 1)
    for k in ("Phi", "Gamma", "H", "fvec"):
        assert_close(got[k], want[k], f"wg held, two slots n{n} m{m} r{r} f{f} v{v} {k}")


def test_workgroup_kernel_gamma_store_width(ctx):
    """Gamma goes out 16 bytes per lane when the buffer is 16-byte aligned and 8 bytes per lane otherwise: the same
    bits either way (device pointers, the second buffer one double further on)."""
    import torch
    from mpc_rs_amd import layout as L
    n, m, r, f, v, batch = 8, 2, 2, 37, 21, 6
    rng = np.random.default_rng(77)
    A, B, C = random_stable_plants(rng, batch, n, m, r)
    dev = torch.device("cuda", 0)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    bufs = [to(L.pack(x, 1)) for x in (A, B, C)]
    w3, q = to(L.colmajor(rng.standard_normal((v * m, v * m)))), to(L.colmajor(rng.standard_normal((r, r))))
    elems = f * r * v * m * batch
    out = torch.zeros(2 * elems + 2, dtype=torch.float64, device=dev)
    assert out.data_ptr() % 16 == 0
    dims = ffi.make_dims(n, m, r, f, v, batch, 0, 1)
    for off in (0, elems + 1):  # byte offsets 0 and 8 (mod 16)
        ctx.qp_build_dev(dims, bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), w3.data_ptr(), q.data_ptr(),
                         None, None, None, out.data_ptr() + 8 * off, None, None, 0)
    torch.cuda.synchronize()
    g = out.cpu().numpy()
    assert np.array_equal(g[:elems], g[elems + 1:2 * elems + 1])
    data = dict(A=A, B=B, C=C, W3=np.zeros((v * m, v * m)), Q=np.eye(r), x0=np.zeros((batch, n)),
                ref=np.zeros((batch, f * r)), f=f, v=v)
    want = oracle_qp(orc, data)["Gamma"]
    got = g[:elems].reshape(batch, v * m, f * r).transpose(0, 2, 1)
    assert_close(got, want, "Gamma through device pointers")


@pytest.mark.parametrize("seed", range(24))
def test_workgroup_kernel_random_shapes(ctx, seed):
    """Seeded random shapes on the workgroup kernels: strides 1, 2 and 4 of the recurrences, held inputs (f > v),
    odd and even v m, output subsets, banded and dense W3."""
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.integers(7, 17))
    m = int(rng.integers(1, 5))
    r = int(rng.integers(1, 7))
    f = int(rng.integers(1, 49))
    v = int(rng.integers(1, f + 1))
    batch = int(rng.integers(1, 6))
    A, B, C = random_stable_plants(rng, batch, n, m, r)
    if seed % 3 == 0:
        w3, _ = synth.block_weights(m, r, v, 0.3, 0.7, 1.0, all_elements=bool(seed % 2))
    else:
        w3 = rng.standard_normal((v * m, v * m))
    data = dict(A=A, B=B, C=C, W3=w3, Q=rng.standard_normal((r, r)), x0=rng.standard_normal((batch, n)),
                ref=rng.standard_normal((batch, f * r)), f=f, v=v)
    want = oracle_qp(orc, data)
    got = _run(ctx, data, 1)
    for k in ("Phi", "Gamma", "H", "fvec"):
        assert_close(got[k], want[k], f"wg random n{n} m{m} r{r} f{f} v{v} {k}")
    sub = ("Phi", "H") if seed % 2 else ("Gamma",)
    part = _run(ctx, data, 1, want=sub)
    for k in sub:
        assert np.array_equal(part[k], got[k]), f"subset {sub}: {k} differs from the full call"


def test_quadrotor_and_long_horizon_workloads(ctx):
    for name, batch in (("quadrotor", 6), ("long", 2)):
        data = synth.make_batch(name, batch)
        want = oracle_qp(orc, data)
        got = _run(ctx, data, 1)
        for k in ("Phi", "Gamma", "H", "fvec"):
            assert_close(got[k], want[k], f"{name} {k}")


def test_output_subsets(ctx):
    for name, il in (("two_mass", 0), ("quadrotor", 1)):
        data = synth.make_batch(name, 70 if name == "two_mass" else 3)
        full = _run(ctx, data, il)
        for subset in (("Phi",), ("Gamma",), ("H",), ("fvec",), ("Phi", "H"), ("Gamma", "fvec")):
            part = _run(ctx, data, il, want=subset)
            assert set(part) == set(subset)
            for k in subset:
                assert np.array_equal(part[k], full[k]), f"{name}: {k} differs when only {subset} is requested"


def test_batch_of_one_and_empty(ctx):
    data = synth.make_batch("two_mass", 1)
    want = oracle_qp(orc, data)
    for il in (0, 1):
        got = _run(ctx, data, il)
        for k in got:
            assert_close(got[k], want[k], f"batch 1 W={il} {k}")
    empty = synth.make_batch("two_mass", 0)
    got = _run(ctx, empty, 1)
    assert got["Phi"].shape == (0, 20, 4) and got["H"].shape == (0, 20, 20)


def test_argument_errors(ctx):
    data = synth.make_batch("two_mass", 4)
    with pytest.raises(ffi.MpcxError) as e:
        ctx.qp_build(data["A"], data["B"], data["C"], 5, 6, W3=np.zeros((6, 6)), W4=data["Q"], x0=data["x0"],
                     ref=np.zeros((4, 5)))
    assert e.value.code == ffi.ERR_DIMS
    with pytest.raises(ffi.MpcxError) as e:
        ctx.qp_build(data["A"], data["B"], data["C"], 20, 20, W3=None, W4=data["Q"], x0=data["x0"], ref=data["ref"])
    assert e.value.code == ffi.ERR_NULL
    with pytest.raises(ffi.MpcxError) as e:
        _run(ctx, data, 48)  # not 0, 1 or a multiple of 64
    assert e.value.code == ffi.ERR_LAYOUT


@pytest.mark.parametrize("n,m,r,f,v", [(12, 4, 12, 10, 10), (12, 4, 12, 9, 4), (8, 2, 2, 40, 40), (8, 2, 2, 30, 7),
                                       (7, 3, 5, 6, 6), (4, 1, 1, 20, 20), (9, 1, 1, 70, 70), (16, 5, 3, 5, 1),
                                       (8, 2, 2, 150, 150), (6, 2, 2, 33, 17)])
def test_hessian_contraction_on_matrix_cores(ctx, n, m, r, f, v):
    """hessian_algo = CONTRACT: H = Gamma^T (W4 Gamma) + W3 through the FP64 MFMA kernel; the other outputs unchanged."""
    rng = np.random.default_rng(n * 11 + m * 5 + r * 3 + f + v)
    batch = 5
    A, B, C = random_stable_plants(rng, batch, n, m, r)
    data = dict(A=A, B=B, C=C, W3=rng.standard_normal((v * m, v * m)), Q=rng.standard_normal((r, r)),
                x0=rng.standard_normal((batch, n)), ref=rng.standard_normal((batch, f * r)), f=f, v=v)
    want = oracle_qp(orc, data)
    got = _run(ctx, data, 1, hessian_algo=ffi.HESSIAN_CONTRACT)
    for k in ("Phi", "Gamma", "H", "fvec"):
        assert_close(got[k], want[k], f"contract n{n} m{m} r{r} f{f} v{v} {k}")
    only_h = _run(ctx, data, 1, want=("H",), hessian_algo=ffi.HESSIAN_CONTRACT)
    assert np.array_equal(only_h["H"], got["H"])
    structured = _run(ctx, data, 1, want=("H",), hessian_algo=ffi.HESSIAN_STRUCTURED)
    assert_close(structured["H"], got["H"], "structured vs contraction")
    # exactly symmetric weights: the kernel computes the tiles on and below the diagonal only and mirrors the rest
    sym = dict(data, W3=data["W3"] + data["W3"].T, Q=data["Q"] + data["Q"].T)
    want_s = oracle_qp(orc, sym, want=("H",))
    got_s = _run(ctx, sym, 1, want=("H",), hessian_algo=ffi.HESSIAN_CONTRACT)
    assert_close(got_s["H"], want_s["H"], f"contract, symmetric weights n{n} m{m} r{r} f{f} v{v}")
    if v * m > 48:  # tiles off the diagonal (48 x 48) are exact copies of their mirror images
        assert np.array_equal(got_s["H"][:, 48:, :48], got_s["H"][:, :48, 48:].transpose(0, 2, 1))
    # one weight symmetric, the other not: no mirroring
    half = dict(data, W3=sym["W3"])
    assert_close(_run(ctx, half, 1, want=("H",), hessian_algo=ffi.HESSIAN_CONTRACT)["H"], oracle_qp(orc, half, want=("H",))["H"],
                 "contract, only W3 symmetric")


@pytest.mark.parametrize("interleave", [0, 64, 128])
@pytest.mark.parametrize("batch", [1, 70, 200])
def test_interleaved_batches_on_the_instance_major_kernels(ctx, interleave, batch):
    """The workgroup kernels, the matrix-core contraction and the dense-weight kernel take instance-major data; an
    interleaved batch goes through an instance-major copy inside the library and must give the same bits as the
    same call with interleave = 1 (ragged last tile, batch of one, output subsets included)."""
    rng = np.random.default_rng(31 * batch + interleave)
    for (n, m, r, f, v) in ((8, 2, 2, 12, 12), (12, 4, 12, 5, 3), (4, 1, 1, 170, 170)):  # the last: a lane shape too long for LDS
        A, B, C = random_stable_plants(rng, batch, n, m, r)
        data = dict(A=A, B=B, C=C, W3=rng.standard_normal((v * m, v * m)), Q=rng.standard_normal((r, r)),
                    x0=rng.standard_normal((batch, n)), ref=rng.standard_normal((batch, f * r)), f=f, v=v)
        base = _run(ctx, data, 1)
        got = _run(ctx, data, interleave)
        for k in ("Phi", "Gamma", "H", "fvec"):
            assert np.array_equal(got[k], base[k]), f"n{n} m{m} r{r} W={interleave} batch {batch}: {k} differs from interleave = 1"
        part = _run(ctx, data, interleave, want=("Gamma", "fvec"))
        assert np.array_equal(part["Gamma"], base["Gamma"]) and np.array_equal(part["fvec"], base["fvec"])
        if batch == 200:  # a scratch cap of 3.5 instances' worth (test knob): the batch goes through in many chunks
            per = 8 * (n * n + n * m + r * n + n + f * r + f * r * n + f * r * v * m + (v * m) ** 2 + v * m)
            ctx.set_option(ffi.OPT_RELAY_CAP_BYTES, int(3.5 * per))
            try:
                chunked = _run(ctx, data, interleave)
            finally:
                ctx.set_option(ffi.OPT_RELAY_CAP_BYTES, 2 << 30)
            for k in ("Phi", "Gamma", "H", "fvec"):
                assert np.array_equal(chunked[k], base[k]), f"chunked relay: {k} differs"
        if n == 8:
            want = oracle_qp(orc, data)
            for k in ("Phi", "Gamma", "H", "fvec"):
                assert_close(got[k], want[k], f"relayed n{n} W={interleave} {k}")
            c1 = _run(ctx, data, 1, want=("H",), hessian_algo=ffi.HESSIAN_CONTRACT)["H"]
            cw = _run(ctx, data, interleave, want=("H",), hessian_algo=ffi.HESSIAN_CONTRACT)["H"]
            assert np.array_equal(cw, c1), "matrix-core contraction: interleaved batch differs from interleave = 1"
            W4 = rng.standard_normal((f * r, f * r))
            kw = dict(W3=data["W3"], W4=W4, x0=data["x0"], ref=data["ref"], weights_mode=ffi.WEIGHTS_DENSE)
            d1 = ctx.qp_build(A, B, C, f, v, interleave=1, **kw)
            dw = ctx.qp_build(A, B, C, f, v, interleave=interleave, **kw)
            for k in ("Phi", "Gamma", "H", "fvec"):
                assert np.array_equal(dw[k], d1[k]), f"dense W4: {k} differs between interleave {interleave} and 1"


@pytest.mark.parametrize("n,m,r,f,v", [(2, 1, 1, 20, 20), (4, 2, 3, 9, 4), (12, 4, 12, 6, 6), (8, 2, 2, 25, 25), (3, 1, 2, 7, 1),
                                       (5, 1, 1, 1, 1)])
def test_dense_output_weight(ctx, n, m, r, f, v):
    """weights_mode = DENSE: an arbitrary dense W4, the literal signature of form_lifted_matrices (controller.rs:23)."""
    rng = np.random.default_rng(n + 10 * m + 100 * r + f * v)
    batch = 4
    A, B, C = random_stable_plants(rng, batch, n, m, r)
    W3, W4 = rng.standard_normal((v * m, v * m)), rng.standard_normal((f * r, f * r))
    x0, ref = rng.standard_normal((batch, n)), rng.standard_normal((batch, f * r))
    want, _ = orc.qp_build_batch(A, B, C, W3, W4, x0, ref, f, v, threads=2)
    got = ctx.qp_build(A, B, C, f, v, W3=W3, W4=W4, x0=x0, ref=ref, interleave=1, weights_mode=ffi.WEIGHTS_DENSE)
    for k in ("Phi", "Gamma", "H", "fvec"):
        assert_close(got[k], want[k], f"dense W4 n{n} m{m} r{r} f{f} v{v} {k}")
    part = ctx.qp_build(A, B, C, f, v, W3=W3, W4=W4, x0=x0, ref=ref, interleave=1, weights_mode=ffi.WEIGHTS_DENSE, want=("fvec",))
    assert np.array_equal(part["fvec"], got["fvec"])
