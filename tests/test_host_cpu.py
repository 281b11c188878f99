"""CPU tests of the host side (no GPU): batch layouts, sharding, the statistics record, the synthetic workloads,
and that libmpcx.so loads and exports every symbol include/mpcx.h declares (no compute call is made here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from mpc_rs_amd import ffi, layout as L, shard, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mpcx.h")).read()
    declared = set(re.findall(r"\b(mpcx_[a-z0-9_]+)\s*\(", header))
    declared -= {"mpcx_ctx", "mpcx_dims", "mpcx_stats"}
    assert declared == set(ffi.EXPORTS), (declared ^ set(ffi.EXPORTS))
    lib = ffi.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mpcx_version() == 100
    # a context cannot be made without a HIP device: the library reports it, it does not fall back
    if not os.path.exists("/dev/kfd"):
        with pytest.raises(ffi.MpcxError) as e:
            ffi.Context(0)
        assert e.value.code == ffi.ERR_NO_DEVICE


def test_rust_sys_crate_declares_the_same_abi():
    """rust/mpcx-sys cannot be compiled in this image; at least its declarations must follow the header: the same entry
    points with the same number of arguments, the same struct fields in the same order, the same constants."""
    header = open(os.path.join(ROOT, "include", "mpcx.h")).read()
    rust = open(os.path.join(ROOT, "rust", "mpcx-sys", "src", "lib.rs")).read()
    header_nc = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    c_fns = {m.group(1): m.group(2) for m in re.finditer(r"\b(mpcx_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", header_nc, flags=re.S)}
    r_fns = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (mpcx_[a-z0-9_]+)\s*\(([^;]*?)\)\s*(?:->[^;]*)?;", rust, flags=re.S)}
    assert set(c_fns) == set(r_fns) == set(ffi.EXPORTS), set(c_fns) ^ set(r_fns)
    for name, args in c_fns.items():
        n_c = 0 if args.strip() in ("", "void") else args.count(",") + 1
        n_r = 0 if not r_fns[name].strip() else r_fns[name].count(",") + 1
        assert n_c == n_r, f"{name}: {n_c} arguments in the header, {n_r} in the Rust declaration"
    for struct in ("mpcx_dims", "mpcx_stats"):
        c_body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), header_nc, flags=re.S).group(1)
        c_fields = [f.lower() for f in re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*;", c_body)]
        r_body = re.search(r"pub struct %s \{(.*?)\}" % struct, rust, flags=re.S).group(1)
        r_fields = re.findall(r"pub ([a-z_0-9]+):", r_body)
        assert c_fields == r_fields, (struct, c_fields, r_fields)
    for name, value in re.findall(r"#define (MPCX_[A-Z_]+) \(?(-?\d+)\)?", header):
        if name == "MPCX_VERSION":
            continue
        m = re.search(r"pub const %s: [a-z_0-9]+ = (-?\d+);" % name, rust)
        assert m and int(m.group(1)) == int(value), f"{name} differs between the header and the Rust crate"


def test_rust_build_script_compiles_the_sources_the_makefile_compiles():
    """rust/mpcx-sys/build.rs and the build.rs shown in INTEGRATION.md drive hipcc themselves: their source lists must be
    the Makefile's SRCS (a list that misses a file yields a libmpcx.so without the symbols lib.rs declares), and both must
    link libdl, which comm.hip needs for its late binding of RCCL."""
    mk = open(os.path.join(ROOT, "mpc-rs_amd", "csrc", "Makefile")).read()
    srcs = re.search(r"^SRCS\s*:=\s*(.*)$", mk, flags=re.M).group(1).split()
    assert srcs and all(os.path.exists(os.path.join(ROOT, "mpc-rs_amd", "csrc", f)) for f in srcs)
    assert "-ldl" in mk
    rs = open(os.path.join(ROOT, "rust", "mpcx-sys", "build.rs")).read()
    m = re.search(r"const SOURCES: \[&str; (\d+)\] = \[(.*?)\];", rs, flags=re.S)
    listed = re.findall(r'"([^"]+\.hip)"', m.group(2))
    assert listed == srcs and int(m.group(1)) == len(srcs)
    assert '"-ldl"' in rs
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"let src = \[(.*?)\];", doc, flags=re.S)
    assert re.findall(r'"([^"]+\.hip)"', m.group(1)) == srcs
    assert '"-ldl"' in doc


def test_struct_layouts_match_the_header():
    assert ctypes.sizeof(ffi.Dims) == 8 * 4 + 3 * 8
    assert ctypes.sizeof(ffi.Stats) == 48
    assert ffi.Dims.batch.offset == 32 and ffi.Dims.T.offset == 40 and ffi.Dims.interleave.offset == 48


@pytest.mark.parametrize("batch,rows,cols", [(1, 3, 2), (5, 4, 4), (64, 20, 1), (130, 2, 3), (1000, 1, 1)])
@pytest.mark.parametrize("il", [0, 1, 64, 128])
def test_pack_unpack_roundtrip_and_index_formula(batch, rows, cols, il):
    rng = np.random.default_rng(batch * 7 + rows)
    mats = rng.standard_normal((batch, rows, cols))
    flat = L.pack(mats, il)
    E = rows * cols
    assert flat.size == L.array_elems(batch, il, E) == ffi.lib().mpcx_array_elems(batch, il, E)
    W = L.resolve_width(batch, il)
    for b, i, j in [(0, 0, 0), (batch - 1, rows - 1, cols - 1), (batch // 2, rows // 2, cols // 2)]:
        e = j * rows + i
        assert flat[(b // W) * E * W + e * W + (b % W)] == mats[b, i, j]
    assert np.array_equal(L.unpack(flat, batch, rows, cols, il), mats)


def test_pack_empty_and_vectors():
    assert L.pack(np.zeros((0, 3, 3)), 0).size == 0
    assert L.unpack(np.zeros(0), 0, 3, 3, 1).shape == (0, 3, 3)
    v = np.arange(12.0).reshape(4, 3)
    assert np.array_equal(L.pack(v, 1), v.reshape(-1))
    assert np.array_equal(L.pack(v, 0), v.T.reshape(-1))


def test_shard_ranges_cover_the_batch():
    for total in (0, 1, 7, 8, 1000, 8388608):
        for world in (1, 2, 3, 8):
            ranges = [shard.shard_range(total, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and sum(c for _, c in ranges) == total
            for (s0, c0), (s1, _) in zip(ranges, ranges[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in ranges) - min(c for _, c in ranges) <= 1
    with pytest.raises(ValueError):
        shard.shard_range(10, 2, 2)


def test_stats_record_merge_is_shard_invariant():
    rng = np.random.default_rng(1)
    v = rng.standard_normal(10007)
    v[17] = np.nan
    v[4000] = np.inf
    whole = shard.host_stats(v)
    assert whole["count"] == 10007 and whole["non_finite"] == 2
    for world in (2, 3, 8):
        parts = [shard.host_stats(v[s:s + c]) for s, c in (shard.shard_range(v.size, world, r) for r in range(world))]
        m = shard.merge_stats(parts)
        assert m["count"] == whole["count"] and m["non_finite"] == 2 and m["checksum"] == whole["checksum"]
        assert m["max_abs"] == whole["max_abs"] and m["sum_sq"] == pytest.approx(whole["sum_sq"], rel=1e-12)
    words = np.array([5, 1, 0, 0, 0, 0], dtype=np.int64)
    words[2:5] = np.array([1.5, 2.25, 3.0]).view(np.int64)
    words[5] = np.array([0xFFFFFFFFFFFFFFF0], dtype=np.uint64).view(np.int64)[0]
    d = shard.stats_from_words(words)
    assert d == {"count": 5, "non_finite": 1, "sum": 1.5, "sum_sq": 2.25, "max_abs": 3.0, "checksum": 0xFFFFFFFFFFFFFFF0}


@pytest.mark.parametrize("name", sorted(synth.WORKLOADS))
def test_synthetic_workloads_are_reproducible_and_shardable(name):
    full = synth.make_batch(name, 12)
    again = synth.make_batch(name, 12)
    part = synth.make_batch(name, 5, start=4)
    for k in ("A", "B", "C", "x0", "ref"):
        assert np.array_equal(full[k], again[k])
        assert np.array_equal(full[k][4:9], part[k]), k
    n, m, r, f, v = (full[k] for k in ("n", "m", "r", "f", "v"))
    assert full["A"].shape == (12, n, n) and full["B"].shape == (12, n, m) and full["C"].shape == (12, r, n)
    assert full["ref"].shape == (12, f * r) and full["W3"].shape == (v * m, v * m) and full["Q"].shape == (r, r)
    rho = max(np.max(np.abs(np.linalg.eigvals(a))) for a in full["A"])
    assert rho <= 1.0 + 1e-9  # backward Euler of a stable plant never leaves the unit disc
    assert not (np.array_equal(full["A"][0], full["A"][1]) and np.array_equal(full["B"][0], full["B"][1]))


def test_block_weights_match_the_oracle():
    from oracle import oracle_ffi as orc
    for (m, r, f, v) in [(1, 1, 20, 20), (2, 3, 6, 4), (4, 12, 5, 5)]:
        w3, q = synth.block_weights(m, r, v, 1.1e-9, 1e-4, 10.0)
        _, _, w3o, w4o = orc.controller_weights(m, r, f, v, 1.1e-9, 1e-4, 10.0)
        assert np.allclose(w3, w3o, rtol=0, atol=1e-18) and np.array_equal(synth.dense_w4(q, f), w4o)
