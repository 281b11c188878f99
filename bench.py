#!/usr/bin/env python3
"""Benchmark of the batched MPC hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload two_mass|msd|quadrotor|long]
                    [--op qp|rollout|solve|gain|loop|plants] [--batch B_per_gpu] [--interleave W]

One "step" = one pass of the hot path over the rank's resident shard of synthetic plants: for --op qp one call of
mpcx_qp_build_f64_dev storing Phi, Gamma, H and fvec for every instance; for --op rollout one call of
mpcx_rollout_f64_dev storing X and Y; --op solve / gain / loop time the batched solve u = -H^-1 f, the gain
H^-1 Gamma^T W4 and the receding-horizon closed loop (X, U, Y stored) on operands a device-side build left in HBM.  Inputs and outputs live in HBM before the timed region starts.  Instances
are sharded by index across ranks (one process per GPU, no collective on the data path, "weak" scaling: the
per-GPU batch is fixed); after the timed region the ranks all-gather one fixed-size statistics record each over
RCCL.  Rank 0 prints ONE JSON line.

`--gpus N` with N > 1 and no torchrun environment (WORLD_SIZE unset) starts N ranks itself: the parent, before it
imports torch or touches HIP, launches N fresh child processes of this script (RANK = LOCAL_RANK = 0..N-1, rendezvous on
127.0.0.1), relays rank 0's JSON line and exits non-zero if any rank does.  Under torchrun the world size must equal
--gpus.  At N = 1 with the default workload the line also carries, under "also", the other configurations (msd,
quadrotor, long-horizon builds incl. held inputs f > v, and rollouts), each timed the same way on a few steps.

The default workload is the one the headline metric is quoted on: the reference's two-mass plant (n=4, m=1, r=1,
f=v=20) perturbed per instance, 1,048,576 instances per GPU (8,388,608 over 8 GPUs).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12      # B/s, MI355X HBM3E
FP64_PEAK = 78.6e12    # FLOP/s, MI355X FP64 vector = matrix

DEFAULT_BATCH = {"two_mass": 1 << 20, "msd": 65536, "quadrotor": 16384, "long": 1024}


def bytes_qp(n, m, r, f, v):
    return 8 * (n * n + n * m + r * n + n + f * r + f * r * n + f * r * v * m + (v * m) ** 2 + v * m)


def bytes_rollout(n, m, r, T):
    return 8 * (n * n + n * m + r * n + n + m * T + n * (T + 1) + r * T)


def bytes_solve(m, v):
    return 8 * ((v * m) ** 2 + 2 * v * m) + 4


def bytes_gain(m, r, f, v):
    return 8 * ((v * m) ** 2 + 2 * f * r * v * m) + 4


def bytes_gain_rows(m, r, f, v):
    # H and Gamma read, the first m rows of the gain stored
    return 8 * ((v * m) ** 2 + f * r * v * m + m * f * r) + 4


def bytes_loop(n, m, r, f, T):
    # A, B, C, x0, Phi, the first m rows of the gain; X, U, Y stored; the reference trajectory is shared
    return 8 * (n * n + n * m + r * n + n + f * r * n + m * f * r + n * (T + 1) + m * T + r * T)


def flops_h(m, r, f, v):
    return (f * r) * (v * m) * (v * m + 1) + 2 * f * r * r * v * m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="two_mass", choices=sorted(DEFAULT_BATCH))
    ap.add_argument("--op", default="qp", choices=["qp", "rollout", "solve", "gain", "gain_rows", "loop", "plants"])
    ap.add_argument("--batch", type=int, default=0, help="instances per GPU (default: the workload's own)")
    ap.add_argument("--interleave", type=int, default=-1, help="-1: layout the kernels are native in")
    ap.add_argument("--f", type=int, default=0, help="prediction horizon override (default: the workload's)")
    ap.add_argument("--v", type=int, default=0, help="control horizon override (default: the workload's, or --f)")
    ap.add_argument("--hessian", type=int, default=0, help="0 auto, 1 dense contraction, 2 structured")
    ap.add_argument("--gather", default="fvec", choices=["none", "fvec", "H", "check"],
                    help="N > 1: result array gathered to rank 0 over RCCL after the timed region (reported apart)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="instances timed on the CPU (default: ~15 s worth)")
    ap.add_argument("--lib-gather", action="store_true",
                    help="N > 1: also exchange the records and gather fvec through the library's own RCCL communicator "
                         "(mpcx_comm_*; always done at N = 1, where it is a one-rank communicator)")
    ap.add_argument("--no-also", action="store_true", help="skip the extra configurations reported under \"also\"")
    ap.add_argument("--also-steps", type=int, default=5)
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # No launcher above us: become one.  Nothing in this process has initialised the GPU (torch is not even
        # imported yet); the ranks are fresh child processes.
        from mpc_rs_amd import launch
        sys.exit(launch.run_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))

    import torch
    import torch.distributed as dist

    from mpc_rs_amd import ffi, layout as L, shard, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher's WORLD_SIZE is {world}")
    # Rehearsal knob (not used by the driver): MPCX_BENCH_REHEARSE=1 runs all ranks on device 0 over gloo, so the
    # N > 1 code path can be exercised on a one-GPU box.  The real run is one rank per GPU over RCCL ("nccl").
    rehearse = os.environ.get("MPCX_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    backend = None
    if not rehearse and torch.cuda.device_count() <= local_rank:
        sys.exit(f"bench.py: rank {rank} needs GPU {local_rank}, this node shows {torch.cuda.device_count()}")
    if world > 1:
        backend = "gloo" if rehearse else "nccl"  # "nccl" is RCCL on ROCm
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ctx = ffi.Context(local_rank)

    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def make_case(workload, op, batch=0, interleave=-1, f_over=0, v_over=0, hessian=0):
        """Synthetic shard of `workload` resident in HBM plus the closure that runs one step of `op` on it."""
        from types import SimpleNamespace
        wl = synth.WORKLOADS[workload]
        n, m, r, f, v, T = wl["n"], wl["m"], wl["r"], wl["f"], wl["v"], wl["T"]
        if f_over:
            f = f_over
            v = v_over or f
        per_gpu = batch or DEFAULT_BATCH[workload]
        lane_native = (m <= 2 and r <= 2 and n <= 6)
        # lane kernels: batch tiled in groups of 4096 instances (measured fastest); workgroup kernels: instance-major
        il = interleave if interleave >= 0 else (4096 if lane_native else L.INTERLEAVE_AOS)
        start, count = shard.shard_range(per_gpu * world, world, rank)
        assert count == per_gpu
        d = synth.make_batch(workload, count, start=start, f=f, v=v)
        bufs = {k: to_dev(L.pack(d[k], il)) for k in ("A", "B", "C")}
        bufs["x0"] = to_dev(L.pack(d["x0"].reshape(count, n, 1), il))
        bufs["ref"] = to_dev(L.pack(d["ref"].reshape(count, f * r, 1), il))
        bufs["W3"] = to_dev(L.colmajor(d["W3"]))
        bufs["Q"] = to_dev(L.colmajor(d["Q"]))
        bufs["U"] = to_dev(L.colmajor(d["U"]))

        def out(elems):
            return torch.empty(L.array_elems(count, il, elems), dtype=torch.float64, device=dev)

        info = None
        h_only = None
        if op == "qp":
            outs = {"Phi": out(f * r * n), "Gamma": out(f * r * v * m), "H": out(v * m * v * m), "fvec": out(v * m)}
            dims = ffi.make_dims(n, m, r, f, v, count, 0, il, ffi.WEIGHTS_BLOCK, hessian)

            def step():
                ctx.qp_build_dev(dims, bufs["A"].data_ptr(), bufs["B"].data_ptr(), bufs["C"].data_ptr(),
                                 bufs["W3"].data_ptr(), bufs["Q"].data_ptr(), bufs["x0"].data_ptr(), bufs["ref"].data_ptr(),
                                 outs["Phi"].data_ptr(), outs["Gamma"].data_ptr(), outs["H"].data_ptr(),
                                 outs["fvec"].data_ptr(), sp)
            def h_only():  # the Hessian alone (with hessian_algo = CONTRACT: the matrix-core kernel on its own)
                ctx.qp_build_dev(dims, bufs["A"].data_ptr(), bufs["B"].data_ptr(), bufs["C"].data_ptr(),
                                 bufs["W3"].data_ptr(), bufs["Q"].data_ptr(), None, None, None, None, outs["H"].data_ptr(),
                                 None, sp)
            unit_bytes = bytes_qp(n, m, r, f, v)
            metric, unit = "batched MPC QP builds/sec (Phi,Gamma,H,f)", "builds/s"
            check_buf = outs["H"]
        elif op == "plants":
            # Model::new for every instance: physical parameters -> discretised A, B, C (two-mass plant, n = 4)
            assert workload == "two_mass", "--op plants is the two-mass parameter sweep"
            rng = np.random.default_rng(1234 + rank)
            prm = np.array([2.0, 2.0, 100.0, 200.0, 1.0, 5.0]) * (1.0 + 0.1 * rng.uniform(-1, 1, (count, 6)))
            bufs["prm"] = to_dev(L.pack(prm.reshape(count, 6, 1), il))
            outs = {"A": out(16), "B": out(4), "C": out(4)}
            info = torch.zeros(count, dtype=torch.int32, device=dev)
            dims = None

            def step():
                ctx.plants_dev(ffi.PLANT_TWO_MASS, count, il, bufs["prm"].data_ptr(), 0.05, outs["A"].data_ptr(),
                               outs["B"].data_ptr(), outs["C"].data_ptr(), info.data_ptr(), sp)
            unit_bytes = 8 * (6 + 16 + 4 + 4) + 4
            metric, unit = "batched plant constructions/sec (parameters -> A,B,C)", "plants/s"
            check_buf = outs["A"]
        elif op in ("solve", "gain", "gain_rows", "loop"):
            # operands come from a device-side build (and gain solve) outside the timed region
            qp = {"Phi": out(f * r * n), "Gamma": out(f * r * v * m), "H": out(v * m * v * m), "fvec": out(v * m)}
            dims = ffi.make_dims(n, m, r, f, v, count, T, il, u_shared=1)
            ctx.qp_build_dev(dims, bufs["A"].data_ptr(), bufs["B"].data_ptr(), bufs["C"].data_ptr(), bufs["W3"].data_ptr(),
                             bufs["Q"].data_ptr(), bufs["x0"].data_ptr(), bufs["ref"].data_ptr(), qp["Phi"].data_ptr(),
                             qp["Gamma"].data_ptr(), qp["H"].data_ptr(), qp["fvec"].data_ptr(), sp)
            info = torch.zeros(count, dtype=torch.int32, device=dev)
            if op == "gain_rows":
                outs = {"rows": out(m * f * r)}

                def step():
                    ctx.qp_gain_rows_dev(dims, qp["H"].data_ptr(), qp["Gamma"].data_ptr(), bufs["Q"].data_ptr(),
                                         outs["rows"].data_ptr(), info.data_ptr(), sp)
                unit_bytes = bytes_gain_rows(m, r, f, v)
                metric, unit = "batched MPC first-move gain rows/sec (rows 0..m-1 of H^-1 Gamma^T W4)", "gains/s"
                check_buf = outs["rows"]
            elif op == "solve":
                outs = {"u_seq": out(v * m)}

                def step():
                    ctx.qp_solve_dev(dims, qp["H"].data_ptr(), qp["fvec"].data_ptr(), outs["u_seq"].data_ptr(), info.data_ptr(), sp)
                unit_bytes = bytes_solve(m, v)
                metric, unit = "batched condensed-QP solves/sec (u = -H^-1 f)", "solves/s"
                check_buf = outs["u_seq"]
            else:
                outs = {"gain": out(v * m * f * r)}

                def gain_step():
                    ctx.qp_gain_dev(dims, qp["H"].data_ptr(), qp["Gamma"].data_ptr(), bufs["Q"].data_ptr(),
                                    outs["gain"].data_ptr(), info.data_ptr(), sp)
                if op == "gain":
                    step = gain_step
                    unit_bytes = bytes_gain(m, r, f, v)
                    metric, unit = "batched MPC gain matrices/sec (H^-1 Gamma^T W4)", "gains/s"
                    check_buf = outs["gain"]
                else:
                    gain_step()
                    pulse = np.zeros((r, T + f))
                    pulse[:, : (T + f) // 3] = 1.0
                    pulse[:, 2 * (T + f) // 3:] = 1.0          # the shape of trajectory.rs:4-8
                    bufs["traj"] = to_dev(L.colmajor(pulse))
                    outs.update({"X": out(n * (T + 1)), "U": out(m * T), "Y": out(r * T)})

                    def step():
                        ctx.closed_loop_dev(dims, bufs["A"].data_ptr(), bufs["B"].data_ptr(), bufs["C"].data_ptr(),
                                            bufs["x0"].data_ptr(), qp["Phi"].data_ptr(), outs["gain"].data_ptr(),
                                            bufs["traj"].data_ptr(), outs["X"].data_ptr(), outs["U"].data_ptr(),
                                            outs["Y"].data_ptr(), sp)
                    unit_bytes = bytes_loop(n, m, r, f, T)
                    metric, unit = "batched MPC closed-loop trajectories/sec (X,U,Y)", "trajectories/s"
                    check_buf = outs["X"]
            outs["_operands"] = qp
        else:
            outs = {"X": out(n * (T + 1)), "Y": out(r * T)}
            dims = ffi.make_dims(n, m, r, 1, 1, count, T, il, u_shared=1)

            def step():
                ctx.rollout_dev(dims, bufs["A"].data_ptr(), bufs["B"].data_ptr(), bufs["C"].data_ptr(), bufs["x0"].data_ptr(),
                                bufs["U"].data_ptr(), outs["X"].data_ptr(), outs["Y"].data_ptr(), sp)
            unit_bytes = bytes_rollout(n, m, r, T) - 8 * m * T  # the shared input is read once, not per instance
            metric, unit = "batched state-space rollout trajectories/sec (X,Y)", "trajectories/s"
            check_buf = outs["X"]
        name = f"{workload}: n={n} m={m} r={r} " + (f"f=v={f}" if f == v else f"f={f} v={v}") + (f" T={T}" if op in ("rollout", "loop") else "")
        return SimpleNamespace(workload=workload, op=op, name=name, n=n, m=m, r=r, f=f, v=v, T=T, il=il, per_gpu=per_gpu,
                               count=count, d=d, bufs=bufs, outs=outs, dims=dims, step=step, unit_bytes=unit_bytes,
                               metric=metric, unit=unit, check_buf=check_buf, info=info, hessian=hessian, h_only=h_only)

    def event_ms(fn, steps):
        """Mean duration of fn() over `steps` back-to-back calls: one HIP event on the launch stream before the first and
        one after the last (an event between two launches makes the second wait for the first one's write-back to
        memory and adds 12-25 us to each step; MPCX_BENCH_EVENTS_PER_STEP=1 brackets every call instead)."""
        if PER_STEP_EVENTS:
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
            for a, b in ev:
                a.record(stream)
                fn()
                b.record(stream)
            torch.cuda.synchronize()
            return float(np.mean([a.elapsed_time(b) for a, b in ev]))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(steps):
            fn()
        b.record(stream)
        torch.cuda.synchronize()
        return a.elapsed_time(b) / steps

    case = make_case(args.workload, args.op, args.batch, args.interleave, args.f, args.v, args.hessian)
    n, m, r, f, v, T, il, per_gpu, count = case.n, case.m, case.r, case.f, case.v, case.T, case.il, case.per_gpu, case.count
    step, outs, unit_bytes, check_buf, bufs, dims = case.step, case.outs, case.unit_bytes, case.check_buf, case.bufs, case.dims

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    if PER_STEP_EVENTS:
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    else:
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))]
    t0 = time.perf_counter()
    if PER_STEP_EVENTS:
        for a, b in ev:
            a.record(stream)
            step()
            b.record(stream)
    else:
        ev[0][0].record(stream)
        for _ in range(args.steps):
            step()
        ev[0][1].record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = float(sum(a.elapsed_time(b) for a, b in ev)) / args.steps  # HIP events on the launch stream

    # ---- hessian_algo = CONTRACT: the dominant kernel is the FP64-MFMA contraction; time it on its own ----
    mfma_ms = None
    if args.op == "qp" and args.hessian == 1:
        case.h_only()
        torch.cuda.synchronize()
        mfma_ms = event_ms(case.h_only, args.steps)

    # ---- statistics record of this shard; ranks exchange records (RCCL), never results ----
    rec = torch.zeros(6, dtype=torch.int64, device=dev)
    ctx.stats_dev(check_buf.data_ptr(), check_buf.numel(), rec.data_ptr(), sp)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        recs = [torch.zeros_like(rec) for _ in range(world)]
        dist.all_gather(recs, rec)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    else:
        recs = [rec]
    torch.cuda.synchronize()
    elapsed = float(tmax.item())
    merged = shard.merge_stats([shard.stats_from_words(x.cpu().numpy()) for x in recs])

    # ---- optional gather of a result array to rank 0 (RCCL over xGMI); never part of the timed region ----
    gathered = None
    if world > 1 and args.gather != "none":
        src = outs.get(args.gather) if args.gather != "check" else check_buf
        if src is None:
            src = check_buf
        try:
            dst = [torch.empty_like(src) for _ in range(world)] if rank == 0 else None
            torch.cuda.synchronize()
            dist.barrier()
            g0 = time.perf_counter()
            dist.gather(src, dst, dst=0)
            torch.cuda.synchronize()
            gms = (time.perf_counter() - g0) * 1e3
            nbytes = src.numel() * src.element_size()
            gathered = {"array": args.gather, "bytes_per_rank": nbytes, "ms": gms,
                        "GB_per_s_into_rank0": nbytes * (world - 1) / gms / 1e6}
            if rank == 0:  # rank 0's own shard must come back unchanged
                gathered["self_check"] = bool(torch.equal(dst[0], src))
            del dst
        except Exception as exc:  # the gather is a report, not the benchmark: never lose the line over it
            gathered = {"array": args.gather, "error": str(exc)[:200]}

    # ---- the same exchange through the library's communicator (what a host without torch.distributed calls) ----
    lib_gather = None
    if world == 1 or args.lib_gather:
        try:
            cid = [ffi.comm_get_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(cid, src=0)
            comm = ffi.Comm(ctx, cid[0], world, rank)
            comm.gather_stats(check_buf.data_ptr(), check_buf.numel(), sp)  # first call: connection set-up
            g0 = time.perf_counter()
            recs2 = comm.gather_stats(check_buf.data_ptr(), check_buf.numel(), sp)
            stats_ms = (time.perf_counter() - g0) * 1e3
            src = outs.get("fvec", check_buf)
            dst = torch.empty(src.numel() * world, dtype=torch.float64, device=dev) if rank == 0 else None
            comm.gather_dev(src.data_ptr(), src.numel(), dst.data_ptr() if rank == 0 else None, 0, sp)
            comm.synchronize()
            g0 = time.perf_counter()
            comm.gather_dev(src.data_ptr(), src.numel(), dst.data_ptr() if rank == 0 else None, 0, sp)
            comm.synchronize()
            arr_ms = (time.perf_counter() - g0) * 1e3
            nbytes = src.numel() * 8
            lib_gather = {"communicator": "mpcx_comm (RCCL, owned by libmpcx.so)", "world": world,
                          "records_equal_torch_exchange": shard.merge_stats(recs2) == merged, "stats_ms": stats_ms,
                          "array_bytes_per_rank": nbytes, "array_ms": arr_ms,
                          "array_GB_per_s_into_rank0": nbytes * max(world - 1, 1) / arr_ms / 1e6}
            if rank == 0:
                lib_gather["self_check"] = bool(torch.equal(dst[rank * src.numel():(rank + 1) * src.numel()], src))
            del dst
            comm.close()
        except Exception as exc:  # a report beside the benchmark: never lose the line over it
            lib_gather = {"error": str(exc)[:200]}

    if rank == 0:
        total = per_gpu * world
        value = total * args.steps / elapsed
        achieved = unit_bytes * per_gpu / (kernel_ms * 1e-3)
        res = {
            "metric": case.metric, "value": value, "unit": case.unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": case.name,
                       "op": args.op, "batch_per_gpu": per_gpu, "global_batch": total, "interleave": il,
                       "parallelism": f"batch-sharded x{world}", "hessian_algo": args.hessian,
                       "world_size": world, "backend": backend},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": _traffic(args.op, args.workload, per_gpu, f, v),
                         "traffic_source": "profiles/traffic.json: rocprofv3 --pmc passes of this command, recorded per launch",
                         "kernel_ms": kernel_ms, "bytes_per_unit": unit_bytes},
            "check": {"elements": merged["count"], "non_finite": merged["non_finite"], "sum_sq": merged["sum_sq"],
                      "checksum": "%016x" % merged["checksum"]},
        }
        if gathered is not None:
            res["gather"] = gathered
        if lib_gather is not None:
            res["lib_gather"] = lib_gather
        if mfma_ms is not None:
            res["roofline"] = _mfma_roofline(case, mfma_ms, kernel_ms, achieved)
        if args.op in ("solve", "gain", "gain_rows"):
            res["roofline"] = _solver_roofline(case, args.op, kernel_ms, res["roofline"])
        if args.op in ("solve", "gain", "gain_rows", "loop", "plants"):
            res["check"]["singular_instances"] = int((case.info != 0).sum().item())
        if world == 1 and not args.no_cpu_baseline and args.op in ("qp", "rollout"):
            res["cpu_baseline"] = _cpu_baseline(args, case.d, n, m, r, f, v, T, case.unit, outs, il, count)

    # ---- the other configurations, N = 1 and the default command only: same timing, a few steps each ----
    also_failed = []
    if world == 1 and not args.no_also and args.workload == "two_mass" and args.op == "qp" and not (args.f or args.batch or args.hessian):
        del case, outs, bufs, check_buf
        torch.cuda.empty_cache()
        also = []
        t_also = time.perf_counter()
        for wl_name, op, fo, vo, hess in ALSO:
            if time.perf_counter() - t_also > ALSO_BUDGET_S:  # a regression, not a detail: reported and the run fails
                also.append({"workload": wl_name, "op": op, "skipped": "time budget"})
                also_failed.append(f"{wl_name}:{op}: not run, the extra configurations exceeded {ALSO_BUDGET_S:.0f} s")
                continue
            try:
                c = make_case(wl_name, op, f_over=fo, v_over=vo, hessian=hess)
                c.step()
                torch.cuda.synchronize()
                t_w = time.perf_counter()
                c.step()
                torch.cuda.synchronize()
                # kernels of a few milliseconds speed up over the first launches of a process (2.53 -> 2.35 ms over six
                # launches of the held-input quadrotor build): about 20 ms of them before the clock starts.  Launches
                # under half a millisecond keep the two warm-up steps (their time does not drift).
                dt_w = time.perf_counter() - t_w
                n_warm = 2 + (min(8, int(0.02 / dt_w)) if dt_w >= 5e-4 else 0)
                for _ in range(n_warm - 2):
                    c.step()
                torch.cuda.synchronize()
                ms = event_ms(c.step, args.also_steps)
                ach = c.unit_bytes * c.per_gpu / (ms * 1e-3)
                rec = {"workload": c.name, "op": op, "batch": c.per_gpu, "interleave": c.il, "steps": args.also_steps, "warmup": n_warm,
                       "ms_per_step": ms, "value": c.per_gpu / (ms * 1e-3), "unit": c.unit,
                       "roofline": {"bound": "hbm", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                    "frac": ach / HBM_PEAK, "traffic": _traffic(op, wl_name, c.per_gpu, c.f, c.v),
                                    "bytes_per_unit": c.unit_bytes}}
                if hess == 1:  # the matrix-core contraction on its own, against the FP64 MFMA peak
                    c.h_only()
                    torch.cuda.synchronize()
                    rec["hessian_algo"] = 1
                    rec["roofline"] = _mfma_roofline(c, event_ms(c.h_only, args.also_steps), ms, ach)
                if op in ("solve", "gain", "gain_rows"):
                    rec["roofline"] = _solver_roofline(c, op, ms, rec["roofline"])
                if c.info is not None:
                    rec["singular_instances"] = int((c.info != 0).sum().item())
                also.append(rec)
                del c
                torch.cuda.empty_cache()
            except Exception as exc:  # reported in the line, and the run fails after printing it
                also.append({"workload": wl_name, "op": op, "error": str(exc)[:200]})
                also_failed.append(f"{wl_name}:{op}: {str(exc)[:200]}")
        res["also"] = also
    if rank == 0:
        print(json.dumps(res), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    if also_failed:  # the line above is complete; an extra configuration that failed or was not run is still a failure
        sys.exit("bench.py: " + "; ".join(also_failed))


# extra configurations of the default run: (workload, op, f override, v override, hessian_algo)
ALSO = (("msd", "qp", 0, 0, 0), ("quadrotor", "qp", 0, 0, 0), ("long", "qp", 0, 0, 0),
        ("quadrotor", "qp", 50, 25, 0), ("long", "qp", 400, 200, 0),
        ("two_mass", "qp", 32, 32, 0), ("two_mass", "qp", 30, 10, 0),   # run-time horizons on the one-lane kernel
        ("long", "qp", 0, 0, 1),                                        # H as the FP64 matrix-core contraction
        ("msd", "rollout", 0, 0, 0), ("two_mass", "rollout", 0, 0, 0), ("quadrotor", "rollout", 0, 0, 0),
        ("two_mass", "solve", 0, 0, 0), ("two_mass", "gain", 0, 0, 0), ("two_mass", "loop", 0, 0, 0),
        ("quadrotor", "loop", 0, 0, 0), ("quadrotor", "gain", 0, 0, 0), ("long", "gain", 0, 0, 0),
        ("two_mass", "gain_rows", 0, 0, 0), ("quadrotor", "gain_rows", 0, 0, 0), ("long", "gain_rows", 0, 0, 0))
PER_STEP_EVENTS = os.environ.get("MPCX_BENCH_EVENTS_PER_STEP", "0") == "1"
ALSO_BUDGET_S = 120.0  # the extras take about 25 s; beyond this something has regressed and the run fails

def _solver_roofline(case, op, ms, hbm_roofline):
    """gain / gain_rows / solve on the blocked solver (more than 64 unknowns): the tighter of the two limits.  Algorithmic
    flops per instance: an LU of the N x N Hessian (2/3 N^3), forward and backward substitution of the right-hand sides
    (2 N^2 each) and, for the gain, forming Gamma^T blockdiag(Q) (2 N (f r) r) -- the blocked elimination executes more
    (it eliminates above the pivot block as well: N^3 for the factorisation), which is not counted.  All of it runs on
    v_mfma_f64_16x16x4_f64.  When the flops at the FP64 matrix peak take longer than the operands at the HBM peak the
    record is the matrix-core one and keeps the HBM fraction beside it."""
    N, fr = case.v * case.m, case.f * case.r
    nrhs = {"gain": fr, "gain_rows": case.m, "solve": 1}[op]
    fl = 2.0 / 3.0 * N ** 3 + 2.0 * N * N * nrhs
    if op == "gain":
        fl += 2.0 * N * fr * case.r
    elif op == "gain_rows":
        fl += 2.0 * case.m * N * fr + 2.0 * case.m * fr * case.r
    t_mfma = fl * case.per_gpu / FP64_PEAK
    t_hbm = case.unit_bytes * case.per_gpu / HBM_PEAK
    if N <= 64 or t_mfma <= t_hbm:
        return hbm_roofline
    ach = fl * case.per_gpu / (ms * 1e-3)
    return {"bound": "mfma", "achieved": ach / 1e12, "peak": FP64_PEAK / 1e12, "unit": "TFLOP/s", "frac": ach / FP64_PEAK,
            "traffic": hbm_roofline.get("traffic"), "flops_per_unit": fl, "bytes_per_unit": case.unit_bytes,
            "hbm_frac": hbm_roofline["frac"], "kernel_ms": ms,
            "note": "algorithmic flops (LU + substitutions) at the FP64 matrix peak take longer than the operands at the HBM peak"}


def _mfma_roofline(case, mfma_ms, step_ms, achieved_bytes_per_s):
    """The contraction kernel against the FP64 matrix-core peak.  `frac` counts the flops the kernel EXECUTES: one
    v_mfma_f64_16x16x4_f64 is 2048 flops and the kernel issues a number of them that depends on the shape only
    (_mfma_insts, checked against the SQ_INSTS_VALU_MFMA_F64 counter pass recorded under profiles/).  The algorithmic
    rate (symmetric count, block-diagonal weight) is a side field: it exceeds what ran only on paper, because the dense
    contraction also multiplies structural zeros."""
    m, r, f, v, per_gpu = case.m, case.r, case.f, case.v, case.per_gpu
    symmetric = bool(np.array_equal(case.d["W3"], case.d["W3"].T) and np.array_equal(case.d["Q"], case.d["Q"].T))
    insts = _mfma_insts(m, r, f, v, per_gpu, symmetric)
    fl_alg = flops_h(m, r, f, v) * per_gpu / (mfma_ms * 1e-3)
    fl_exec = (insts * 2048.0 / (mfma_ms * 1e-3)) if insts else None
    return {"bound": "mfma", "achieved": (fl_exec or 0.0) / 1e12, "peak": FP64_PEAK / 1e12, "unit": "TFLOP/s",
            "frac": (fl_exec / FP64_PEAK) if fl_exec else None, "traffic": None, "kernel_ms": mfma_ms,
            "kernel": "hessian_mfma_kernel", "mfma_insts_per_launch": insts, "symmetric_tiles_skipped": symmetric,
            "algorithmic_TFLOP_per_s": fl_alg / 1e12, "algorithmic_frac": fl_alg / FP64_PEAK,
            "flops_per_unit": flops_h(m, r, f, v), "dense_flops_per_unit": 2 * f * r * (v * m) ** 2,
            "step_ms_all_kernels": step_ms, "hbm_frac_whole_step": achieved_bytes_per_s / HBM_PEAK}


def _mfma_insts(m, r, f, v, per_gpu, symmetric=False):
    """v_mfma_f64_16x16x4_f64 instructions one launch of the contraction kernel issues: a wavefront owns a 48 x 48 tile
    of H as 3 x 3 accumulators and walks the contraction index in steps of 4 from the first row where both operand tiles
    can be non-zero (hessian_mfma.hip); with Q and W3 exactly symmetric only the tiles on and below the diagonal are
    computed.  The count depends on the shape only; the SQ_INSTS_VALU_MFMA_F64 counter passes under profiles/
    (186,753,024 for n=8 m=2 r=2 f=v=400, batch 1024, all tiles) are what this formula was checked against."""
    FR, VM, kT = f * r, v * m, 48
    nt = (VM + kT - 1) // kT
    total = 0
    for ta in range(nt):
        for tb in range(ta + 1 if symmetric else nt):
            sa, sb = (ta * kT // m) * r, (tb * kT // m) * r
            kstart = max(sa, sb) & ~3
            total += 9 * ((FR - kstart + 3) // 4)
    return total * per_gpu


def _traffic(op, workload, per_gpu, f=0, v=0):
    """HBM bytes per launch from the PMC passes recorded under profiles/ (null when no matching record)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            table = json.load(fh)
        rec = table.get(f"{op}:{workload}:{per_gpu}:{f}:{v}") or (table.get(f"{op}:{workload}:{per_gpu}") if f == v else None)
        return rec["hbm_bytes_per_launch"] if rec else None
    except (OSError, ValueError, KeyError):
        return None


def _cpu_baseline(args, d, n, m, r, f, v, T, unit, outs, il, count):
    """The CPU oracle (a port that keeps the reference's algorithmic structure) on a bounded sample of the same
    workload, all host cores of this box's share, timed around the C call only.  The outputs it produces for the
    first instances are also held against what the GPU left in HBM (`parity`: relative Frobenius error per output)."""
    from mpc_rs_amd import layout as L, synth
    from oracle import oracle_ffi as orc
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")), 16)
    est = {"two_mass": 1.2e4, "msd": 1.5e4, "quadrotor": 8.0, "long": 0.4}[args.workload] * cores  # builds/s, rough
    if args.op == "rollout":
        est = 2.0e4 * cores * (200.0 / max(T, 1))
    sample = args.cpu_sample or int(max(cores, min(d["A"].shape[0], est * 12.0)))
    sl = slice(0, sample)
    tm = {}
    if args.op == "qp":
        w4 = synth.dense_w4(d["Q"], f)
        want, _ = orc.qp_build_batch(d["A"][sl], d["B"][sl], d["C"][sl], d["W3"], w4, d["x0"][sl], d["ref"][sl], f, v,
                                     threads=cores, timing=tm)
        shapes = {"Phi": (f * r, n), "Gamma": (f * r, v * m), "H": (v * m, v * m), "fvec": (v * m, 1)}
        secs = tm["seconds"]
    else:
        X, Y, _ = orc.rollout_batch(d["A"][sl], d["B"][sl], d["C"][sl], d["x0"][sl], d["U"], threads=cores, timing=tm)
        want = {"X": X, "Y": Y}
        shapes = {"X": (n, T + 1), "Y": (r, T)}
        secs = tm["seconds"]
    res = {"value": sample / secs, "unit": unit, "cores": cores, "kind": "port",
           "sample": f"first {sample} instances of the same workload, {secs:.1f} s, OpenMP over instances"}
    # parity of what the timed GPU steps left in HBM with the port's outputs, on the first instances of the shard
    w = L.resolve_width(count, il)
    npar = min(sample, count, 4096 if (w == 1 or w >= count) else w)  # tiled layouts: within the first tile
    par = {}
    for k, (rows, cols) in shapes.items():
        E = rows * cols
        if w >= count:  # W = batch: the first instances are the first entries of every element plane
            got = outs[k].view(E, count)[:, :npar].cpu().numpy().T.reshape(npar, cols, rows).transpose(0, 2, 1)
        else:
            got = L.unpack(outs[k][: -(-npar // w) * w * E].cpu().numpy(), npar, rows, cols, il)
        ref = want[k][:npar].reshape(npar, rows, cols)
        num = np.sqrt(((got - ref) ** 2).reshape(npar, -1).sum(axis=1))
        den = np.maximum(np.sqrt((ref ** 2).reshape(npar, -1).sum(axis=1)), 1e-300)
        par[k] = float((num / den).max())
    res["parity"] = {"instances": npar, "max_rel_frobenius_vs_port": par, "tolerance": 1e-10}
    # one core of the same port, on a sample a sixteenth the size
    one = max(1, sample // 16)
    tm1 = {}
    if args.op == "qp":
        orc.qp_build_batch(d["A"][:one], d["B"][:one], d["C"][:one], d["W3"], w4, d["x0"][:one], d["ref"][:one], f, v,
                           threads=1, timing=tm1)
    else:
        orc.rollout_batch(d["A"][:one], d["B"][:one], d["C"][:one], d["x0"][:one], d["U"], threads=1, timing=tm1)
    res["single_core"] = {"value": one / tm1["seconds"], "unit": unit, "cores": 1,
                          "sample": f"first {one} instances, {tm1['seconds']:.1f} s"}
    if args.op == "qp":
        # second line: a CPU implementation written for speed (O(f) recurrences, block weights, no heap traffic), so
        # that the GPU/CPU ratio is not inflated by the reference's own loop structure
        fast_sample = min(d["A"].shape[0], sample * 16)
        fs = slice(0, fast_sample)
        tm2 = {}
        orc.qp_build_batch(d["A"][fs], d["B"][fs], d["C"][fs], d["W3"], synth.dense_w4(d["Q"], f), d["x0"][fs], d["ref"][fs],
                           f, v, threads=cores, timing=tm2, fast=True)
        res["optimised"] = {"value": fast_sample / tm2["seconds"], "unit": unit, "cores": cores,
                            "sample": f"first {fast_sample} instances, {tm2['seconds']:.1f} s"}
    return res


if __name__ == "__main__":
    main()
