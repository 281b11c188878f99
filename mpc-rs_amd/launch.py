"""One process per GPU, started from a parent that has not touched the GPU.

`run_ranks(cmd, n)` starts `n` fresh child processes of `cmd`, child k with RANK = LOCAL_RANK = k, WORLD_SIZE = n
and a rendezvous address on 127.0.0.1, relays rank 0's stdout to the parent's stdout (the single JSON line of
bench.py) and every other stream to stderr, and returns 0 only when every child exited 0.  The parent must call it
before anything initialises HIP (it imports neither torch nor the GPU library): children are new processes, never a
replacement of a process that holds the GPU.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
import time


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank: int, world: int, port: int, base: dict | None = None) -> dict:
    """Environment of one rank: what torch.distributed's env:// rendezvous reads."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs across processes on this driver
    return env


def _pump(src, dst, prefix: str = ""):
    for line in iter(src.readline, b""):
        dst.write((prefix.encode() + line) if prefix else line)
        dst.flush()
    src.close()


def run_ranks(cmd: list[str], world: int, timeout: float | None = None, extra_env: dict | None = None) -> int:
    """Runs `cmd` as `world` ranks; returns 0 when all exit 0, else the first non-zero exit code (the remaining ranks
    are terminated as soon as one fails, so that a rank stuck in a collective cannot hang the launch)."""
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port()
    procs, pumps = [], []
    out, err = getattr(sys.stdout, "buffer", sys.stdout), getattr(sys.stderr, "buffer", sys.stderr)
    for k in range(world):
        env = rank_env(k, world, port)
        if extra_env:
            env.update(extra_env)
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
        procs.append(p)
        pumps.append(threading.Thread(target=_pump, args=(p.stdout, out if k == 0 else err, "" if k == 0 else f"[rank {k}] "), daemon=True))
        pumps.append(threading.Thread(target=_pump, args=(p.stderr, err, f"[rank {k}] "), daemon=True))
    for t in pumps:
        t.start()
    t0 = time.monotonic()
    rc = 0
    live = set(range(world))
    while live:
        for k in sorted(live):
            code = procs[k].poll()
            if code is None:
                continue
            live.discard(k)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 128 - code
        if rc != 0 or (timeout is not None and time.monotonic() - t0 > timeout):
            if rc == 0:
                rc = 124
            for k in live:  # exactly the children started above
                procs[k].terminate()
            deadline = time.monotonic() + 10.0
            for k in live:
                try:
                    procs[k].wait(timeout=max(0.1, deadline - time.monotonic()))
                except subprocess.TimeoutExpired:
                    procs[k].kill()
                    procs[k].wait()
            live.clear()
        else:
            time.sleep(0.05)
    for t in pumps:
        t.join(timeout=5.0)
    return rc
