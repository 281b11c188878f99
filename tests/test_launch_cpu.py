"""The rank launcher behind `python bench.py --gpus N` (no torchrun): N fresh children, each with its own
LOCAL_RANK / RANK, one WORLD_SIZE and one rendezvous address; rank 0's stdout is the parent's stdout; a failing rank
fails the launch and takes the others down.  The two-rank rendezvous itself runs over gloo on the CPU."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_launcher(tmp_path, body, world, timeout=120):
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(body))
    driver = tmp_path / "driver.py"
    driver.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        from mpc_rs_amd import launch
        assert "torch" not in sys.modules  # the parent stays off the GPU stack
        rc = launch.run_ranks([sys.executable, {str(script)!r}, {str(tmp_path)!r}], {world}, timeout=90)
        assert "torch" not in sys.modules
        sys.exit(rc)
    """))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, str(driver)], capture_output=True, text=True, timeout=timeout, env=env)


def test_launcher_hands_out_distinct_ranks(tmp_path):
    body = """
        import json, os, sys
        out = sys.argv[1]
        keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")
        with open(os.path.join(out, "rank%s.json" % os.environ["RANK"]), "w") as fh:
            json.dump({k: os.environ[k] for k in keys}, fh)
        print("line from rank", os.environ["RANK"])
    """
    world = 4
    p = _run_launcher(tmp_path, body, world)
    assert p.returncode == 0, p.stderr
    recs = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(world)]
    assert sorted(int(r["LOCAL_RANK"]) for r in recs) == list(range(world))
    assert [r["RANK"] for r in recs] == [r["LOCAL_RANK"] for r in recs]
    assert {r["WORLD_SIZE"] for r in recs} == {str(world)}
    assert {r["MASTER_ADDR"] for r in recs} == {"127.0.0.1"} and len({r["MASTER_PORT"] for r in recs}) == 1
    # only rank 0's stdout is the parent's stdout (one JSON line in bench.py); the others go to stderr, tagged
    assert p.stdout.strip() == "line from rank 0"
    for k in range(1, world):
        assert f"[rank {k}] line from rank {k}" in p.stderr


def test_launcher_fails_when_a_rank_fails_and_stops_the_others(tmp_path):
    body = """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(3)
        time.sleep(600)  # a rank waiting in a collective for the one that died
    """
    p = _run_launcher(tmp_path, body, 3, timeout=60)
    assert p.returncode == 3


def test_two_ranks_rendezvous_over_gloo(tmp_path):
    body = """
        import os, sys
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo")
        assert dist.get_world_size() == int(os.environ["WORLD_SIZE"]) == 2
        t = torch.tensor([float(dist.get_rank() + 1)])
        dist.all_reduce(t)
        dist.barrier()
        if dist.get_rank() == 0:
            print("sum", t.item())
        dist.destroy_process_group()
    """
    p = _run_launcher(tmp_path, body, 2, timeout=240)
    assert p.returncode == 0, p.stderr
    assert p.stdout.strip().splitlines()[-1] == "sum 3.0"  # gloo may print its own connection notice first


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       timeout=240, env=env)
    assert p.returncode != 0 and "WORLD_SIZE is 2" in p.stderr
