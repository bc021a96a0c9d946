"""l3ster_amd/launch.py on the CPU: a script typed with --gpus N and no launcher becomes N child ranks with the rendezvous
variables of torch.distributed.run; rank 0's standard output is relayed, a failing rank fails the command and ends the others."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, {root!r})
    from l3ster_amd import launch
    n = int(sys.argv[1])
    if launch.needs_self_launch(n):
        launch.self_launch(__file__, sys.argv[1:], n)
        raise SystemExit(0)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert world == n and os.environ["LOCAL_RANK"] == str(rank) and os.environ["MASTER_ADDR"] == "127.0.0.1"
    if len(sys.argv) > 2 and sys.argv[2] == str(rank):
        raise SystemExit(7)          # this rank fails ...
    if len(sys.argv) > 2:
        time.sleep(60)               # ... while the others would wait (in a collective) for a minute
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    print(f"rank {{rank}} of {{world}} sum {{t.item()}} port {{os.environ['MASTER_PORT']}}", flush=True)
    dist.destroy_process_group()
""")


def run(tmp_path, *args, timeout=120):
    script = tmp_path / "ranks.py"
    script.write_text(SCRIPT.format(root=ROOT))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, str(script), *args], env=env, capture_output=True, text=True, timeout=timeout)


def test_self_launch_runs_the_ranks_and_relays_rank_0(tmp_path):
    r = run(tmp_path, "3")
    assert r.returncode == 0, r.stdout + r.stderr
    out = [ln for ln in r.stdout.splitlines() if ln.strip() and not ln.startswith("[Gloo]")]  # (gloo announces itself on stdout)
    assert len(out) == 1 and out[0].startswith("rank 0 of 3 sum 6.0"), r.stdout  # only rank 0 on standard output
    assert "rank 1 of 3" in r.stderr and "rank 2 of 3" in r.stderr               # the others on standard error


def test_self_launch_fails_when_a_rank_fails(tmp_path):
    import time
    t0 = time.time()
    r = run(tmp_path, "2", "1")
    assert r.returncode != 0 and "rank 1 of 2 exited with status 7" in r.stderr, r.stdout + r.stderr
    assert time.time() - t0 < 45  # the surviving rank was ended, not waited for


def test_under_a_launcher_the_script_is_a_rank(tmp_path):
    script = tmp_path / "ranks.py"
    script.write_text(SCRIPT.format(root=ROOT))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    r = subprocess.run([sys.executable, str(script), "1"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "rank 0 of 1 sum 1.0 port 29611" in r.stdout, r.stdout + r.stderr
