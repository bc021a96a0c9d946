"""One process per GPU without an external launcher.

`python bench.py --gpus N` (and the tools that take --gpus) may be typed as such: when no launcher has set WORLD_SIZE, the script
calls `self_launch`, which starts its N ranks as CHILD processes -- the parent never touches the GPU and never replaces itself --
with the rendezvous variables torch.distributed.run would set (RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR =
127.0.0.1, a free MASTER_PORT), relays rank 0's standard output (the one JSON line), sends the other ranks' to standard error,
and fails if any rank fails (the remaining ranks are ended by handle, not by pattern).  The torch.distributed.run spelling keeps
working: with WORLD_SIZE set the script is a rank and this module is not used.
"""
import os
import socket
import subprocess
import sys
import threading
import time


def needs_self_launch(n_ranks):
    return n_ranks > 1 and "WORLD_SIZE" not in os.environ


def self_launch(script, argv, n_ranks, poll_s=0.2):
    """Runs `python script argv...` as n_ranks ranks; returns 0, or raises SystemExit naming the first rank that failed."""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        failed = next(((r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)), None)
        time.sleep(poll_s)
    if failed is None:
        failed = next(((r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0), None)
    if failed is not None:
        for p in procs:  # exactly the processes started above
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=10)
    sys.stdout.write("".join(lines))
    sys.stdout.flush()
    if failed is not None:
        raise SystemExit(f"{os.path.basename(script)}: rank {failed[0]} of {n_ranks} exited with status {failed[1]}")
    return 0
