"""Wall time of l3k_elevate_order (device order elevation) on a structured cube's order-1 connectivity with
randomly rotated local frames, next to the host generator of the same mesh.  python tools/bench_elevation.py [--ne 64]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from l3ster_amd import system  # noqa: E402
from test_order_elevation import cube_conn, rotate_elements  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ne", type=int, default=64)
ap.add_argument("--order", type=int, default=6)
a = ap.parse_args()
ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
verts, conn = cube_conn(a.ne)
conn = rotate_elements(conn, seed=1)
system.elevate_order(ctx, conn[:8], verts.shape[0], a.order)  # warm-up (module load)
t = time.time()
en, n_nodes, n_nonint = system.elevate_order(ctx, conn, verts.shape[0], a.order)
t_dev = time.time() - t
t = time.time()
part = system.CubePartition(a.ne, a.order)
t_host = time.time() - t
assert n_nodes == part.n_owned_nodes
print(json.dumps({"workload": f"order elevation {a.ne}^3 hexes to order {a.order}", "elements": int(conn.shape[0]),
                  "nodes": int(n_nodes), "elevate_order_s (H2D conn + device + D2H of the element-node table)": round(t_dev, 3),
                  "host_structured_generator_s": round(t_host, 3), "elem_nodes_MB": round(en.nbytes / 1e6, 1)}))
