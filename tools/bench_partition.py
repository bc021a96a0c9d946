"""Time of the whole unstructured bring-up of one rank of 8: device order elevation of the global order-1 mesh, extraction
of the rank's part (l3ster_amd/partition.py, torch on the GPU), l3k_mesh_create.   python tools/bench_partition.py [--ne 64]"""
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

from l3ster_amd import partition, system  # noqa: E402
from test_order_elevation import cube_conn, rotate_elements  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ne", type=int, default=64)
ap.add_argument("--order", type=int, default=6)
ap.add_argument("--rank", type=int, default=7)
a = ap.parse_args()
ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
verts, conn = cube_conn(a.ne)
conn = rotate_elements(conn, seed=1)
ne = a.ne
idx = np.arange(ne ** 3)
part = (idx % ne >= ne // 2).astype(int) + 2 * ((idx // ne) % ne >= ne // 2).astype(int) + 4 * (idx // (ne * ne) >= ne // 2).astype(int)
system.elevate_order(ctx, conn[:8], verts.shape[0], a.order)
t0 = time.time()
en, n_nodes, n_nonint = system.elevate_order(ctx, conn, verts.shape[0], a.order)
t1 = time.time()
ev = verts[conn.astype(np.int64)]
m = partition.PartitionedMesh(en, ev, n_nonint, part, a.rank, 8, a.order)
torch.cuda.synchronize()
t2 = time.time()
mesh = system.DeviceMesh(ctx, m, 4)
t3 = time.time()
print(json.dumps({"workload": f"{ne}^3 hexes to order {a.order}, {n_nodes} nodes, rank {a.rank} of 8", "elevate_s": round(t1 - t0, 3),
                  "partition_extraction_s": round(t2 - t1, 3), "mesh_create_s": round(t3 - t2, 3), "rank_elements": m.n_elems,
                  "rank_owned_nodes": m.n_owned_nodes, "rank_ghost_nodes": m.n_ghost_nodes, "neighbours": len(m.nbr_rank)}))
