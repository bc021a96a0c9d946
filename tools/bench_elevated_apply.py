"""Apply throughput on a device-elevated mesh (numbering [vertices | edges | faces | internal], elements in the input order,
local frames rotated at random) next to the structured generator's mesh (face-run numbering, brick traversal) of the same
cube.   python tools/bench_elevated_apply.py [--ne 32] [--order 6]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("L3K_GENERIC_BELOW", "0")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from l3ster_amd import system  # noqa: E402
from test_order_elevation import cube_conn, rotate_elements  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ne", type=int, default=32)
ap.add_argument("--order", type=int, default=6)
a = ap.parse_args()
U = 4
ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
verts, conn = cube_conn(a.ne)
out = {}
for name, mesh in (("elevated, rotated frames", system.ElevatedHexMesh(ctx, verts, rotate_elements(conn, 1), a.order)),
                   ("elevated", system.ElevatedHexMesh(ctx, verts, conn, a.order)),
                   ("structured generator", system.CubePartition(a.ne, a.order))):
    xyz = mesh.node_coords()
    on_bnd = np.any((np.abs(xyz) < 1e-12) | (np.abs(xyz - 1) < 1e-12), axis=1)
    mask = np.zeros((mesh.n_local_nodes, U), np.uint8)
    mask[on_bnd, 0] = 1
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, mesh, U, mask.reshape(-1)), system.KERNEL_DIFFUSION3D, [1.0, 1.0])
    X = torch.rand((1, mesh.n_local_nodes * U), dtype=torch.float64, device="cuda")
    Y = torch.zeros_like(X)
    for _ in range(3):
        mf.apply(X, Y)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
    ev[0].record()
    for i in range(10):
        mf.apply(X, Y)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(10)]))
    out[name] = {"ms_per_apply": round(ms, 3), "dof_per_s": mesh.n_local_nodes * U / ms * 1e3}
print(json.dumps({"workload": f"Diffusion3D apply, {a.ne}^3 hexes, order {a.order}", **out}))
