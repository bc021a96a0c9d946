"""Kernels on a SUBSET of the node's dofs (field_inds, several kernels sharing one dof map): the strided-dof variant of the one-wave
kernel against the generic kernel that such layouts took before.  Diffusion3D (U = 4) on 4 of 6 dofs per node, 64^3 hexes, orders 4 and 6.

    python tools/r04_strided_dofs.py > profiles/r04_strided_dofs.jsonl
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from l3ster_amd import system  # noqa: E402

torch.cuda.set_device(0)
ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
U, dpn, fi = 4, 6, [4, 0, 5, 2]


def time_apply(mf, X, Y, steps):
    for _ in range(3):
        mf.apply_elems(2, X, None, Y, None, 1.0, 0.0)
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    for i in range(steps):
        e0[i].record()
        mf.apply_elems(2, X, None, Y, None, 1.0, 0.0)
        e1[i].record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in zip(e0, e1)]))


for p in (4, 6):
    part = system.CubePartition(64, p, perturb=0.1)
    mask = np.zeros((part.n_local_nodes, dpn), np.uint8)
    mask[part.node_boundary != 0, fi[0]] = 1
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, dpn, mask.reshape(-1)), system.KERNEL_DIFFUSION3D, [1.0, 1.0], field_inds=fi)
    X = system.synthetic_vector_torch(part.node_grid_id, dpn, "cuda")
    Y = torch.zeros_like(X)
    out = {"order": p, "ne": 64, "dofs_per_node": dpn, "field_inds": fi, "kernel_dofs": part.n_global_nodes * U}
    ys = {}
    for route, below in (("strided", 0), ("generic", 10 ** 9)):
        with ctx.tuning(generic_below=below):
            out[route + "_route"] = mf.route()
            ms = time_apply(mf, X, Y, 10 if route == "strided" else 4)
            Y.zero_()
            mf.apply_elems(2, X, None, Y, None, 1.0, 0.0)
            ys[route] = Y.clone()
        out[route + "_ms"] = ms
        out[route + "_dof_per_s"] = out["kernel_dofs"] / (ms * 1e-3)
    out["strided_over_generic"] = out["generic_ms"] / out["strided_ms"]
    out["rel_diff"] = float((ys["strided"] - ys["generic"]).norm() / ys["generic"].norm())
    print(json.dumps(out), flush=True)
    del mf, X, Y, ys
