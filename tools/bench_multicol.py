"""Multi-column apply (n_rhs columns in one pass over the elements) against column-by-column launches of the single-column
kernel (l3k_tuning::column_by_column): Diffusion3D, order 6 / 4.   python tools/bench_multicol.py [--ne 48] [--cols 3]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from l3ster_amd import system

ap = argparse.ArgumentParser()
ap.add_argument("--ne", type=int, default=48)
ap.add_argument("--order", type=int, default=6)
ap.add_argument("--cols", type=int, default=3)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
torch.cuda.set_device(0)
ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
U = 4
part = system.CubePartition(a.ne, a.order, perturb=0.1)
mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, U, part.dirichlet_mask(U)), system.KERNEL_DIFFUSION3D, [1.0, 1.0], n_rhs=a.cols)
X = torch.as_tensor(part.synthetic_vector(U, ncols=a.cols), device="cuda")
out = {}
for mode in ("one pass", "column by column"):
    ctx.set_tuning(column_by_column=int(mode == "column by column"))
    Y = torch.zeros_like(X)
    for _ in range(5):
        mf.apply(X, Y, 1.0, 0.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        mf.apply(X, Y, 1.0, 0.0)
    e1.record()
    torch.cuda.synchronize()
    out[mode] = (e0.elapsed_time(e1) / a.steps, Y.clone())
ctx.set_tuning(column_by_column=0)
d = (out["one pass"][1] - out["column by column"][1]).norm().item() / out["column by column"][1].norm().item()
print(json.dumps({"order": a.order, "ne": a.ne, "cols": a.cols, "ms_one_pass": out["one pass"][0], "ms_column_by_column": out["column by column"][0],
                  "rel_diff": d, "dof_per_s_one_pass": part.n_global_nodes * U * a.cols / (out["one pass"][0] * 1e-3)}))
