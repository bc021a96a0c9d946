"""The reference's Diffusion3D benchmark program (benchmarks/Diffusion3D.hpp:27-141) end to end on one GPU:
mesh -> diag/rhs -> Jacobi-PCG driven by the matrix-free apply -> L2 residual norms, with wall times per phase.
    python tools/diffusion3d_benchmark.py [--ne 6] [--order 6] [--tol 1e-6]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from l3ster_amd import solve, system  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ne", type=int, default=6)
ap.add_argument("--order", type=int, default=6)
ap.add_argument("--tol", type=float, default=1e-6)
ap.add_argument("--check-every", type=int, default=10)
a = ap.parse_args()
torch.cuda.set_device(0)
ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
U, p = 4, a.order
t = {}


def timed(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    t[name] = time.perf_counter() - t0
    return out


part = timed("mesh_s", lambda: system.CubePartition(a.ne, p))
mesh = timed("upload_s", lambda: system.DeviceMesh(ctx, part, U, part.dirichlet_mask(U)))
mf = system.MatrixFreeSystem(mesh, system.KERNEL_DIFFUSION3D, [1.0, 1.0])
diag, rhs = timed("diag_rhs_s", lambda: mf.diag_rhs(None))
x = torch.zeros_like(diag)
minv = solve.jacobi_inverse_native(ctx, diag)
res = timed("solve_s", lambda: solve.pcg(mf, rhs[0], x, minv, tol=a.tol, residual_scaling="rhs", max_iters=100000,
                                         check_every=a.check_every))
fields = x.view(-1, U).T.contiguous()
err = timed("error_norm_s", lambda: system.norm_l2(mesh, system.RESIDUAL_DIFFUSION3D_ERROR, fields, kernel_params=[1.0, 1.0]))
print(json.dumps({"config": f"Diffusion3D benchmark, hex {a.ne}^3, order {p}, Jacobi-PCG rel tol {a.tol}",
                  "dofs": part.n_global_nodes * U, "iterations": res.num_iters, "achieved_tol": res.tol,
                  "l2_error_components": [float(e) for e in err], **{k: round(v, 4) for k, v in t.items()}}))
