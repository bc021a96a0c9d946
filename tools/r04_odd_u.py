"""Kernels with an odd number of unknowns on the one-wave-per-element kernel: element kernel alone, fast route
against the generic LDS kernel, for the scalar advection kernel (U = 1, E = 1, F = 3), the div-curl kernel (U = 3, E = 4) and, as
the yardstick, Diffusion3D (U = 4, E = 7), at orders 4 and 6.  Prints one JSON line per (kernel, order): routes, ms, dof/s, the
fraction of the HBM roofline by SURVEY 8(d)'s algorithmic bytes (16 + (4 npe + 192) / (p^3 U) + 8 F / U per dof), fast / generic.

    python tools/r04_odd_u.py [--ne 64] [--steps 10] > profiles/r04_odd_u.log
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from l3ster_amd import system  # noqa: E402

HBM_PEAK = 8.0e12


def bytes_per_dof(p, U, F):
    return 16.0 + (4.0 * (p + 1) ** 3 + 192.0) / (p ** 3 * U) + 8.0 * F / U


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ne", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--orders", default="4,6")
    a = ap.parse_args()
    torch.cuda.set_device(0)
    ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
    kernels = [("advection3d", system.KERNEL_ADVECTION3D, [0.05]), ("divcurl3d", system.KERNEL_DIVCURL3D, [0.6]),
               ("diffusion3d", system.KERNEL_DIFFUSION3D, [1.0, 1.0])]
    for p in [int(s) for s in a.orders.split(",")]:
        part = system.CubePartition(a.ne, p, perturb=0.1)
        for name, kid, kpar in kernels:
            info = system.kernel_info(kid)
            U, F = info["n_unknowns"], info["n_fields"]
            mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, U, part.dirichlet_mask(U)), kid, kpar)
            if F:
                mf.set_fields(system.synthetic_vector_torch(part.node_grid_id, F, "cuda", seed=7).view(-1, F).t().contiguous())
            X = system.synthetic_vector_torch(part.node_grid_id, U, "cuda")
            Y = torch.zeros_like(X)
            out = {"kernel": name, "order": p, "ne": a.ne, "U": U, "F": F, "dofs": part.n_global_nodes * U}
            ys = {}
            for route, below in (("fast", 0), ("generic", 10 ** 9)):
                with ctx.tuning(generic_below=below):
                    out[route + "_route"] = mf.route()
                    ms = time_apply(mf, X, Y, a.steps if route == "fast" else max(2, a.steps // 3))
                    Y.zero_()
                    mf.apply_elems(2, X, None, Y, None, 1.0, 0.0)
                    ys[route] = Y.clone()
                out[route + "_ms"] = ms
                out[route + "_dof_per_s"] = out["dofs"] / (ms * 1e-3)
            bpd = bytes_per_dof(p, U, F)
            out["bytes_per_dof"] = bpd
            out["fast_hbm_frac"] = out["fast_dof_per_s"] * bpd / HBM_PEAK
            out["fast_over_generic"] = out["generic_ms"] / out["fast_ms"]
            out["fast_vs_generic_rel_diff"] = float((ys["fast"] - ys["generic"]).norm() / ys["generic"].norm())
            out["ns_per_element_fast"] = out["fast_ms"] * 1e6 / part.n_elems
            print(json.dumps(out), flush=True)
            del mf, X, Y, ys


if __name__ == "__main__":
    main()
