"""Randomised sweep of the assembled path on the device: random small meshes (1..3 elements per direction, perturbed), orders
1..6, Diffusion3D / Mass3D, random Dirichlet sides.  Checked per case: (1) l3k_assemble_global (tiled layout, two streams, small
random workspaces: many sub-batches) gives the CSR values of l3k_local_assemble + l3k_assembled_scatter (row-major, both scatter
kernels) to rounding; (2) with skip_dirichlet the assembled operator applied to a random x equals the matrix-free apply on the
free dofs (the reference's cross-path property, tests/LocalOperatorTests.cpp:3-95, at mesh level); (3) the tiled element matrices
are the row-major ones; (4) the streaming mode's checksums (its two kernels: diagonal blocks by halves) are those of the stored
matrices, and the stored mode through the same two kernels gives the same, bitwise symmetric matrices.
    python tools/fuzz_assembled.py [--seconds 120] [--seed 0]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from l3ster_amd import system

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
torch.cuda.set_device(0)
ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)


def csr_graph(part, U):
    import scipy.sparse as sp
    dofs = (part.elem_nodes.astype(np.int64)[:, :, None] * U + np.arange(U)[None, None, :]).reshape(part.n_elems, -1)
    nd = dofs.shape[1]
    n = part.n_local_nodes * U
    G = sp.coo_matrix((np.ones(dofs.size * nd), (np.repeat(dofs, nd, axis=1).ravel(), np.tile(dofs, (1, nd)).ravel())), shape=(n, n)).tocsr()
    G.sort_indices()
    return torch.as_tensor(G.indptr.astype(np.int64), device="cuda"), torch.as_tensor(G.indices.astype(np.int32), device="cuda"), n


t_end, n_cases, worst = time.time() + a.seconds, 0, [0.0, 0.0, 0.0, 0.0]
t_progress = time.time() + 60.0
while time.time() < t_end:
    kid, kpar, U = [(system.KERNEL_DIFFUSION3D, [float(rng.uniform(0.5, 2)), 1.0], 4), (system.KERNEL_MASS3D, None, 2)][int(rng.integers(0, 2))]
    p = int(rng.choice([1, 2, 3, 4, 1, 2, 3, 4, 5, 6])) if kid == system.KERNEL_DIFFUSION3D else 2  # (the shapes instantiated in libl3k.so)
    ne = tuple(int(v) for v in rng.integers(1, 4 if p < 4 else 3, 3)) if p < 5 else tuple(int(v) for v in rng.permutation([1, 1, int(rng.integers(1, 3))]))
    sides = [s for s in range(6) if rng.random() < 0.4]
    part = system.CubePartition(ne, p, perturb=0.15)
    mask = part.dirichlet_mask(U, sides=sides) if U == 4 else np.zeros(part.n_local_nodes * U, np.uint8)
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, U, mask), kid, kpar)
    RP, CI, n = csr_graph(part, U)
    Nd = (p + 1) ** 3 * U
    case = dict(kid=kid, p=p, ne=ne, sides=sides)
    vals = {}
    for route in ("global", "rows", "per_entry"):
        for skip in (False, True):
            v = torch.zeros(CI.numel(), dtype=torch.float64, device="cuda")
            rhs = torch.zeros((1, n), dtype=torch.float64, device="cuda")
            if route == "global":
                ws = int(2 * 8 * (Nd * Nd + Nd + 10000) * rng.integers(1, 4))  # room for one to three elements per half
                miss = mf.assemble_global(RP, CI, v, rhs, skip_dirichlet=skip, workspace_bytes=ws)
            else:
                K, Fe, _ = mf.local_assemble()
                with ctx.tuning(scatter_per_entry=int(route == "per_entry")):
                    miss = mf.assembled_scatter(K, Fe, RP, CI, v, rhs, skip_dirichlet=skip)
            assert miss == 0, (miss, case)
            vals[(route, skip)] = (v, rhs)
    torch.cuda.synchronize()
    scale = float(vals[("rows", False)][0].abs().max())
    for skip in (False, True):
        for route in ("global", "per_entry"):
            e1 = float((vals[(route, skip)][0] - vals[("rows", skip)][0]).abs().max()) / scale
            e2 = float((vals[(route, skip)][1] - vals[("rows", skip)][1]).abs().max()) / max(1.0, float(vals[("rows", skip)][1].abs().max()))
            worst[0] = max(worst[0], e1, e2)
            if not (e1 < 1e-12 and e2 < 1e-12):
                print("FAIL routes differ", route, skip, e1, e2, case)
                sys.exit(1)
    # assembled == matrix-free on the free dofs
    x = torch.as_tensor(rng.standard_normal((1, n)), device="cuda")
    y = torch.zeros_like(x)
    mf.apply(x, y, 1.0, 0.0)
    A = torch.sparse_csr_tensor(RP, CI.to(torch.int64), vals[("global", True)][0], size=(n, n))
    want = (A @ x[0].unsqueeze(1)).squeeze(1) + torch.as_tensor(mask, device="cuda").double() * x[0]
    e3 = float((y[0] - want).norm() / want.norm())
    worst[1] = max(worst[1], e3)
    # tiled element matrices == row-major ones
    if U <= 4:
        K, _, _ = mf.local_assemble(want_F=False)
        Kt = mf.local_assemble_tiled()
        nn = p + 1
        K2 = Kt.permute(0, 4, 6, 5, 1, 8, 7, 3, 2).reshape(part.n_elems, Nd, Nd)
        e4 = float((K2 - K).abs().max() / K.abs().max())
        worst[2] = max(worst[2], e4)
    else:
        e4 = 0.0
    # the streaming mode (diagonal blocks by halves in merged iterations, diagonal / off-diagonal blocks as two kernels): its
    # checksums are those of the stored matrices; the stored mode through the same two kernels gives the same symmetric matrices
    K, _, _ = mf.local_assemble(want_F=False)
    _, _, cs = mf.local_assemble(want_K=False, want_F=False, want_checksum=True)
    with ctx.tuning(assemble_two_launches=1):
        K2l, _, _ = mf.local_assemble(want_F=False)
    ii = torch.arange(Nd, device="cuda")
    wgt = (1 + (ii[:, None] * 31 + ii[None, :] * 17) % 7).double()
    cs_ref, cs_abs = (K * wgt).sum(dim=(1, 2)), (K.abs() * wgt).sum(dim=(1, 2))
    e5 = float(((cs - cs_ref).abs() / cs_abs).max())
    e6 = float((K2l - K).abs().max() / K.abs().max())
    sym = bool((K2l == K2l.transpose(1, 2)).all())
    worst[3] = max(worst[3], e5, e6)
    if not (e3 < 1e-11 and e4 < 1e-12 and e5 < 1e-12 and e6 < 1e-12 and sym):
        print("FAIL", e3, e4, e5, e6, sym, case)
        sys.exit(1)
    n_cases += 1
    if time.time() > t_progress:
        print(f"... {n_cases} cases so far, worst {worst}", flush=True)
        t_progress = time.time() + 60.0
    del mf
print(f"{n_cases} cases; worst: routes differ by {worst[0]:.2e} (relative to |values|_max), assembled vs matrix-free {worst[1]:.2e}, "
      f"tiled vs row-major {worst[2]:.2e}, streaming checksums / two-kernel stored matrices vs the stored matrices {worst[3]:.2e}")
