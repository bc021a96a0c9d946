"""computeDiagAndRhs on the device (l3k_mf_diag_rhs) against one apply, 64^3 hexes: the right-hand side on the single-wave kernel's RHS
variant (default) and on the generic kernel in RHS mode (l3k_tuning::generic_below = huge), the diagonal kernel in both.

    python tools/r04_diag_rhs.py > profiles/r04_diag_rhs.jsonl
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from l3ster_amd import system  # noqa: E402

torch.cuda.set_device(0)
ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)


def t(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for kid, p, kpar in ((system.KERNEL_DIFFUSION3D, 6, [1.0, 1.0]), (system.KERNEL_DIFFUSION3D, 4, [1.0, 1.0]), (system.KERNEL_ADVDIFF3D, 4, None)):
    info = system.kernel_info(kid)
    U, F = info["n_unknowns"], info["n_fields"]
    part = system.CubePartition(64, p, perturb=0.1)
    mask = part.dirichlet_mask(U, unknowns=[0])
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, U, mask), kid, kpar)
    if F:
        mf.set_fields(system.synthetic_vector_torch(part.node_grid_id, F, "cuda", seed=7).view(-1, F).t().contiguous())
    g = system.synthetic_vector_torch(part.node_grid_id, U, "cuda") * torch.as_tensor(mask, device="cuda")[None, :]
    X = system.synthetic_vector_torch(part.node_grid_id, U, "cuda")
    Y = torch.zeros_like(X)
    out = {"kernel": info["name"], "order": p, "ne": 64, "dofs": part.n_global_nodes * U, "apply_ms": t(lambda: mf.apply(X, Y))}
    res = {}
    for name, below in (("single_wave_rhs", 0), ("generic_rhs", 10 ** 9)):
        with ctx.tuning(generic_below=below):
            out[f"diag_rhs_ms_{name}"] = t(lambda: mf.diag_rhs(g), 3)
            res[name] = mf.diag_rhs(g)[1].clone()
    free = torch.as_tensor(mask == 0, device="cuda")
    a, b = res["single_wave_rhs"][0][free], res["generic_rhs"][0][free]
    out["rhs_rel_diff_on_free_dofs"] = float((a - b).norm() / b.norm())
    out["diag_rhs_over_apply"] = out["diag_rhs_ms_single_wave_rhs"] / out["apply_ms"]
    print(json.dumps(out), flush=True)
    del mf
