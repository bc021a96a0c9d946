"""Randomised parity sweep over the KERNELS of round 4 against the CPU oracle: Diffusion3D, Diffusion3DPoint (operators and rhs read
the point and the time), Advection3D (U = 1, F = 3), DivCurl3D (U = 3), AdvDiff3D (F = 3) at every instantiated (order, nq); random
mesh extents, geometry perturbation, Dirichlet sides / unknowns, alpha / beta, time, fields; the one-wave-per-element and the generic
route (l3k_tuning::generic_below drawn per case), the reference's z = 0 mode on a fraction of the cases (oracle with the same switch),
deterministic mode on a fraction; diag + lifted rhs on a third, K_e / F_e of a random element entry by entry on a tenth (orders <= 4);
on a third of the cases the kernel's unknowns are a random SUBSET of a wider dof map (dofs_per_node up to U + 3, random field_inds:
the strided-dof variant of the one-wave kernel or the generic kernel).
Prints the worst relative error; exits non-zero on a case above 1e-11.
    python tools/fuzz_kernels.py [--seconds 300] [--seed 0]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle_lib as O
from helpers import oracle_mesh, rel_err
from l3ster_amd import system

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=300.0)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
torch.cuda.set_device(0)
dev = lambda v: torch.as_tensor(np.ascontiguousarray(v), dtype=torch.float64, device="cuda")
KPAR = {system.KERNEL_DIFFUSION3D: [0.7, 1.0], system.KERNEL_DIFFUSION3D_POINT: [0.8, 1.2], system.KERNEL_ADVECTION3D: [0.05],
        system.KERNEL_DIVCURL3D: [0.6], system.KERNEL_ADVDIFF3D: [0.7, 1.3, 0.5]}
shapes = [(k, p, nq) for (k, p, nq, r) in system.instances() if r == 1 and k in KPAR]
t_end, n, worst, by_kernel = time.time() + a.seconds, 0, (0.0, None), {}
t_progress = time.time() + 60.0
while time.time() < t_end:
    kid, p, nq = shapes[int(rng.integers(len(shapes)))]
    info = system.kernel_info(kid)
    U, F = info["n_unknowns"], info["n_fields"]
    vo = next(v for v in (1, 2, 3) if system.n_qps1d(p, v) == nq)
    cap = {1: 8, 2: 7, 3: 6, 4: 5, 5: 4, 6: 4, 7: 3, 8: 3}[p]
    ne = tuple(int(v) for v in rng.integers(1, cap + 1, 3))
    perturb = float(rng.choice([0.0, 0.1, 0.2]))
    alpha, beta = float(rng.uniform(-2, 2)), float(rng.choice([0.0, 1.0, rng.uniform(-1, 1)]))
    sides = [s for s in range(6) if rng.random() < 0.5]
    unknowns = [u for u in range(U) if rng.random() < 0.4] or [0]
    det, z0, t = rng.random() < 0.2, rng.random() < 0.25, float(rng.choice([0.0, rng.uniform(-1, 1)]))
    below = int(rng.choice([0, 1500, 10 ** 9]))
    part = system.CubePartition(ne, p, perturb=perturb)
    subset = rng.random() < 0.33
    dpn = U + int(rng.integers(1, 4)) if subset else U
    fi = [int(v) for v in rng.permutation(dpn)[:U]] if subset else list(range(U))
    mask_u = part.dirichlet_mask(U, unknowns=unknowns, sides=sides).reshape(-1, U)
    mask = np.zeros((part.n_local_nodes, dpn), np.uint8)
    mask[:, fi] = mask_u
    if subset:  # (Dirichlet flags on dofs of other kernels must not matter)
        others = np.setdiff1d(np.arange(dpn), fi)
        mask[:, others] = rng.random((part.n_local_nodes, len(others))) < 0.2
    mask = mask.reshape(-1)
    ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
    ctx.set_tuning(generic_below=below)
    if det:
        ctx.set_deterministic(True)
    ctx.set_reference_z0(z0)
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, dpn, mask), kid, KPAR[kid], asm_opts=(vo, 0, 0), field_inds=fi if subset else None)
    fields = rng.uniform(-1, 1, (F, part.n_local_nodes)) if F else None
    if F:
        mf.set_fields(dev(fields))
    mf.set_time(t)
    om = oracle_mesh(part, nq, dpn, np.asarray(fi), mask, fields)
    x = part.synthetic_vector(dpn, seed=int(rng.integers(1 << 30)))
    y0 = rng.uniform(-1, 1, x.shape)
    O.set_reference_z0(z0)
    try:
        y_ref = O.mf_apply(om, kid, x.T, np.asfortranarray(y0.T.copy()), alpha=alpha, beta=beta, kparams=KPAR[kid], time=t, nthreads=8)
    finally:
        O.set_reference_z0(False)
    Y = dev(y0)
    mf.apply(dev(x), Y, alpha, beta)
    torch.cuda.synchronize()
    err = rel_err(Y.cpu().numpy().T, y_ref)
    case = dict(kernel=info["name"], p=p, nq=nq, ne=ne, perturb=perturb, alpha=alpha, beta=beta, sides=sides, unknowns=unknowns, det=det,
                z0=z0, time=t, generic_below=below, route=mf.route().split(":")[0], dpn=dpn, field_inds=fi)
    if rng.random() < 0.33:  # diag + lifted rhs (the reference's local-element path: true point in both modes)
        g = rng.uniform(-1, 1, (1, part.n_local_nodes * dpn)) * mask[None, :]
        diag, rhs = mf.diag_rhs(dev(g))
        torch.cuda.synchronize()
        d_ref, r_ref = O.mf_diag_rhs(om, kid, 1, np.asfortranarray(g.T), kparams=KPAR[kid], time=t)
        err = max(err, rel_err(diag.cpu().numpy(), d_ref), rel_err(rhs.cpu().numpy().T, r_ref))
        case["diag_rhs"] = True
    if p <= 4 and rng.random() < 0.1:
        e = int(rng.integers(part.n_elems))
        K, Fe, _ = mf.local_assemble(e, 1)
        nf = fields[:, part.elem_nodes[e]].T if F else None
        K_ref, F_ref = O.assemble_local(kid, p, nq, 1, part.elem_verts[e], nf, KPAR[kid], time=t)
        err = max(err, float(np.abs(K.cpu().numpy()[0] - K_ref).max() / np.abs(K_ref).max()),
                  float(np.abs(Fe.cpu().numpy()[0].T - F_ref).max() / max(1.0, np.abs(F_ref).max())))
        case["assemble"] = e
    n += 1
    by_kernel[info["name"]] = by_kernel.get(info["name"], 0) + 1
    if time.time() > t_progress:
        print(f"... {n} cases so far, worst {worst[0]:.3e}", flush=True)
        t_progress = time.time() + 60.0
    if err > worst[0]:
        worst = (err, case)
    if not err < 1e-11:
        print("FAIL", err, case)
        sys.exit(1)
print(f"{n} cases {by_kernel}, worst relative error {worst[0]:.3e} at {worst[1]}")
