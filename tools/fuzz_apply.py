"""Randomised parity sweep of the matrix-free apply against the CPU oracle: random mesh extents (1..8 elements per edge,
non-cubic), orders 1..8, 1..3 columns, random alpha / beta, random Dirichlet sides and unknowns, perturbed or uniform
geometry, diag + lifted rhs on a third of the single-column cases, fast (single-wave) and generic kernel routes (L3K_GENERIC_BELOW drawn per case), deterministic mode on a
fraction of the cases.  Prints the worst relative error; exits non-zero on a case above 1e-11.
    python tools/fuzz_apply.py [--seconds 120] [--seed 0]"""
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
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
torch.cuda.set_device(0)
dev = lambda v: torch.as_tensor(np.ascontiguousarray(v), dtype=torch.float64, device="cuda")
t_end, n, worst = time.time() + a.seconds, 0, (0.0, None)
t_progress = time.time() + 60.0
U, kid = 4, system.KERNEL_DIFFUSION3D
while time.time() < t_end:
    p = int(rng.integers(1, 9))
    cap = {1: 8, 2: 8, 3: 7, 4: 6, 5: 5, 6: 5, 7: 3, 8: 3}[p]
    ne = tuple(int(v) for v in rng.integers(1, cap + 1, 3))
    ncols = int(rng.integers(1, 4))
    perturb = float(rng.choice([0.0, 0.1, 0.2]))
    alpha, beta = float(rng.uniform(-2, 2)), float(rng.choice([0.0, 1.0, rng.uniform(-1, 1)]))
    sides = [s for s in range(6) if rng.random() < 0.5]
    unknowns = [u for u in range(U) if rng.random() < 0.4] or [0]
    det = rng.random() < 0.25
    os.environ["L3K_GENERIC_BELOW"] = str(int(rng.choice([0, 1500])))
    part = system.CubePartition(ne, p, perturb=perturb)
    mask = part.dirichlet_mask(U, unknowns=unknowns, sides=sides)
    ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
    if det:
        ctx.set_deterministic(True)
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, U, mask), kid, [0.7, 1.0], n_rhs=ncols)
    x = part.synthetic_vector(U, seed=int(rng.integers(1 << 30)), ncols=ncols)
    y0 = rng.uniform(-1, 1, x.shape)
    y_ref = O.mf_apply(oracle_mesh(part, p + 1, U, np.arange(U), mask), kid, x.T, np.asfortranarray(y0.T.copy()), alpha=alpha,
                       beta=beta, kparams=[0.7, 1.0], nthreads=8)
    Y = dev(y0)
    mf.apply(dev(x), Y, alpha, beta)
    torch.cuda.synchronize()
    err = rel_err(Y.cpu().numpy().T, y_ref)
    case = dict(p=p, ne=ne, ncols=ncols, perturb=perturb, alpha=alpha, beta=beta, sides=sides, unknowns=unknowns, det=det,
                generic_below=os.environ["L3K_GENERIC_BELOW"])
    if ncols == 1 and rng.random() < 0.3:  # diag + lifted rhs of the same system (MatrixFreeSystem::computeDiagAndRhs)
        g = rng.uniform(-1, 1, (1, part.n_local_nodes * U)) * mask[None, :]
        diag, rhs = mf.diag_rhs(dev(g))
        torch.cuda.synchronize()
        d_ref, r_ref = O.mf_diag_rhs(oracle_mesh(part, p + 1, U, np.arange(U), mask), kid, 1, np.asfortranarray(g.T), kparams=[0.7, 1.0])
        err = max(err, rel_err(diag.cpu().numpy(), d_ref), rel_err(rhs.cpu().numpy().T, r_ref))
    n += 1
    if time.time() > t_progress:  # (a long run must keep writing: the GPU box takes 7 silent minutes for a hang)
        print(f"... {n} cases so far, worst {worst[0]:.3e}", flush=True)
        t_progress = time.time() + 60.0
    if err > worst[0]:
        worst = (err, case)
    if not err < 1e-11:
        print("FAIL", err, case)
        sys.exit(1)
print(f"{n} cases, worst relative error {worst[0]:.3e} at {worst[1]}")
