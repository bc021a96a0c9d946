"""Times the plain apply and the apply that also delivers <x, A x> (l3k_mf_apply_energy: the PCG's apply), element kernel included,
back to back and interleaved with a bandwidth-bound kernel (as inside the PCG iteration)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from l3ster_amd import system
torch.cuda.set_device(0)
ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
U = 4
for p, ne in ((6, 64), (4, 64)):
    part = system.CubePartition(ne, p, perturb=0.1)
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, U, part.dirichlet_mask(U)), system.KERNEL_DIFFUSION3D, [1.0, 1.0])
    X = system.synthetic_vector_torch(part.node_grid_id, U, "cuda")
    Y = torch.empty_like(X)
    Z = torch.empty_like(X)
    s = torch.zeros(8, dtype=torch.float64, device="cuda")

    def timed(fn, n=20):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    t_plain = timed(lambda: mf.apply(X, Y, 1.0, 0.0))
    t_energy = timed(lambda: mf.apply_energy(X[0], Y[0], s))
    t_stream = timed(lambda: torch.add(X, Y, out=Z))

    def both():
        mf.apply_energy(X[0], Y[0], s)
        torch.add(X, Y, out=Z)
    t_both = timed(both)
    mf.apply(X, Y, 1.0, 0.0)
    ref = float((X[0] * Y[0]).sum())
    mf.apply_energy(X[0], Y[0], s)
    torch.cuda.synchronize()
    print(f"p={p}: apply {t_plain:.3f} ms, apply + <x,Ax> {t_energy:.3f} ms, a 3-vector streaming kernel {t_stream:.3f} ms, both in turn {t_both:.3f} ms "
          f"(sum {t_energy + t_stream:.3f}); <x,Ax> rel err {abs(s[1].item() - ref) / abs(ref):.2e}", flush=True)
