"""BASELINE.json config 5: advection-diffusion-reaction in first-order least-squares form (unknowns c, q; velocity = 3
interpolated fields, the karman-style "kernel reads interpolated field values"), hex order 4, Jacobi-PCG driven by the
matrix-free apply.  Reports DOF/s per apply inside the solve and the iteration count.
    python tools/bench_config5.py [--ne 64] [--order 4] [--tol 1e-6]                       # one GPU
    python tools/bench_config5.py --gpus N      # starts its N ranks itself; or, under a launcher:
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_config5.py --gpus N
(N in 1, 2, 4, 8: --ne elements per edge PER GPU, blocks 2x1x1 / 2x2x1 / 2x2x2, RCCL neighbour exchange + 2 scalar
all-reduces per iteration)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from l3ster_amd import solve, system  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ne", type=int, default=64)
ap.add_argument("--order", type=int, default=4)
ap.add_argument("--tol", type=float, default=1e-6)
ap.add_argument("--check-every", type=int, default=10)
ap.add_argument("--gpus", type=int, default=1)
a = ap.parse_args()
import torch.distributed as dist  # noqa: E402
from l3ster_amd.distributed import DistributedOperator, HaloPlan, HostStagedTransport  # noqa: E402
from l3ster_amd import launch  # noqa: E402
PARTS = {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}
if a.gpus not in PARTS:
    raise SystemExit("--gpus must be 1, 2, 4 or 8")
if launch.needs_self_launch(a.gpus):  # typed without a launcher: this process starts the ranks and stays off the GPU
    launch.self_launch(__file__, sys.argv[1:], a.gpus)
    raise SystemExit(0)
world, rank, local_rank = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
if world != a.gpus:
    raise SystemExit(f"--gpus {a.gpus} but the launcher set WORLD_SIZE={world}")
# L3K_BENCH_REHEARSAL=1: all ranks on GPU 0 with gloo and host-staged messages (the N > 1 code path on a one-GPU box; the
# numbers of such a run mean nothing)
rehearsal = os.environ.get("L3K_BENCH_REHEARSAL") == "1"
if rehearsal:
    local_rank = 0
torch.cuda.set_device(local_rank)
use_dist = world > 1 or os.environ.get("L3K_FORCE_DIST") == "1"
if use_dist:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    if rehearsal:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
ctx = system.Context(local_rank, torch.cuda.current_stream().cuda_stream)
p, U, kid = a.order, 4, system.KERNEL_ADVDIFF3D
parts = PARTS[world]
part = system.CubePartition(tuple(a.ne * q for q in parts), p, parts, rank, perturb=0.1)
mask = part.dirichlet_mask(U)
mesh = system.DeviceMesh(ctx, part, U, mask)
mf = system.MatrixFreeSystem(mesh, kid, [1.0, 0.5, 1.0])  # k, sigma, s
# smooth analytic velocity sampled at the nodes' reference grid position (synthetic field data, SoA [3][n_nodes])
Nx = p * a.ne * parts[0] + 1  # (the partitions used here have equal edge counts where it matters: x is split first)
Ny, Nz = p * a.ne * parts[1] + 1, p * a.ne * parts[2] + 1
gid = torch.as_tensor(part.node_grid_id, device="cuda")
gx, gy, gz = (gid % Nx).double() / (Nx - 1), ((gid // Nx) % Ny).double() / (Ny - 1), (gid // (Nx * Ny)).double() / (Nz - 1)
fields = torch.stack([0.5 * torch.sin(np.pi * gy), 0.25 * torch.cos(np.pi * gx), 0.1 * gz]).contiguous()
mf.set_fields(fields)
op = DistributedOperator(mf, HaloPlan(part, U, "cuda"), transport=HostStagedTransport() if rehearsal else None) if use_dist else None


def host_allreduce(view):  # (rehearsal: gloo reduces host tensors)
    t = view.cpu()
    dist.all_reduce(t)
    view.copy_(t)

diag, rhs = op.diag_rhs(None) if use_dist else mf.diag_rhs(None)  # homogeneous Dirichlet c = 0
minv = solve.jacobi_inverse_native(ctx, diag)
x = torch.zeros_like(diag)
torch.cuda.synchronize()
if use_dist:
    dist.barrier()
t0 = time.perf_counter()
if use_dist:  # fused l3k_cg_* kernels + neighbour exchange in the apply + 2 scalar all-reduces per iteration
    res = solve.pcg_distributed(op, ctx, rhs[0], x, minv, tol=a.tol, residual_scaling="rhs", max_iters=5000,
                                check_every=a.check_every, allreduce=host_allreduce if rehearsal else None)
else:  # l3k_pcg_solve: apply + fused vector kernels + reductions behind the C ABI (one 32-byte readback per check)
    res = solve.pcg(mf, rhs[0], x, minv, tol=a.tol, residual_scaling="rhs", max_iters=5000, check_every=a.check_every)
torch.cuda.synchronize()
if use_dist:
    dist.barrier()
dt = time.perf_counter() - t0
n_apply = [res.num_iters + 1]
dofs = part.n_global_nodes * U
# apply alone, same operator
y = torch.empty_like(x)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    (op if use_dist else mf).apply(x[None, :], y[None, :])
e1.record()
torch.cuda.synchronize()
ms_apply = e0.elapsed_time(e1) / 10
if rank == 0:
    print(json.dumps({"n_gpus": world, **({"rehearsal": True} if rehearsal else {}), "config": f"advection-diffusion 3D (F=3 fields), hex {a.ne}^3 per GPU x {parts}, order {p}, Jacobi-PCG rel tol {a.tol}",
                      "dofs": dofs, "iterations": res.num_iters, "achieved_tol": res.tol, "solve_s": dt, "applies": n_apply[0],
                      "dof_per_s_inside_solve": dofs * n_apply[0] / dt, "ms_per_apply_alone": ms_apply,
                      "dof_per_s_apply_alone": dofs / (ms_apply * 1e-3)}))
if use_dist:
    dist.destroy_process_group()
