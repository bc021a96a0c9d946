"""The per-rank work of the partitioned apply, without a network: one rank's block of a px x py x pz partition (default:
rank 0... of 2x2x2 over 128^3, i.e. what every GPU does in `bench.py --gpus 8`) runs the full schedule of
l3ster_amd.distributed.DistributedOperator with a transport that moves no data (ghost values stay zero).  Shows what the
split into first half / border / second half, the pack / unpack kernels and the extra launches cost next to the one-GPU
apply of the same 64^3 block.   python tools/bench_rank_schedule.py [--ne 128] [--parts 2 2 2] [--rank 0]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("L3K_GENERIC_BELOW", "0")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from l3ster_amd import system  # noqa: E402
from l3ster_amd.distributed import DistributedOperator, HaloPlan  # noqa: E402


class NullTransport:
    def post(self, sends, recvs):
        return []

    @staticmethod
    def wait(reqs):
        pass


ap = argparse.ArgumentParser()
ap.add_argument("--ne", type=int, default=128)
ap.add_argument("--order", type=int, default=6)
ap.add_argument("--parts", type=int, nargs=3, default=[2, 2, 2])
ap.add_argument("--rank", type=int, default=7)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
U = 4
ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
part = system.CubePartition(a.ne, a.order, parts=tuple(a.parts), rank=a.rank)
mesh = system.DeviceMesh(ctx, part, U, part.dirichlet_mask(U))
mf = system.MatrixFreeSystem(mesh, system.KERNEL_DIFFUSION3D, [1.0, 1.0])
op = DistributedOperator(mf, HaloPlan(part, U, "cuda"), transport=NullTransport())
X = system.synthetic_vector_torch(part.node_grid_id[:part.n_owned_nodes], U, "cuda")
Y = torch.zeros_like(X)
for _ in range(5):
    op.apply(X, Y, 1.0, 0.0)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
ev[0].record()
for i in range(a.steps):
    op.apply(X, Y, 1.0, 0.0)
    ev[i + 1].record()
torch.cuda.synchronize()
ms = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(a.steps)])
owned_dofs = part.n_owned_nodes * U
print(json.dumps({"workload": f"rank {a.rank} of {a.parts} over {a.ne}^3, order {a.order}: full partitioned-apply schedule, no network",
                  "elements": int(part.n_elems), "interior_elements": int(part.n_interior_elems), "owned_dofs": int(owned_dofs),
                  "ghost_nodes": int(part.n_ghost_nodes), "neighbours": len(part.nbr_rank),
                  "ms_per_apply_median": round(float(np.median(ms)), 3), "owned_dof_per_s": float(owned_dofs / np.median(ms) * 1e3)}))
