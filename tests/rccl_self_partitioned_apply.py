"""Run by tests/test_gpu_apply.py::test_partitioned_apply_through_rccl_on_one_gpu in a child process.

The whole partitioned apply of 8 (2x2x2) logical ranks on ONE GPU with the messages carried by RCCL: the logical ranks
are threads of this process, every exchange phase of all of them is issued as ONE torch.distributed batch_isend_irecv on
the world-size-1 nccl process group, every message a send to and a receive from the own rank (RCCL matches the i-th send
to self with the i-th receive from self of a group and carries them as device copies).  So the buffers the product
hands to RCCL -- ghost slabs, packed rows, their shapes, strides and stream ordering against the pack / element /
unpack kernels -- go through the real library; only the xGMI hop between two GPUs is missing.  The assembled result
must equal the one-rank apply on the whole mesh."""
import os
import sys
import threading

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ.setdefault("MASTER_PORT", "29547")
os.environ["L3K_GENERIC_BELOW"] = "0"
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from l3ster_amd import system  # noqa: E402
from l3ster_amd.distributed import DistributedOperator, HaloPlan  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))


class RcclSelfExchange:
    """Shared by the logical ranks: collects one phase's messages of all of them, issues them as one RCCL group."""

    def __init__(self, world):
        self.world, self.lock, self.barrier = world, threading.Lock(), threading.Barrier(world)
        self.sends, self.recvs, self.reqs = {}, {}, []

    def post(self, rank, sends, recvs):
        with self.lock:
            for peer, t in sends:
                self.sends[(rank, peer)] = t
            for peer, t in recvs:
                self.recvs[(peer, rank)] = t
        if self.barrier.wait() == 0:
            assert sorted(self.sends) == sorted(self.recvs), (sorted(self.sends), sorted(self.recvs))
            keys = sorted(self.sends)
            for k in keys:  # what RCCL requires of the product's buffers
                assert self.sends[k].is_contiguous() and self.recvs[k].is_contiguous()
                assert self.sends[k].numel() == self.recvs[k].numel(), k
            ops = [dist.P2POp(dist.isend, self.sends[k], 0) for k in keys] + [dist.P2POp(dist.irecv, self.recvs[k], 0) for k in keys]
            self.reqs = dist.batch_isend_irecv(ops) if ops else []
            self.n_messages = getattr(self, "n_messages", 0) + len(keys)
            self.sends, self.recvs = {}, {}
        self.barrier.wait()

    def wait(self):
        if self.barrier.wait() == 0:
            for r in self.reqs:
                r.wait()
            self.reqs = []
        self.barrier.wait()


class RankTransport:
    def __init__(self, rank, exchange):
        self.rank, self.exchange = rank, exchange

    def post(self, sends, recvs):
        self.exchange.post(self.rank, sends, recvs)
        return []

    def wait(self, reqs):
        self.exchange.wait()


ne, p, parts, U = (4, 4, 4), 4, (2, 2, 2), 4
world = int(np.prod(parts))
exchange = RcclSelfExchange(world)
out, errors = {}, []


def run(rank):
    try:
        torch.cuda.set_device(0)
        part = system.CubePartition(ne, p, parts, rank, perturb=0.1)
        ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
        mesh = system.DeviceMesh(ctx, part, U, part.dirichlet_mask(U))
        mf = system.MatrixFreeSystem(mesh, system.KERNEL_DIFFUSION3D, [0.7, 1.0])
        n_owned = part.n_owned_nodes * U
        X = torch.as_tensor(part.synthetic_vector(U)[:, :n_owned], device="cuda")
        Y = torch.as_tensor(part.synthetic_vector(U, seed=7)[:, :n_owned], device="cuda")
        op = DistributedOperator(mf, HaloPlan(part, U, "cuda"), transport=RankTransport(rank, exchange))
        for _ in range(3):
            Yc = Y.clone()
            op.apply(X, Yc, 1.25, -0.5)
        torch.cuda.synchronize()
        out[rank] = (Yc.cpu().numpy(), part.node_grid_id[:part.n_owned_nodes].copy())
    except Exception as exc:  # pragma: no cover
        errors.append((rank, repr(exc)))
        exchange.barrier.abort()


threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
for t in threads:
    t.start()
for t in threads:
    t.join(240)
assert not errors, errors
whole = system.CubePartition(ne, p, perturb=0.1)
ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, whole, U, whole.dirichlet_mask(U)), system.KERNEL_DIFFUSION3D, [0.7, 1.0])
X = torch.as_tensor(whole.synthetic_vector(U), device="cuda")
Y = torch.as_tensor(whole.synthetic_vector(U, seed=7), device="cuda")
mf.apply(X, Y, 1.25, -0.5)
y_ref = Y.cpu().numpy().reshape(-1, U)
row_of = {int(g): i for i, g in enumerate(whole.node_grid_id)}
worst = 0.0
for r in range(world):
    y, gid = out[r]
    rows = np.array([row_of[int(g)] for g in gid])
    worst = max(worst, float(np.linalg.norm(y.reshape(-1, U) - y_ref[rows]) / np.linalg.norm(y_ref[rows])))
print(f"partitioned apply through RCCL: {world} logical ranks, {exchange.n_messages} messages, worst relative error {worst:.2e}")
print("RCCL partitioned apply", "ok" if worst < 1e-12 else "FAILED")
dist.destroy_process_group()
