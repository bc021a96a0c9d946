"""Randomised parity sweep of the multi-rank schedule (DistributedOperator: pack -> import || interior -> border -> export ->
unpack-add -> Dirichlet rows; ranks are threads of this process on one GPU, messages through queues): random block partitions
(1..3 blocks per direction, up to 8 ranks, uneven block sizes), orders 1..4, random Dirichlet sides, against the CPU oracle on
the whole mesh.    python tools/fuzz_partitioned.py [--seconds 120] [--seed 0]"""
import argparse, os, queue, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle_lib as O
from helpers import oracle_mesh
from l3ster_amd import system
from l3ster_amd.distributed import DistributedOperator, HaloPlan, InprocGroup, NativeDistributedOperator, NativeHalo


class ThreadTransport:
    def __init__(self, rank, boxes):
        self.rank, self.boxes = rank, boxes

    def post(self, sends, recvs):
        torch.cuda.synchronize()
        for peer, t in sends:
            self.boxes[(self.rank, peer)].put(t.clone())
        return recvs

    def wait(self, recvs):
        for peer, t in recvs:
            t.copy_(self.boxes[(peer, self.rank)].get(timeout=120))
        torch.cuda.synchronize()


ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--native", action="store_true", help="the C-ABI schedule l3k_mf_apply_dist with the library's in-process transport "
                                                        "(round 3) instead of the Python-side schedule")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
torch.cuda.set_device(0)
dev = lambda v: torch.as_tensor(np.ascontiguousarray(v), dtype=torch.float64, device="cuda")
U, kid = 4, system.KERNEL_DIFFUSION3D
t_end, n, worst = time.time() + a.seconds, 0, (0.0, None)
t_progress = time.time() + 60.0
while time.time() < t_end:
    while True:
        parts = tuple(int(v) for v in rng.integers(1, 4, 3))
        if 2 <= int(np.prod(parts)) <= 8:
            break
    world = int(np.prod(parts))
    p = int(rng.integers(1, 5))
    ne = tuple(int(q + rng.integers(0, 4)) for q in parts)  # at least one element per block, uneven splits
    sides = [s for s in range(6) if rng.random() < 0.5]
    alpha, beta = float(rng.uniform(-2, 2)), float(rng.choice([0.0, rng.uniform(-1, 1)]))
    os.environ["L3K_GENERIC_BELOW"] = str(int(rng.choice([0, 1500])))
    boxes = {(i, j): queue.Queue() for i in range(world) for j in range(world)}
    group = InprocGroup(world) if a.native else None
    out, errors = {}, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            part = system.CubePartition(ne, p, parts, rank, perturb=0.1)
            mask = part.dirichlet_mask(U, sides=sides)
            c = system.Context(0, torch.cuda.current_stream().cuda_stream)
            mf = system.MatrixFreeSystem(system.DeviceMesh(c, part, U, mask), kid, [0.7, 1.0])
            n_owned = part.n_owned_nodes * U
            X = dev(part.synthetic_vector(U)[:, :n_owned])
            Y = dev(part.synthetic_vector(U, seed=7)[:, :n_owned])
            if a.native:
                with torch.cuda.stream(torch.cuda.Stream()):
                    c2 = system.Context(0, torch.cuda.current_stream().cuda_stream)
                    mf2 = system.MatrixFreeSystem(system.DeviceMesh(c2, part, U, mask), kid, [0.7, 1.0])
                    torch.cuda.current_stream().wait_stream(torch.cuda.default_stream())
                    NativeDistributedOperator(mf2, NativeHalo(c2, part, U, rank, world, transport=group)).apply(X, Y, alpha, beta)
                    torch.cuda.current_stream().synchronize()
            else:
                DistributedOperator(mf, HaloPlan(part, U, "cuda"), transport=ThreadTransport(rank, boxes)).apply(X, Y, alpha, beta)
            torch.cuda.synchronize()
            out[rank] = (Y.cpu().numpy(), part.node_grid_id[:part.n_owned_nodes].copy())
        except Exception as exc:
            errors.append((rank, repr(exc)))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    case = dict(parts=parts, ne=ne, p=p, sides=sides, alpha=alpha, beta=beta, generic_below=os.environ["L3K_GENERIC_BELOW"])
    if errors:
        print("FAIL (exception)", errors, case)
        sys.exit(1)
    whole = system.CubePartition(ne, p, perturb=0.1)
    mask = whole.dirichlet_mask(U, sides=sides)
    x, y0 = whole.synthetic_vector(U), whole.synthetic_vector(U, seed=7)
    y_ref = O.mf_apply(oracle_mesh(whole, p + 1, U, np.arange(U), mask), kid, x.T, np.asfortranarray(y0.T.copy()), alpha=alpha,
                       beta=beta, kparams=[0.7, 1.0], nthreads=8).reshape(whole.n_local_nodes, U)
    row_of = {int(g): i for i, g in enumerate(whole.node_grid_id)}
    num = den = 0.0
    seen = 0
    for r in range(world):
        y, gid = out[r]
        rows = np.array([row_of[int(g)] for g in gid], dtype=np.int64)
        num += float(np.sum((y.reshape(len(rows), U) - y_ref[rows]) ** 2))
        den += float(np.sum(y_ref[rows] ** 2))
        seen += len(rows)
    err = (num / max(den, 1e-300)) ** 0.5
    n += 1
    if time.time() > t_progress:  # (a long run must keep writing: the GPU box takes 7 silent minutes for a hang)
        print(f"... {n} cases so far, worst {worst[0]:.3e}", flush=True)
        t_progress = time.time() + 60.0
    if err > worst[0]:
        worst = (err, case)
    if seen != whole.n_local_nodes or not err < 1e-11:
        print("FAIL", err, seen, whole.n_local_nodes, case)
        sys.exit(1)
print(f"{n} cases, worst relative error {worst[0]:.3e} at {worst[1]}")
