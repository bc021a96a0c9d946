"""The C-ABI halo (l3k_halo_*, l3k_mf_apply_dist: RCCL neighbour exchange inside the library) on ONE GPU: a one-rank cube
made periodic in x through the ghost machinery, so that import and export are ncclSend / ncclRecv of the rank to itself in
one group call -- the only neighbour exchange a single GPU can carry through RCCL.  Checked against the oracle on the
periodic mesh (connectivity with the identification carried out), against the Python-side schedule with a loop-back
transport, and for import / export as building blocks.  Run as a script (own process: it creates an RCCL communicator):

    python tests/rccl_native_halo_periodic.py [--ne 4 3 3] [--order 6] [--bench N]
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]

import oracle_lib as O  # noqa: E402
from helpers import PeriodicXPartition, rel_err  # noqa: E402
from l3ster_amd import system  # noqa: E402
from l3ster_amd.distributed import DistributedOperator, HaloPlan, NativeDistributedOperator, NativeHalo  # noqa: E402


class LoopbackTransport:
    """every message goes from the rank to itself"""

    def post(self, sends, recvs):
        torch.cuda.synchronize()
        for (_, s), (_, r) in zip(sends, recvs):
            r.copy_(s)
        return []

    def wait(self, reqs):
        torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ne", type=int, nargs=3, default=[4, 3, 3])
    ap.add_argument("--order", type=int, default=6)
    ap.add_argument("--bench", type=int, default=0, help="timed applies (no oracle beyond 3000 elements)")
    a = ap.parse_args()
    os.environ["L3K_GENERIC_BELOW"] = "0"
    torch.cuda.set_device(0)
    U, p, kid, kpar = 4, a.order, system.KERNEL_DIFFUSION3D, [0.7, 1.0]
    part = PeriodicXPartition(system.CubePartition(tuple(a.ne), p, perturb=0.1))
    side_bits = 0b001111  # Dirichlet on the z and y sides (unknown 0); x is periodic
    mask = np.zeros((part.n_local_nodes, U), np.uint8)
    mask[(part.node_boundary & side_bits) != 0, 0] = 1
    mask = mask.reshape(-1)
    ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
    mesh = system.DeviceMesh(ctx, part, U, mask)
    mf = system.MatrixFreeSystem(mesh, kid, kpar, n_rhs=2)
    halo = NativeHalo(ctx, part, U)  # world of one rank, the neighbour list names the rank itself
    op = NativeDistributedOperator(mf, halo)
    n_owned = part.n_owned_nodes * U
    for ncols in (1, 2):
        x = system.synthetic_vector_torch(part.node_grid_id[:part.n_owned_nodes], U, "cuda", seed=5, ncols=ncols)
        y0 = torch.as_tensor(np.random.default_rng(2).uniform(-1, 1, (ncols, n_owned)), device="cuda")
        y = y0.clone()
        for _ in range(2):  # twice: buffers are reused
            y.copy_(y0)
            op.apply(x, y, 1.25, -0.5)
        torch.cuda.synchronize()
        ref_op = DistributedOperator(mf, HaloPlan(part, U, "cuda"), transport=LoopbackTransport())
        y2 = y0.clone()
        ref_op.apply(x, y2, 1.25, -0.5)
        torch.cuda.synchronize()
        assert rel_err(y.cpu().numpy(), y2.cpu().numpy()) < 1e-13, "native schedule differs from the Python schedule"
        if part.n_elems <= 3000:
            om = O.MeshView(3, p, p + 1, part.merged, part.elem_verts, part.n_owned_nodes, U, np.arange(U), mask[:n_owned])
            y_ref = O.mf_apply(om, kid, x.cpu().numpy().T, np.asfortranarray(y0.cpu().numpy().T.copy()), alpha=1.25, beta=-0.5,
                               kparams=kpar, nthreads=8)
            err = rel_err(y.cpu().numpy().T, y_ref)
            assert err < 1e-11, err
    # the building blocks: import copies the partners' rows into the ghost rows, export adds them back
    v = torch.as_tensor(np.random.default_rng(3).uniform(-1, 1, (2, n_owned)), device="cuda")
    ghost = op.import_ghosts(v)
    torch.cuda.synchronize()
    rows = (part.send_nodes[0].astype(np.int64)[:, None] * U + np.arange(U)[None, :]).reshape(-1)
    assert torch.equal(ghost[:, :len(rows)], v[:, rows])
    owned = v.clone()
    op.export_add(ghost, owned)
    torch.cuda.synchronize()
    want = v.clone()
    want[:, rows] += ghost[:, :len(rows)]
    assert torch.equal(owned, want)
    print(f"native halo ok: {part.n_elems} elements of order {p}, {part.n_ghost_nodes} ghost nodes exchanged with the rank itself "
          f"through RCCL, 1 and 2 columns", flush=True)
    if a.bench:
        x = system.synthetic_vector_torch(part.node_grid_id[:part.n_owned_nodes], U, "cuda", seed=5)
        y = torch.empty_like(x)
        for _ in range(3):
            op.apply(x, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.bench):
            op.apply(x, y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.bench
        print(f"partitioned apply with self exchange: {ms:.3f} ms per apply, {part.n_owned_nodes * U / ms * 1e3:.3e} dof/s", flush=True)


if __name__ == "__main__":
    main()
