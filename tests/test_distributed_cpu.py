"""world_size-2/4 gloo tests of the multi-rank operator: the partition, the exchange plan and the schedule of
l3ster_amd.distributed.DistributedOperator are the product's; the local element work is done here by the CPU oracle
(test infrastructure) through the same backend interface the HIP path implements.  The assembled result must equal the
single-rank operator on the whole mesh (tests/MpiImportExportTest.cpp, Diffusion2D*Test at np in {1,2,4} in the
reference: same problem on 1, 2, 4 ranks)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from helpers import oracle_mesh
from l3ster_amd import system
from l3ster_amd.distributed import DistributedOperator, HaloPlan, HostStagedTransport


class OracleBackend:
    """CPU stand-in for l3ster_amd.system.MatrixFreeSystem's split-phase interface (test infrastructure)."""

    def __init__(self, part, kid, nq, dpn, mask, kparams=None, fields=None):
        self.part, self.kid, self.kparams = part, kid, kparams
        self.mesh = oracle_mesh(part, nq, dpn, np.arange(system.kernel_info(kid)["n_unknowns"]), mask, fields)
        self.n_owned = part.n_owned_nodes * dpn
        self.mask_owned = torch.as_tensor(mask[:self.n_owned].astype(bool))

    def scale(self, Y, beta):
        if beta == 0.0:
            Y.zero_()
        else:
            Y.mul_(beta)

    def pack_rows(self, X, idx, dst):
        dst.copy_(X[:, idx.long()])

    def unpack_add_rows(self, src, idx, Y):
        Y[:, idx.long()] += src

    def apply_elems(self, which, X, XG, Y, YG, alpha, beta):  # beta already applied to every row by scale()
        ng = self.mesh.n_local_dofs - self.n_owned
        nc = X.shape[0]
        xg = XG[:, :ng] if XG is not None else torch.zeros((nc, ng), dtype=torch.float64)
        yg = YG[:, :ng] if YG is not None else torch.zeros((nc, ng), dtype=torch.float64)
        xl = np.asfortranarray(torch.cat([X, xg], dim=1).numpy().T)
        yl = np.asfortranarray(torch.cat([Y, yg], dim=1).numpy().T.copy())
        ni = self.part.n_interior_elems
        e0, e1 = {0: (0, ni), 1: (ni, self.part.n_elems), 2: (0, self.part.n_elems), 3: (0, ni // 2), 4: (ni // 2, ni)}[which]
        O.mf_apply(self.mesh, self.kid, xl, yl, alpha=alpha, kparams=self.kparams, e_begin=e0, e_end=e1, do_scale=False,
                   do_dirichlet_rows=False)
        Y.copy_(torch.as_tensor(yl[:self.n_owned].T.copy()))
        if YG is not None:
            YG[:, :ng].copy_(torch.as_tensor(yl[self.n_owned:].T.copy()))
        else:
            assert np.all(yl[self.n_owned:] == 0.0)  # interior elements never touch ghosts

    def dirichlet_rows(self, X, Y, alpha):
        Y[:, self.mask_owned] += alpha * X[:, self.mask_owned]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ne, p, parts, kid, ncols, out_dir, staged=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        info = system.kernel_info(kid)
        U, F = info["n_unknowns"], info["n_fields"]
        part = system.CubePartition(ne, p, parts, rank, perturb=0.1)
        mask = part.dirichlet_mask(U)
        fields = None
        if F:
            gid = part.node_grid_id.astype(np.float64)
            fields = np.stack([np.sin(0.37 * gid + f) for f in range(F)])
        be = OracleBackend(part, kid, p + 1, U, mask, fields=fields)
        op = DistributedOperator(be, HaloPlan(part, U, "cpu"), transport=HostStagedTransport() if staged else None)
        n_owned = part.n_owned_nodes * U
        x_all = part.synthetic_vector(U, ncols=ncols)
        X = torch.as_tensor(x_all[:, :n_owned].copy())
        Y = torch.as_tensor(part.synthetic_vector(U, seed=7, ncols=ncols)[:, :n_owned].copy())
        for _ in range(2):  # twice: buffers are reused (races / stale state, cf. MpiImportExportTest's repetitions)
            Yc = Y.clone()
            op.apply(X, Yc, 1.25, -0.5)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), y=Yc.numpy(), gid=part.node_grid_id[:part.n_owned_nodes])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ne,p,parts,kid,ncols,staged", [
    ((4, 2, 2), 2, (2, 1, 1), system.KERNEL_DIFFUSION3D, 1, False),
    ((2, 4, 3), 3, (1, 2, 1), system.KERNEL_DIFFUSION3D, 2, False),
    ((4, 4, 2), 2, (2, 2, 1), system.KERNEL_ADVDIFF3D, 1, False),
    ((4, 2, 2), 2, (2, 1, 1), system.KERNEL_DIFFUSION3D, 2, True),  # the rehearsal transport of bench.py (host-staged messages)
])
def test_distributed_apply_equals_single_rank(tmp_path, ne, p, parts, kid, ncols, staged):
    world = int(np.prod(parts))
    mp.spawn(_worker, args=(world, _free_port(), ne, p, parts, kid, ncols, str(tmp_path), staged), nprocs=world, join=True)
    info = system.kernel_info(kid)
    U, F = info["n_unknowns"], info["n_fields"]
    whole = system.CubePartition(ne, p, perturb=0.1)
    mask = whole.dirichlet_mask(U)
    fields = None
    if F:
        gid = whole.node_grid_id.astype(np.float64)
        fields = np.stack([np.sin(0.37 * gid + f) for f in range(F)])
    x = whole.synthetic_vector(U, ncols=ncols)
    y0 = whole.synthetic_vector(U, seed=7, ncols=ncols)
    y_ref = O.mf_apply(oracle_mesh(whole, p + 1, U, np.arange(U), mask, fields), kid, x.T,
                       np.asfortranarray(y0.T.copy()), alpha=1.25, beta=-0.5)
    row_of = {int(g): i for i, g in enumerate(whole.node_grid_id)}
    seen = 0
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        rows = np.array([row_of[int(g)] for g in d["gid"]])
        ref = y_ref.reshape(whole.n_local_nodes, U, ncols)[rows]          # [node, u, col]
        got = d["y"].reshape(ncols, len(rows), U).transpose(1, 2, 0)
        assert np.linalg.norm(got - ref) < 1e-12 * np.linalg.norm(ref)
        seen += len(rows)
    assert seen == whole.n_local_nodes
