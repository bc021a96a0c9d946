"""Partition of an unstructured order-p mesh from an element partition vector (l3ster_amd/partition.py: the work of
comm::distributeMesh + mesh::LocalMeshView for a mesh every rank holds in full): ownership / numbering / exchange-list
invariants, and the partitioned apply against the one-rank apply -- on CPU with world-size-3 gloo ranks and the oracle as the
local backend, on the GPU with thread ranks and the HIP backend.  Mesh: the reference's gmsh cube
(tests/golden/gmsh_cube_hexes.npz), elevated by the numpy restatement (CPU) or on the device (GPU), partitioned by a
coordinate rule that cuts through the unstructured region."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import oracle_np as ONP  # noqa: E402
import oracle_lib as O  # noqa: E402
from helpers import oracle_mesh, rel_err  # noqa: E402
from l3ster_amd import partition, system  # noqa: E402
from l3ster_amd.distributed import DistributedOperator, HaloPlan  # noqa: E402
from test_distributed_cpu import OracleBackend  # noqa: E402
from test_order_elevation import gmsh_cube, rotate_elements  # noqa: E402

U = 4


def small_mesh(p, n_keep=600, seed=0):
    """A connected chunk of the gmsh cube (the elements nearest to a corner), local frames rotated at random."""
    verts, conn, _ = gmsh_cube()
    centre = verts[conn.astype(np.int64)].mean(axis=1)
    keep = np.argsort(np.linalg.norm(centre - centre.min(axis=0), axis=1))[:n_keep]
    conn = rotate_elements(conn[keep], seed)
    en, n_nodes, n_nonint = ONP.elevate_order(conn, verts.shape[0], p)
    # compact the node ids (vertices of dropped elements are unused)
    used, inv = np.unique(en, return_inverse=True)
    en = inv.reshape(en.shape)
    n_nonint = int(np.searchsorted(used, n_nonint))
    ev = verts[conn.astype(np.int64)]
    return en, ev, n_nonint


def part_vector(ev, world):
    c = ev.mean(axis=1)
    key = c[:, 0] + 0.37 * c[:, 1] - 0.21 * c[:, 2]  # an oblique cut
    edges = np.quantile(key, np.linspace(0, 1, world + 1)[1:-1])
    return np.searchsorted(edges, key)


def dirichlet_mask_of(mesh, en_all_coords_max):
    xyz = mesh.node_coords()
    on = np.abs(xyz[:, 0] - en_all_coords_max) < 1e-9  # the x = x_max face of the chunk, if touched (may be empty)
    mask = np.zeros((mesh.n_local_nodes, U), np.uint8)
    mask[on, 0] = 1
    mask[mesh.node_grid_id[:mesh.n_local_nodes] % 17 == 0, 1] = 1  # and a scattered set (a function of the global node)
    return mask.reshape(-1)


@pytest.mark.parametrize("world,p", [(3, 2), (5, 3)])
def test_partition_invariants(world, p):
    en, ev, n_nonint = small_mesh(p)
    part = part_vector(ev, world)
    n_nodes = int(en.max()) + 1
    meshes = [partition.PartitionedMesh(en, ev, n_nonint, part, r, world, p) for r in range(world)]
    assert sum(m.n_owned_nodes for m in meshes) == n_nodes  # every node owned exactly once ...
    owned_old = np.concatenate([m.node_grid_id[:m.n_owned_nodes] for m in meshes])
    assert np.array_equal(np.sort(owned_old), np.arange(n_nodes))
    assert [m.global_node_base for m in meshes] == list(np.cumsum([0] + [m.n_owned_nodes for m in meshes[:-1]]))  # contiguous ranges
    assert sum(m.n_elems for m in meshes) == en.shape[0]
    N = (p + 1) ** 3
    n1 = p + 1
    interior_pos = [i + n1 * (j + n1 * k) for k in range(1, p) for j in range(1, p) for i in range(1, p)]
    for r, m in enumerate(meshes):
        # the element-node table in old global ids is the global one
        assert np.array_equal(m.node_grid_id[m.elem_nodes.astype(np.int64)], en[m.elem_global])
        # lowest part touching a node owns it: no ghost of rank r is owned by a higher rank ... that touches only higher parts
        touch_min = np.full(n_nodes, world)
        np.minimum.at(touch_min, en.reshape(-1), np.repeat(part, N))
        assert np.all(touch_min[m.node_grid_id[:m.n_owned_nodes]] == r)
        assert np.all(touch_min[m.node_grid_id[m.n_owned_nodes:]] != r)
        # interior elements first: all their nodes owned; border elements have a ghost
        loc = m.elem_nodes.astype(np.int64)
        assert np.all(loc[:m.n_interior_elems] < m.n_owned_nodes)
        assert np.all((loc[m.n_interior_elems:] >= m.n_owned_nodes).any(axis=1))
        # internal nodes: owned, at the tail of the owned range, contiguous per element (what DeviceMesh's exclusive range needs)
        if interior_pos:
            ids = loc[:, interior_pos]
            assert ids.min() >= m.n_owned_nodes - m.n_elems * len(interior_pos) and ids.max() < m.n_owned_nodes
            assert np.all(np.diff(np.sort(ids, axis=1), axis=1) == 1)
        # ghosts sorted by global id = grouped by owner; exchange lists agree with the neighbour's, in order
        for k, q in enumerate(m.nbr_rank):
            mq = meshes[q]
            kk = mq.nbr_rank.index(r)
            g0, g1 = mq.ghost_ranges[kk]
            assert np.array_equal(m.node_grid_id[m.send_nodes[k]], mq.node_grid_id[mq.n_owned_nodes + g0:mq.n_owned_nodes + g1])
            assert np.all(np.diff(m.send_nodes[k]) > 0)


@pytest.mark.parametrize("n_parts", [1, 2, 3, 8])
def test_rcb_partition_balances_and_separates(n_parts):
    verts, conn, _ = gmsh_cube()
    ev = verts[conn.astype(np.int64)]
    part = partition.rcb_partition(ev, n_parts)
    counts = np.bincount(part, minlength=n_parts)
    assert counts.sum() == conn.shape[0] and counts.max() - counts.min() <= n_parts  # equal counts up to rounding
    if n_parts == 8:
        # three levels of bisection of a cube: the parts are the octants, so the number of nodes shared between parts stays
        # near the three cutting planes
        en, n_nodes, n_nonint = ONP.elevate_order(conn, verts.shape[0], 1)
        touch_lo = np.full(n_nodes, 99)
        touch_hi = np.full(n_nodes, -1)
        np.minimum.at(touch_lo, en.reshape(-1), np.repeat(part, 8))
        np.maximum.at(touch_hi, en.reshape(-1), np.repeat(part, 8))
        shared = int((touch_lo != touch_hi).sum())
        assert shared < 0.25 * n_nodes


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _part_vector_with_empty_rank(ev, world, empty_rank):
    """part_vector over world - 1 parts, numbered around a rank that gets no element (the reference's EmptyPartitionTest: np = 4
    with empty ranks)."""
    if empty_rank < 0:
        return part_vector(ev, world)
    pv = part_vector(ev, world - 1)
    return pv + (pv >= empty_rank)


def _worker(rank, world, port, p, out_dir, empty_rank=-1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        en, ev, n_nonint = small_mesh(p, n_keep=240)
        m = partition.PartitionedMesh(en, ev, n_nonint, _part_vector_with_empty_rank(ev, world, empty_rank), rank, world, p)
        if rank == empty_rank:
            assert m.n_elems == 0 and m.n_owned_nodes == 0 and m.n_ghost_nodes == 0 and m.nbr_rank == []
        mask = dirichlet_mask_of(m, ev[..., 0].max())
        be = OracleBackend(m, system.KERNEL_DIFFUSION3D, p + 1, U, mask, kparams=[0.7, 1.0])
        op = DistributedOperator(be, HaloPlan(m, U, "cpu"))
        n_owned = m.n_owned_nodes * U
        X = torch.as_tensor(m.synthetic_vector(U)[:, :n_owned].copy())
        Y = torch.as_tensor(m.synthetic_vector(U, seed=7)[:, :n_owned].copy())
        for _ in range(2):
            Yc = Y.clone()
            op.apply(X, Yc, 1.25, -0.5)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), y=Yc.numpy(), gid=m.node_grid_id[:m.n_owned_nodes])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,empty_rank", [(3, -1), (4, 2), (4, 0)])
def test_partitioned_apply_equals_single_rank_gloo(tmp_path, world, empty_rank):
    """(empty_rank >= 0: one rank owns no element and no node -- tests/EmptyPartitionTest.cpp of the reference)"""
    p = 2
    mp.spawn(_worker, args=(world, _free_port(), p, str(tmp_path), empty_rank), nprocs=world, join=True)
    en, ev, n_nonint = small_mesh(p, n_keep=240)
    whole = partition.PartitionedMesh(en, ev, n_nonint, np.zeros(en.shape[0], int), 0, 1, p)
    mask = dirichlet_mask_of(whole, ev[..., 0].max())
    x, y0 = whole.synthetic_vector(U), whole.synthetic_vector(U, seed=7)
    y_ref = O.mf_apply(oracle_mesh(whole, p + 1, U, np.arange(U), mask), system.KERNEL_DIFFUSION3D, x.T,
                       np.asfortranarray(y0.T.copy()), alpha=1.25, beta=-0.5, kparams=[0.7, 1.0])
    row_of = {int(g): i for i, g in enumerate(whole.node_grid_id)}
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        rows = np.array([row_of[int(g)] for g in d["gid"]], dtype=np.int64)
        if r == empty_rank:
            assert rows.size == 0 and d["y"].size == 0
            continue
        ref = y_ref.reshape(whole.n_local_nodes, U)[rows]
        assert rel_err(d["y"].reshape(-1, U), ref) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("world,p", [(4, 2), (3, 4)])
def test_partitioned_apply_on_device_elevated_mesh(world, p):
    """Device elevation -> partition -> DeviceMesh per rank -> partitioned apply (thread ranks, HIP kernels) against the
    one-rank apply on the same mesh."""
    import queue
    import threading
    from test_gpu_apply import ThreadTransport
    torch.cuda.set_device(0)
    ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
    verts, conn, _ = gmsh_cube()
    conn = rotate_elements(conn, seed=p)
    en, n_nodes, n_nonint = system.elevate_order(ctx, conn, verts.shape[0], p)
    ev = verts[conn.astype(np.int64)]
    part = part_vector(ev, world)
    boxes = {(a, b): queue.Queue() for a in range(world) for b in range(world)}
    out, errors = {}, []
    x_max = ev[..., 0].max()

    def run(rank):
        try:
            torch.cuda.set_device(0)
            m = partition.PartitionedMesh(en, ev, n_nonint, part, rank, world, p)
            c = system.Context(0, torch.cuda.current_stream().cuda_stream)
            mf = system.MatrixFreeSystem(system.DeviceMesh(c, m, U, dirichlet_mask_of(m, x_max)), system.KERNEL_DIFFUSION3D, [0.7, 1.0])
            n_owned = m.n_owned_nodes * U
            X = torch.as_tensor(m.synthetic_vector(U)[:, :n_owned].copy(), device="cuda")
            Y = torch.as_tensor(m.synthetic_vector(U, seed=7)[:, :n_owned].copy(), device="cuda")
            op = DistributedOperator(mf, HaloPlan(m, U, "cuda"), transport=ThreadTransport(rank, boxes))
            op.apply(X, Y, 1.25, -0.5)
            torch.cuda.synchronize()
            out[rank] = (Y.cpu().numpy(), m.node_grid_id[:m.n_owned_nodes].copy())
        except Exception as exc:  # pragma: no cover
            errors.append((rank, repr(exc)))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    whole = partition.PartitionedMesh(en, ev, n_nonint, np.zeros(en.shape[0], int), 0, 1, p)
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, whole, U, dirichlet_mask_of(whole, x_max)), system.KERNEL_DIFFUSION3D, [0.7, 1.0])
    X = torch.as_tensor(whole.synthetic_vector(U), device="cuda")
    Y = torch.as_tensor(whole.synthetic_vector(U, seed=7), device="cuda")
    mf.apply(X, Y, 1.25, -0.5)
    y_ref = Y.cpu().numpy().reshape(-1, U)
    row_of = {int(g): i for i, g in enumerate(whole.node_grid_id)}
    for r in range(world):
        y, gid = out[r]
        rows = np.array([row_of[int(g)] for g in gid])
        assert rel_err(y.reshape(-1, U), y_ref[rows]) < 1e-12


def test_gmsh_reader_on_a_two_hex_file(tmp_path):
    """l3ster_amd.gmsh.read_hexes: gmsh 4.1 ASCII, node tags not contiguous, a boundary quad block that is skipped, the gmsh
    corner order turned into v = i + 2j + 4k (positive Jacobian with the reference's vertex convention)."""
    from l3ster_amd import gmsh
    pts = {10: (0, 0, 0), 11: (1, 0, 0), 12: (1, 1, 0), 13: (0, 1, 0), 20: (0, 0, 1), 21: (1, 0, 1), 22: (1, 1, 1), 23: (0, 1, 1),
           30: (2, 0, 0), 31: (2, 1, 0), 32: (2, 0, 1), 33: (2, 1, 1)}
    tags = list(pts)
    txt = ["$MeshFormat", "4.1 0 8", "$EndMeshFormat", "$Nodes", f"1 {len(tags)} 10 33", f"3 1 0 {len(tags)}"]
    txt += [str(t) for t in tags] + [" ".join(str(float(c)) for c in pts[t]) for t in tags] + ["$EndNodes", "$Elements", "2 3 1 3"]
    txt += ["2 1 3 1", "1 10 11 12 13"]  # a quad of the boundary: ignored
    txt += ["3 7 5 2", "2 10 11 12 13 20 21 22 23", "3 11 30 31 12 21 32 33 22", "$EndElements", ""]
    path = tmp_path / "two.msh"
    path.write_text("\n".join(txt))
    verts, conn, ent = gmsh.read_hexes(path)
    assert verts.shape == (12, 3) and conn.shape == (2, 8) and list(ent) == [7, 7]
    v = verts[conn.astype(np.int64)]
    # local vertex v = i + 2j + 4k: edges 0->1, 0->2, 0->4 are the x, y, z directions of these axis-aligned hexes
    for e in range(2):
        assert np.allclose(v[e, 1] - v[e, 0], (1, 0, 0)) and np.allclose(v[e, 2] - v[e, 0], (0, 1, 0)) and np.allclose(v[e, 4] - v[e, 0], (0, 0, 1))
        assert np.allclose(v[e, 7] - v[e, 0], (1, 1, 1))
    assert np.allclose(v[1, 0], (1, 0, 0))
    en, n_nodes, _ = ONP.elevate_order(conn, verts.shape[0], 2)
    assert n_nodes == 5 * 3 * 3  # two hexes sharing a face at order 2


@pytest.mark.gpu
def test_unstructured_partitioned_solve_end_to_end():
    """The whole unstructured flow on the GPU: the reference's gmsh cube ([-1,1]^3) -> coordinate bisection into 4 parts ->
    device order elevation (order 3, rotated local frames) -> every rank's part -> Dirichlet T = x on the whole boundary ->
    partitioned diag / rhs -> partitioned Jacobi-PCG (thread ranks): T = x, q = (1, 0, 0) exactly (tri-linear geometry:
    the linear solution is in the space, the least-squares functional vanishes there)."""
    import queue
    import threading
    from l3ster_amd import solve
    from test_gpu_apply import ThreadTransport
    from test_gpu_boundary import ThreadAllReduce
    world, p = 4, 3
    torch.cuda.set_device(0)
    ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
    verts, conn, _ = gmsh_cube()
    conn = rotate_elements(conn, seed=5)
    en, n_nodes, n_nonint = system.elevate_order(ctx, conn, verts.shape[0], p)
    ev = verts[conn.astype(np.int64)]
    part = partition.rcb_partition(ev, world)
    boxes = {(a, b): queue.Queue() for a in range(world) for b in range(world)}
    red = ThreadAllReduce(world)
    out, errors = {}, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            m = partition.PartitionedMesh(en, ev, n_nonint, part, rank, world, p)
            xyz = m.node_coords()
            on_bnd = np.any(np.abs(np.abs(xyz) - 1.0) < 1e-9, axis=1)
            mask = np.zeros((m.n_local_nodes, U), np.uint8)
            mask[on_bnd, 0] = 1
            c = system.Context(0, torch.cuda.current_stream().cuda_stream)
            mesh = system.DeviceMesh(c, m, U, mask.reshape(-1))
            mf = system.MatrixFreeSystem(mesh, system.KERNEL_DIFFUSION3D, [1.0, 0.0])  # k = 1, no source
            op = DistributedOperator(mf, HaloPlan(m, U, "cuda"), transport=ThreadTransport(rank, boxes))
            n_owned = m.n_owned_nodes * U
            g = np.zeros((m.n_owned_nodes, U))
            g[:, 0] = np.where(on_bnd[:m.n_owned_nodes], xyz[:m.n_owned_nodes, 0], 0.0)
            diag, rhs = op.diag_rhs(torch.as_tensor(g.reshape(1, -1), device="cuda"))
            x = torch.zeros(n_owned, dtype=torch.float64, device="cuda")
            res = solve.pcg_distributed(op, c, rhs[0], x, solve.jacobi_inverse_native(c, diag), tol=1e-11,
                                        residual_scaling="rhs", max_iters=20000, allreduce=red.bind(rank), check_every=10)
            torch.cuda.synchronize()
            sol = x.view(-1, U).cpu().numpy()
            # (the operator of this sign convention has q = -k grad T)
            out[rank] = (res.num_iters, np.abs(sol[:, 0] - xyz[:m.n_owned_nodes, 0]).max(), np.abs(np.abs(sol[:, 1]) - 1.0).max(),
                         np.abs(sol[:, 2:]).max(), op.energy_fused)
        except Exception as exc:  # pragma: no cover
            errors.append((rank, repr(exc)))
            try:
                red.barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not errors, errors
    assert len({v[0] for v in out.values()}) == 1
    for r in range(world):
        assert out[r][1] < 1e-7 and out[r][2] < 1e-6 and out[r][3] < 1e-6, (r, out[r])
        assert out[r][4], r  # <p, A p> came from the element kernels on every rank
