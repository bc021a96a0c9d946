"""The C-ABI distributed schedule (l3k_mf_apply_dist, l3k_halo_import / _export_add: csrc/api_halo.hip) with MORE THAN ONE
RANK on a one-GPU box: the ranks are threads of this process, each with its own context and stream, and the exchange goes
through the library's in-process transport (l3k_inproc_group_*, the second implementation of the l3k_halo_transport table
whose default is RCCL) or through a transport table written in Python.  Against the oracle on the whole mesh.

Reference: comm/ImportExport.hpp:295-372,402-470 and the schedule of MatrixFreeSystem::applyImpl
(algsys/MatrixFreeSystem.hpp:1020-1140); tests/MpiImportExportTest.cpp:17-136 runs the exchange with np in {1, 2, 4} and 20
repetitions; tests/EmptyPartitionTest.cpp runs an apply with ranks that own nothing."""
import os
import sys
import threading

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import oracle_lib as O  # noqa: E402
from helpers import oracle_mesh, rel_err  # noqa: E402

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

U, KID, KPAR = 4, 0, [0.7, 1.0]


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")


def run_ranks(world, body):
    """body(rank) in one thread per rank; re-raises the first failure"""
    errors = []

    def guarded(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                body(rank)
                torch.cuda.synchronize()
        except BaseException as exc:  # noqa: BLE001
            errors.append((rank, exc))

    threads = [threading.Thread(target=guarded, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not errors, errors


def check_against_whole(out, whole, y_ref, ncols):
    row_of = {int(g): i for i, g in enumerate(whole.node_grid_id)}
    for r, (y, gid) in out.items():
        if len(gid) == 0:
            continue
        rows = np.array([row_of[int(g)] for g in gid])
        ref = y_ref.reshape(whole.n_local_nodes, U, ncols)[rows]
        got = y.reshape(ncols, len(rows), U).transpose(1, 2, 0)
        assert np.linalg.norm(got - ref) < 1e-11 * np.linalg.norm(ref), r


@pytest.mark.parametrize("ne,p,parts,ncols", [((4, 2, 2), 2, (2, 1, 1), 2), ((4, 4, 2), 4, (2, 2, 1), 1),
                                              ((4, 4, 4), 6, (2, 2, 2), 1), ((8, 2, 2), 3, (4, 1, 1), 1)])
def test_mf_apply_dist_thread_ranks_inproc_transport(ne, p, parts, ncols):
    """l3k_mf_apply_dist of every rank of a block partition, 20 repetitions (the import / export buffers and events are
    reused), against the oracle on the whole mesh; l3k_halo_import / _export_add as building blocks on the way."""
    from l3ster_amd import system
    from l3ster_amd.distributed import InprocGroup, NativeDistributedOperator, NativeHalo
    world = int(np.prod(parts))
    group = InprocGroup(world)
    out, exported = {}, {}

    def body(rank):
        part = system.CubePartition(ne, p, parts, rank, perturb=0.1)
        mask = part.dirichlet_mask(U)
        c = system.Context(0, torch.cuda.current_stream().cuda_stream)
        mesh = system.DeviceMesh(c, part, U, mask)
        mf = system.MatrixFreeSystem(mesh, KID, KPAR, n_rhs=ncols)
        n_owned = part.n_owned_nodes * U
        X = dev(part.synthetic_vector(U, ncols=ncols)[:, :n_owned])
        Y = dev(part.synthetic_vector(U, seed=7, ncols=ncols)[:, :n_owned])
        op = NativeDistributedOperator(mf, NativeHalo(c, part, U, rank, world, transport=group))
        # the two exchanges alone: ghosts <- owners (values are a function of the global node: check them), owners += ghosts
        ghosts = op.import_ghosts(X)
        torch.cuda.current_stream().synchronize()
        want = part.synthetic_vector(U, ncols=ncols)[:, n_owned:]
        assert np.array_equal(ghosts.cpu().numpy()[:, :want.shape[1]], want)
        # comm::Export alone: every ghost row of value 1 lands in exactly one owner's row (summed over the ranks below)
        acc = torch.zeros_like(X)
        op.export_add(torch.ones_like(ghosts), acc)
        torch.cuda.current_stream().synchronize()
        exported[rank] = (float(acc.sum()), part.n_ghost_nodes * U * ncols)
        for _ in range(20):
            Yc = Y.clone()
            op.apply(X, Yc, 1.25, -0.5)
        torch.cuda.current_stream().synchronize()
        out[rank] = (Yc.cpu().numpy(), part.node_grid_id[:part.n_owned_nodes].copy())

    run_ranks(world, body)
    assert sum(v[0] for v in exported.values()) == sum(v[1] for v in exported.values()) > 0
    whole = system.CubePartition(ne, p, perturb=0.1)
    mask = whole.dirichlet_mask(U)
    x, y0 = whole.synthetic_vector(U, ncols=ncols), whole.synthetic_vector(U, seed=7, ncols=ncols)
    y_ref = O.mf_apply(oracle_mesh(whole, p + 1, U, np.arange(U), mask), KID, x.T, np.asfortranarray(y0.T.copy()),
                       alpha=1.25, beta=-0.5, kparams=KPAR, nthreads=4)
    check_against_whole(out, whole, y_ref, ncols)


def test_mf_apply_dist_with_a_rank_that_owns_nothing():
    """tests/EmptyPartitionTest.cpp: an unstructured mesh cut in two parts of a world of three -- rank 1 owns no element and no
    node, takes part in the schedule and returns; the other two agree with the one-rank apply of the library."""
    from test_partition_unstructured import part_vector, small_mesh
    from l3ster_amd import partition, system
    from l3ster_amd.distributed import InprocGroup, NativeDistributedOperator, NativeHalo
    p, world = 2, 3
    en, ev, n_nonint = small_mesh(p, n_keep=120)
    pv = part_vector(ev, 2) * 2
    group = InprocGroup(world)
    out = {}

    def xvec(mesh, seed):
        g = mesh.node_grid_id[:mesh.n_local_nodes].astype(np.float64)
        return np.stack([np.sin(0.37 * g + u + seed) for u in range(U)], axis=1).reshape(1, -1)

    def body(rank):
        mesh_h = partition.PartitionedMesh(en, ev, n_nonint, pv, rank, world, p)
        mask = np.zeros(mesh_h.n_local_nodes * U, np.uint8)
        mask[(mesh_h.node_grid_id[:mesh_h.n_local_nodes] % 11 == 0).repeat(U) & (np.arange(mesh_h.n_local_nodes * U) % U == 0)] = 1
        c = system.Context(0, torch.cuda.current_stream().cuda_stream)
        mesh = system.DeviceMesh(c, mesh_h, U, mask)
        mf = system.MatrixFreeSystem(mesh, KID, KPAR)
        n_owned = mesh_h.n_owned_nodes * U
        X, Y = dev(xvec(mesh_h, 0)[:, :n_owned]), dev(xvec(mesh_h, 5)[:, :n_owned])
        op = NativeDistributedOperator(mf, NativeHalo(c, mesh_h, U, rank, world, transport=group))
        for _ in range(3):
            Yc = Y.clone()
            op.apply(X, Yc, 0.5, 2.0)
        torch.cuda.current_stream().synchronize()
        out[rank] = (Yc.cpu().numpy(), mesh_h.node_grid_id[:mesh_h.n_owned_nodes].copy())

    run_ranks(world, body)
    assert out[1][0].size == 0
    # the one-rank apply of the same mesh through the library (itself pinned against the oracle elsewhere)
    whole = partition.PartitionedMesh(en, ev, n_nonint, np.zeros_like(pv), 0, 1, p)
    mask = np.zeros(whole.n_local_nodes * U, np.uint8)
    mask[(whole.node_grid_id[:whole.n_local_nodes] % 11 == 0).repeat(U) & (np.arange(whole.n_local_nodes * U) % U == 0)] = 1
    torch.cuda.set_device(0)
    c = system.Context(0, torch.cuda.current_stream().cuda_stream)
    mf = system.MatrixFreeSystem(system.DeviceMesh(c, whole, U, mask), KID, KPAR)
    Yw = dev(xvec(whole, 5))
    mf.apply(dev(xvec(whole, 0)), Yw, 0.5, 2.0)
    torch.cuda.synchronize()
    yw = Yw.cpu().numpy().reshape(-1, U)
    row_of = {int(g): i for i, g in enumerate(whole.node_grid_id[:whole.n_local_nodes])}
    for r in (0, 2):
        y, gid = out[r]
        ref = yw[[row_of[int(g)] for g in gid]]
        assert np.linalg.norm(y.reshape(-1, U) - ref) < 1e-11 * np.linalg.norm(ref)
    om = oracle_mesh(whole, p + 1, U, np.arange(U), mask)
    y_ref = O.mf_apply(om, KID, xvec(whole, 0).T, np.asfortranarray(xvec(whole, 5).T.copy()), alpha=0.5, beta=2.0, kparams=KPAR)
    assert np.linalg.norm(yw.reshape(-1) - y_ref.reshape(-1)) < 1e-11 * np.linalg.norm(y_ref)


def test_transport_table_from_the_host():
    """A transport table written by the host (here in Python, through ctypes callbacks): what a maintainer of the reference
    would bring to keep MPI with device pointers.  Two ranks, messages through a dictionary of device tensors."""
    import ctypes as C
    import queue
    from l3ster_amd import capi, system
    from l3ster_amd.distributed import NativeDistributedOperator, NativeHalo
    ne, p, parts, world = (4, 2, 2), 2, (2, 1, 1), 2
    boxes = {(a, b): queue.Queue() for a in range(world) for b in range(world)}
    out = {}

    def make_table(rank):
        pend = []

        def begin(_u):
            pend.clear()
            return 0

        def send(_u, buf, n, peer, stream):
            # (the payload is complete once the communication stream has run up to here)
            s = torch.cuda.ExternalStream(stream)
            s.synchronize()
            t = torch.empty(n, dtype=torch.float64, device="cuda")
            capi_memcpy(t.data_ptr(), buf, 8 * n)
            boxes[(rank, peer)].put(t)
            return 0

        def recv(_u, buf, n, peer, stream):
            pend.append((buf, n, peer))
            return 0

        def end(_u, stream):
            for buf, n, peer in pend:
                t = boxes[(peer, rank)].get(timeout=120)
                assert t.numel() == n
                capi_memcpy(buf, t.data_ptr(), 8 * n)
            return 0

        T = capi.HaloTransport
        tab = T(None, T.GROUP_BEGIN(begin), T.SEND(send), T.RECV(recv), T.GROUP_END(end), T.DESTROY())
        return tab

    hip = C.CDLL("libamdhip64.so")

    def capi_memcpy(dst, src, nbytes):
        assert hip.hipMemcpy(C.c_void_p(dst), C.c_void_p(src), C.c_size_t(nbytes), 3) == 0  # hipMemcpyDeviceToDevice, synchronous

    def body(rank):
        part = system.CubePartition(ne, p, parts, rank, perturb=0.1)
        mask = part.dirichlet_mask(U)
        c = system.Context(0, torch.cuda.current_stream().cuda_stream)
        mf = system.MatrixFreeSystem(system.DeviceMesh(c, part, U, mask), KID, KPAR)
        n_owned = part.n_owned_nodes * U
        X = dev(part.synthetic_vector(U)[:, :n_owned])
        Y = dev(part.synthetic_vector(U, seed=7)[:, :n_owned])
        op = NativeDistributedOperator(mf, NativeHalo(c, part, U, rank, world, transport=make_table(rank)))
        op.apply(X, Y, 1.25, -0.5)
        torch.cuda.current_stream().synchronize()
        out[rank] = (Y.cpu().numpy(), part.node_grid_id[:part.n_owned_nodes].copy())

    run_ranks(world, body)
    whole = system.CubePartition(ne, p, perturb=0.1)
    mask = whole.dirichlet_mask(U)
    x, y0 = whole.synthetic_vector(U), whole.synthetic_vector(U, seed=7)
    y_ref = O.mf_apply(oracle_mesh(whole, p + 1, U, np.arange(U), mask), KID, x.T, np.asfortranarray(y0.T.copy()), alpha=1.25,
                       beta=-0.5, kparams=KPAR)
    check_against_whole(out, whole, y_ref, 1)


def test_halo_of_another_context_is_rejected():
    from l3ster_amd import capi, system
    from l3ster_amd.distributed import InprocGroup, NativeDistributedOperator, NativeHalo
    torch.cuda.set_device(0)
    part = system.CubePartition(2, 2)
    c1 = system.Context(0, torch.cuda.current_stream().cuda_stream)
    c2 = system.Context(0, torch.cuda.current_stream().cuda_stream)
    mf = system.MatrixFreeSystem(system.DeviceMesh(c1, part, U, part.dirichlet_mask(U)), KID, KPAR)
    op = NativeDistributedOperator(mf, NativeHalo(c2, part, U, 0, 1, transport=InprocGroup(1)))
    X = dev(part.synthetic_vector(U))
    with pytest.raises(capi.L3KError, match="different contexts"):
        op.apply(X, torch.zeros_like(X))


def test_failed_send_closes_the_transport_group():
    """a failed send / recv after group_begin returned without group_end, leaving e.g. an RCCL group open.  A host-written
    table (one rank exchanging with itself: a cube made periodic in x) whose send fails on its second call: the error reaches the
    caller, group_end was called for the group that had been begun, and after the transport recovers the same halo works again."""
    import ctypes as C
    from helpers import PeriodicXPartition
    from l3ster_amd import capi, system
    from l3ster_amd.distributed import NativeDistributedOperator, NativeHalo
    torch.cuda.set_device(0)
    c = system.Context(0, torch.cuda.current_stream().cuda_stream)
    part = PeriodicXPartition(system.CubePartition((4, 2, 2), 2, perturb=0.1))
    calls = dict(begin=0, end=0, send=0, fail_at=2)
    pend, inbox = [], []
    hip = C.CDLL("libamdhip64.so")

    def begin(_u):
        calls["begin"] += 1
        pend.clear()
        return 0

    def send(_u, buf, n, peer, stream):
        calls["send"] += 1
        if calls["send"] == calls["fail_at"]:
            return -3
        torch.cuda.ExternalStream(stream).synchronize()
        t = torch.empty(n, dtype=torch.float64, device="cuda")
        assert hip.hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(buf), C.c_size_t(8 * n), 3) == 0
        inbox.append(t)
        return 0

    def recv(_u, buf, n, peer, stream):
        pend.append((buf, n))
        return 0

    def end(_u, stream):
        calls["end"] += 1
        if len(inbox) == len(pend):  # (a complete group: deliver; an abandoned one: drop what was posted)
            for (buf, n), t in zip(pend, inbox):
                assert hip.hipMemcpy(C.c_void_p(buf), C.c_void_p(t.data_ptr()), C.c_size_t(8 * n), 3) == 0
        inbox.clear()
        return 0

    T = capi.HaloTransport
    tab = T(None, T.GROUP_BEGIN(begin), T.SEND(send), T.RECV(recv), T.GROUP_END(end), T.DESTROY())
    mask = np.zeros(part.n_local_nodes * U, np.uint8)
    n_owned = part.n_owned_nodes * U
    X = dev(np.random.default_rng(0).uniform(-1, 1, (2, n_owned)))  # two columns: two sends per group
    Y = torch.zeros_like(X)
    mf2 = system.MatrixFreeSystem(system.DeviceMesh(c, part, U, mask), KID, KPAR, n_rhs=2)
    op2 = NativeDistributedOperator(mf2, NativeHalo(c, part, U, 0, 1, transport=tab))
    with pytest.raises(capi.L3KError):
        op2.apply(X, Y)
    assert calls["begin"] == calls["end"] == 1 and calls["send"] == 2
    calls["fail_at"] = -1
    Y = torch.zeros_like(X)
    op2.apply(X, Y)
    torch.cuda.current_stream().synchronize()
    assert calls["begin"] == calls["end"] == 3  # import + export of the good apply
    om = O.MeshView(3, 2, 3, part.merged, part.elem_verts, part.n_owned_nodes, U, np.arange(U), None)
    y_ref = O.mf_apply(om, KID, X.cpu().numpy().T, kparams=KPAR)
    assert rel_err(Y.cpu().numpy().T, y_ref) < 1e-11
