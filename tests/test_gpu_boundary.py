"""GPU parity of the boundary-kernel terms and the post-processing integrals through the C ABI:
against the oracle on the same inputs, against the reference's side-area known answers, and end to end on 3-D twins of
the reference's K6 (exact linear solution with an adiabatic boundary kernel) and K7 (benchmarks/Diffusion3D.hpp)."""
import numpy as np
import pytest

import helpers
import oracle_lib as O

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    from l3ster_amd import system
    return system


@pytest.fixture(scope="module")
def ctx(S):
    torch.cuda.set_device(0)
    return S.Context(0, torch.cuda.current_stream().cuda_stream)


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")


HEXM = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0, 0, 1], [1, 0, 1.5], [0, 1, 1.5], [1, 1, 2]], float)


def test_side_areas_and_volume(S, ctx):
    # tests/MappingTests.cpp:555-610: quadrature order 10 (6 points) on the order-1 hex
    part = helpers.SingleElementMesh(1, HEXM)
    mesh = S.DeviceMesh(ctx, part, 1)
    opts = (5, 0, 0)
    assert S.n_qps1d(1, 5, 0) == 6
    for side, area in enumerate([1.0, np.sqrt(1.5), 1.25, 1.75, 1.25, 1.75]):
        got = S.integrate(mesh, S.RESIDUAL_UNIT3D, asm_opts=opts, face_elem=[0], face_side=[side])
        assert got[0] == pytest.approx(area, abs=1e-13)
    vol = S.integrate(mesh, S.RESIDUAL_UNIT3D, asm_opts=opts)
    want = O.integrate_local(-1, O.RESIDUAL_UNIT3D, 1, 6, HEXM, None)
    assert vol[0] == pytest.approx(want[0], rel=1e-13)


def _setup(S, ctx, ne, p, perturb=0.15, dir_sides=(0, 5)):
    part = S.CubePartition(ne, p, perturb=perturb)
    U = 4
    mask = part.dirichlet_mask(U, sides=dir_sides)
    mesh = S.DeviceMesh(ctx, part, U, mask)
    return part, mesh, mask, U


@pytest.mark.parametrize("kernel", ["robin", "normalflux", "robinpoint"])
@pytest.mark.parametrize("p,opts,R", [(2, (1, 0, 0), 1), (2, (1, 0, 0), 2), (4, (1, 0, 0), 1), (3, (2, 0, 0), 1)])
def test_boundary_term_vs_oracle(S, ctx, p, opts, R, kernel):
    """`normalflux` fills A1..A3 (n . grad T + c dq_x/dx + h T = g): the side kernel's derivative path, in which every node
    of the element takes part (algsys/EvaluateLocalOperator.hpp:238-274), against the oracle's dense side matrices."""
    part, mesh, mask, U = _setup(S, ctx, 3, p)
    nq = S.n_qps1d(p, *opts[:2])
    kp = [2.0, 0.7, 0.3] if kernel == "normalflux" else [2.0, 0.7]
    dev_kid, orc_kid = {"robin": (S.KERNEL_ROBIN3D, O.KERNEL_ROBIN3D), "normalflux": (S.KERNEL_NORMALFLUX3D, O.KERNEL_NORMALFLUX3D),
                        "robinpoint": (S.KERNEL_ROBINPOINT3D, O.KERNEL_ROBINPOINT3D)}[kernel]
    # (`robinpoint`: the coefficients read in.point.space and in.point.time -- l3k_bnd_set_time -- as the reference's wall kernels do)
    t = 0.65 if kernel == "robinpoint" else 0.0
    fe, fs = part.boundary_sides([1, 3, 4, 2] if kernel == "robin" else [0, 1, 2, 3, 4, 5])
    term = S.BoundaryTerm(mesh, dev_kid, fe, fs, kernel_params=kp, asm_opts=opts, n_rhs=R)
    term.set_time(t)
    om = helpers.oracle_mesh(part, nq, U, [0, 1, 2, 3], dirichlet=mask)
    rng = np.random.default_rng(p)
    x = rng.standard_normal((R, part.n_local_nodes * U))
    y0 = rng.standard_normal((R, part.n_local_nodes * U))
    Y = dev(y0)
    term.apply(dev(x), Y, alpha=1.5)
    want = np.asfortranarray(y0.T.copy())
    O.bnd_apply(om, orc_kid, fe, fs, np.asfortranarray(x.T), want, alpha=1.5, kparams=kp, time=t)
    assert helpers.rel_err(Y.cpu().numpy().T, want) < 1e-12
    # diag / rhs with Dirichlet lifting
    g = np.where(mask[None, :] != 0, rng.standard_normal((R, mask.size)), 0.0)
    diag = torch.zeros(mask.size, dtype=torch.float64, device="cuda")
    rhs = torch.zeros((R, mask.size), dtype=torch.float64, device="cuda")
    term.diag_rhs(diag, rhs, dirichlet_vals=dev(g))
    wd = np.zeros(mask.size)
    wr = np.zeros((mask.size, R), order="F")
    O.bnd_diag_rhs(om, orc_kid, fe, fs, wd, wr, dirichlet_vals=np.asfortranarray(g.T), kparams=kp, time=t)
    assert helpers.rel_err(diag.cpu().numpy(), wd) < 1e-12
    assert helpers.rel_err(rhs.cpu().numpy().T, wr) < 1e-11
    if kernel == "robinpoint":  # the time really enters: another time, another term
        Y2 = dev(y0)
        term.set_time(0.0)
        term.apply(dev(x), Y2, alpha=1.5)
        assert helpers.rel_err(Y2.cpu().numpy().T, want) > 1e-4


def test_attached_boundary_in_apply(S, ctx):
    p = 2
    part, mesh, mask, U = _setup(S, ctx, 4, p)
    nq = S.n_qps1d(p)
    fe, fs = part.boundary_sides([1, 2, 3, 4])
    mf = S.MatrixFreeSystem(mesh, S.KERNEL_DIFFUSION3D, [1.0, 1.0])
    term = S.BoundaryTerm(mesh, S.KERNEL_ADIABATIC3D, fe, fs)
    mf.attach_boundary(term)
    om = helpers.oracle_mesh(part, nq, U, [0, 1, 2, 3], dirichlet=mask)
    x = part.synthetic_vector(U, seed=3)
    y0 = part.synthetic_vector(U, seed=4)
    Y = dev(y0)
    mf.apply(dev(x), Y, alpha=0.5, beta=-2.0)
    want = O.mf_apply(om, O.KERNEL_DIFFUSION3D, x.T, y=np.asfortranarray(y0.T.copy()), alpha=0.5, beta=-2.0, kparams=[1.0, 1.0])
    # the boundary rows are added before the Dirichlet rows in the device schedule; both commute
    O.bnd_apply(om, O.KERNEL_ADIABATIC3D, fe, fs, np.asfortranarray(x.T), want, alpha=0.5)
    assert helpers.rel_err(Y.cpu().numpy().T, want) < 1e-12
    diag, rhs = mf.diag_rhs(None)
    wd, wr = O.mf_diag_rhs(om, O.KERNEL_DIFFUSION3D, kparams=[1.0, 1.0], finalize=False)
    O.bnd_diag_rhs(om, O.KERNEL_ADIABATIC3D, fe, fs, wd, wr)
    wd[mask != 0] = 1.0
    wr[mask != 0] = 0.0
    assert helpers.rel_err(diag.cpu().numpy(), wd) < 1e-12
    assert helpers.rel_err(rhs.cpu().numpy().T, wr) < 1e-11


def test_k6_twin_linear_solution_with_adiabatic_walls(S, ctx):
    """3-D twin of tests/Diffusion2D.hpp on the device: first-order diffusion system without source, Dirichlet T = x on
    the x- and x+ sides, adiabatic boundary kernel on the other four, distorted mesh.  T = x, q = (1,0,0) lies in the
    discrete space, so the least-squares solution reproduces it: L2 errors < 1e-8 on the domain and the boundary."""
    from l3ster_amd import solve
    p, ne, U = 2, 4, 4
    part = S.CubePartition(ne, p, perturb=0.1)
    mask = part.dirichlet_mask(U, sides=(4, 5))
    mesh = S.DeviceMesh(ctx, part, U, mask)
    mf = S.MatrixFreeSystem(mesh, S.KERNEL_DIFFUSION3D, [1.0, 0.0])  # k = 1, s = 0
    fe, fs = part.boundary_sides([0, 1, 2, 3])
    mf.attach_boundary(S.BoundaryTerm(mesh, S.KERNEL_ADIABATIC3D, fe, fs))
    coords = part.node_coords()
    # setDirichletBCValues: the boundary residual kernel out[0] = x at the nodes of the Dirichlet sides, on the device
    dfe, dfs = part.boundary_sides([4, 5])
    g = S.values_at_nodes(mesh, S.RESIDUAL_COORDX3D, [0], torch.zeros(part.n_local_nodes * U, dtype=torch.float64, device="cuda"),
                          face_elem=dfe, face_side=dfs)
    want = np.where(mask.reshape(-1, U) != 0, np.stack([coords[:, 0]] + [np.zeros(len(coords))] * 3, axis=1), 0.0)
    assert np.abs(g.cpu().numpy().reshape(-1, U) - want).max() < 1e-14
    diag, rhs = mf.diag_rhs(g[None, :])
    x = torch.zeros_like(diag)
    res = solve.cg(lambda v, out: mf.apply(v[None, :], out[None, :]), rhs[0], x, solve.jacobi_inverse(diag), tol=1e-12,
                   residual_scaling="rhs", max_iters=5000)
    assert res.converged
    sol = x.view(-1, U)
    assert (sol[:, 0] - dev(coords[:, 0])).abs().max().item() < 1e-9
    fields = sol.T.contiguous()  # SolutionManager layout: SoA [field][node]
    err = S.norm_l2(mesh, S.RESIDUAL_LINEAR3D_ERROR, fields)
    all_fe, all_fs = part.boundary_sides(range(6))
    berr = S.norm_l2(mesh, S.RESIDUAL_LINEAR3D_ERROR, fields, face_elem=all_fe, face_side=all_fs)
    assert np.linalg.norm(err) < 1e-8 and np.linalg.norm(berr) < 1e-8  # tests/Diffusion2D.hpp:117-119
    # oracle parity of the two norms on the same solution (values are ~1e-10: compare the squared integrals absolutely)
    om = O.MeshView(3, p, S.n_qps1d(p), part.elem_nodes, part.elem_verts, part.n_local_nodes, U, [0, 1, 2, 3],
                    fields=fields.cpu().numpy())
    nq2 = S.n_qps1d(p, 2, 0)
    assert np.allclose(err ** 2, O.mf_integrate(om, O.RESIDUAL_LINEAR3D_ERROR, nq2, square=True), atol=1e-18, rtol=1e-6)
    area = S.integrate(mesh, S.RESIDUAL_UNIT3D, asm_opts=(2, 0, 0), face_elem=all_fe, face_side=all_fs)
    assert area[0] == pytest.approx(O.mf_integrate(om, O.RESIDUAL_UNIT3D, nq2, face_elem=all_fe, face_side=all_fs)[0], rel=1e-13)


def test_k7_diffusion3d_benchmark_error_norms(S, ctx):
    """benchmarks/Diffusion3D.hpp:80-135 end to end on the device: 6^3 hexes of order 6 on [0,1]^3, T = 0 on the six
    sides, k = s = 1, Jacobi-PCG, then the four L2 residual components (the reference prints them without a threshold).
    Parity: the same four numbers from the oracle on the device solution; sanity: they are small."""
    from l3ster_amd import solve
    p, ne, U = 6, 6, 4
    part = S.CubePartition(ne, p)
    mask = part.dirichlet_mask(U)
    mesh = S.DeviceMesh(ctx, part, U, mask)
    mf = S.MatrixFreeSystem(mesh, S.KERNEL_DIFFUSION3D, [1.0, 1.0])
    diag, rhs = mf.diag_rhs(None)
    x = torch.zeros_like(diag)
    res = solve.cg(lambda v, out: mf.apply(v[None, :], out[None, :]), rhs[0], x, solve.jacobi_inverse(diag), tol=1e-10,
                   residual_scaling="rhs", max_iters=20000)
    assert res.converged
    fields = x.view(-1, U).T.contiguous()
    err = S.norm_l2(mesh, S.RESIDUAL_DIFFUSION3D_ERROR, fields, kernel_params=[1.0, 1.0])
    om = O.MeshView(3, p, S.n_qps1d(p), part.elem_nodes, part.elem_verts, part.n_local_nodes, U, [0, 1, 2, 3],
                    fields=fields.cpu().numpy())
    want = np.sqrt(O.mf_integrate(om, O.RESIDUAL_DIFFUSION3D_ERROR, S.n_qps1d(p, 2, 0), square=True, kparams=[1.0, 1.0]))
    assert np.allclose(err, want, rtol=1e-9, atol=0)
    assert np.all(err < 5e-2) and err[1:].max() < 1e-3
    # T is positive inside, zero on the boundary; the maximum of the Poisson solution on the unit cube is ~0.0562
    assert x.view(-1, U)[:, 0].max().item() == pytest.approx(0.0562, abs=2e-3)


@pytest.mark.parametrize("ne,parts", [((4, 2, 2), (2, 1, 1)), ((4, 4, 2), (2, 2, 1))])
def test_boundary_term_in_multi_rank_schedule(S, ctx, ne, parts):
    """A boundary term attached on every rank takes part in the split-phase schedule (sides of interior elements with the
    interior launch, sides of border elements after the import): the assembled result equals the oracle on the whole
    mesh.  Ranks are threads sharing the GPU (ThreadTransport of test_gpu_apply.py)."""
    import queue
    import threading
    from l3ster_amd.distributed import DistributedOperator, HaloPlan
    from test_gpu_apply import ThreadTransport
    p, U, kp = 2, 4, [2.0, 0.7]
    sides = [1, 3, 4, 5]
    world = int(np.prod(parts))
    boxes = {(a, b): queue.Queue() for a in range(world) for b in range(world)}
    out, errors = {}, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            part = S.CubePartition(ne, p, parts, rank, perturb=0.1)
            mask = part.dirichlet_mask(U, sides=(0, 2))
            c = S.Context(0, torch.cuda.current_stream().cuda_stream)
            mesh = S.DeviceMesh(c, part, U, mask)
            mf = S.MatrixFreeSystem(mesh, S.KERNEL_DIFFUSION3D, [0.7, 1.0])
            mf.attach_boundary(S.BoundaryTerm(mesh, S.KERNEL_ROBIN3D, *part.boundary_sides(sides), kernel_params=kp))
            n_owned = part.n_owned_nodes * U
            X = dev(part.synthetic_vector(U)[:, :n_owned])
            Y = dev(part.synthetic_vector(U, seed=7)[:, :n_owned])
            op = DistributedOperator(mf, HaloPlan(part, U, "cuda"), transport=ThreadTransport(rank, boxes))
            op.apply(X, Y, 1.25, -0.5)
            torch.cuda.synchronize()
            out[rank] = (Y.cpu().numpy(), part.node_grid_id[:part.n_owned_nodes].copy())
        except Exception as exc:  # pragma: no cover
            errors.append((rank, exc))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    whole = S.CubePartition(ne, p, perturb=0.1)
    mask = whole.dirichlet_mask(U, sides=(0, 2))
    om = helpers.oracle_mesh(whole, p + 1, U, np.arange(U), mask)
    x, y0 = whole.synthetic_vector(U), whole.synthetic_vector(U, seed=7)
    y_ref = O.mf_apply(om, O.KERNEL_DIFFUSION3D, x.T, np.asfortranarray(y0.T.copy()), alpha=1.25, beta=-0.5,
                       kparams=[0.7, 1.0])
    O.bnd_apply(om, O.KERNEL_ROBIN3D, *whole.boundary_sides(sides), np.asfortranarray(x.T), y_ref, alpha=1.25, kparams=kp)
    row_of = {int(g): i for i, g in enumerate(whole.node_grid_id)}
    for r in range(world):
        y, gid = out[r]
        rows = np.array([row_of[int(g)] for g in gid])
        ref = y_ref.reshape(whole.n_local_nodes, U)[rows]
        assert np.linalg.norm(y.reshape(len(rows), U) - ref) < 1e-11 * np.linalg.norm(ref)


def test_values_at_nodes_vs_oracle(S, ctx):
    """computeValuesAtNodes on the device == the oracle: sums and visit counts of a field-reading kernel with
    derivatives (Diffusion3D error kernel) on sides and on the whole mesh, distorted elements."""
    p, U = 2, 4
    part = S.CubePartition(3, p, perturb=0.15)
    mesh = S.DeviceMesh(ctx, part, U)
    rng = np.random.default_rng(3)
    fields = rng.standard_normal((4, part.n_local_nodes))
    om = helpers.oracle_mesh(part, p + 1, U, [0, 1, 2, 3], fields=fields)
    fe, fs = part.boundary_sides([0, 3, 5])
    for faces in ((fe, fs), (None, None)):
        ws, wc = O.values_at_nodes(om, O.RESIDUAL_DIFFUSION3D_ERROR, [2, 0, 3, 1], faces[0], faces[1], kparams=[0.7, 1.3])
        vals = torch.full((part.n_local_nodes * U,), 7.0, dtype=torch.float64, device="cuda")
        S.values_at_nodes(mesh, S.RESIDUAL_DIFFUSION3D_ERROR, [2, 0, 3, 1], vals, fields=dev(fields), kernel_params=[0.7, 1.3],
                          face_elem=faces[0], face_side=faces[1])
        want = np.where(wc > 0, ws / np.maximum(wc, 1), 7.0)
        assert helpers.rel_err(vals.cpu().numpy(), want) < 1e-12


class ThreadAllReduce:
    """Sum all-reduce between the rank-threads of the single-GPU multi-rank emulation."""

    def __init__(self, world):
        import threading
        self.world, self.slots, self.barrier = world, [None] * world, threading.Barrier(world)

    def bind(self, rank):
        def allreduce(view):
            torch.cuda.synchronize()
            self.slots[rank] = view.clone()
            self.barrier.wait(timeout=120)
            total = sum(self.slots[r] for r in range(self.world))
            self.barrier.wait(timeout=120)
            view.copy_(total)
            torch.cuda.synchronize()
        return allreduce


@pytest.mark.parametrize("ne,parts,adiabatic", [((4, 4, 2), (2, 2, 1), True), ((4, 4, 4), (2, 2, 2), True),
                                                  ((4, 4, 4), (2, 2, 2), False), ((6, 3, 3), (2, 1, 1), False)])
def test_partitioned_solve_end_to_end(S, ctx, ne, parts, adiabatic):
    """The whole partitioned flow on emulated ranks: Dirichlet values from a boundary residual kernel (contributions
    exported to the owners and averaged), diag / rhs with import of the Dirichlet values and export of the ghost rows,
    boundary term, native PCG pieces with all-reduced scalars -- reproduces T = x, q = (1, 0, 0) on a distorted mesh.
    adiabatic = False: T = x prescribed on all six sides instead of the adiabatic boundary term on four of them; without
    boundary terms the element kernels deliver <p, A p> themselves (l3k_mf_energy_*), which every rank must report."""
    import queue
    import threading
    from l3ster_amd import solve
    from l3ster_amd.distributed import DistributedOperator, HaloPlan
    from test_gpu_apply import ThreadTransport
    p, U = 2, 4
    world = int(np.prod(parts))
    boxes = {(a, b): queue.Queue() for a in range(world) for b in range(world)}
    red = ThreadAllReduce(world)
    out, errors = {}, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            part = S.CubePartition(ne, p, parts, rank, perturb=0.1)
            dir_sides = (4, 5) if adiabatic else tuple(range(6))
            mask = part.dirichlet_mask(U, sides=dir_sides)
            c = S.Context(0, torch.cuda.current_stream().cuda_stream)
            mesh = S.DeviceMesh(c, part, U, mask)
            mf = S.MatrixFreeSystem(mesh, S.KERNEL_DIFFUSION3D, [1.0, 0.0])
            if adiabatic:
                mf.attach_boundary(S.BoundaryTerm(mesh, S.KERNEL_ADIABATIC3D, *part.boundary_sides([0, 1, 2, 3])))
            op = DistributedOperator(mf, HaloPlan(part, U, "cuda"), transport=ThreadTransport(rank, boxes))
            n_owned = part.n_owned_nodes * U
            # setDirichletBCValues on a partition: local sums / counts, ghost rows to their owners, average
            import ctypes as C
            from l3ster_amd import capi
            lib = capi.load()
            nl = part.n_local_nodes * U
            sc = torch.zeros((2, nl), dtype=torch.float64, device="cuda")
            fe, fs = part.boundary_sides(list(dir_sides))
            di = (C.c_int * 1)(0)
            capi.check(lib.l3k_values_at_nodes(c._h, mesh._h, S.RESIDUAL_COORDX3D, None, 0, None, 0, 0.0, fe.size,
                                               fe.ctypes.data_as(capi.c_int64_p), fs.ctypes.data_as(capi.c_uint8_p), di,
                                               C.c_void_p(sc[0].data_ptr()), C.c_void_p(sc[1].data_ptr())))
            owned = sc[:, :n_owned].contiguous()
            op.export_add(sc[:, n_owned:].contiguous() if nl > n_owned else torch.zeros((2, 1), dtype=torch.float64, device="cuda"), owned)
            g = torch.zeros(n_owned, dtype=torch.float64, device="cuda")
            capi.check(lib.l3k_average_values(c._h, C.c_void_p(owned[0].data_ptr()), C.c_void_p(owned[1].data_ptr()), n_owned,
                                              C.c_void_p(g.data_ptr())))
            diag, rhs = op.diag_rhs(g[None, :])
            x = torch.zeros(n_owned, dtype=torch.float64, device="cuda")
            res = solve.pcg_distributed(op, c, rhs[0], x, solve.jacobi_inverse_native(c, diag), tol=1e-12,
                                        residual_scaling="rhs", max_iters=5000, allreduce=red.bind(rank))
            torch.cuda.synchronize()
            coords = part.node_coords()[:part.n_owned_nodes]
            sol = x.view(-1, U).cpu().numpy()
            out[rank] = (res.num_iters, np.abs(sol[:, 0] - coords[:, 0]).max(), np.abs(sol[:, 1] - 1.0).max(),
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
        t.join(300)
    assert not errors, errors
    iters = {v[0] for v in out.values()}
    assert len(iters) == 1  # every rank saw the same reduced scalars
    for r in range(world):
        assert out[r][1] < 1e-9 and out[r][2] < 1e-8 and out[r][3] < 1e-8, (r, out[r])
        # <p, A p> from the element kernels unless a boundary term is attached -- or the run is in deterministic mode, which
        # takes the fixed-order dot product instead
        import os
        assert out[r][4] == (not adiabatic and not os.environ.get("L3K_DETERMINISTIC")), (r, out[r])
