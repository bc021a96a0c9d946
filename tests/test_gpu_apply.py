"""GPU parity tests of the matrix-free apply: HIP path (through the C ABI) vs golden fixtures and vs the CPU oracle.
Tolerances: relative L2 <= 1e-12 per element / 1e-11 per mesh (stated fp64 tolerance; the reference's
own cross-path bar is 1e-8 absolute, tests/SumFactorizationTests.cpp:26-27)."""
import numpy as np
import pytest

import oracle_lib as O
from helpers import HEX, SingleElementMesh, oracle_mesh, rel_err
from l3ster_amd import system

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    torch.cuda.set_device(0)
    return system.Context(0, torch.cuda.current_stream().cuda_stream)


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")


ELEMENT_CASES = ["hex_p3_diff", "hex_p3_var", "hex_p4_diff", "hex_p6_diff", "hex_p4_advdiff", "hex_p2_advdiff"]


@pytest.mark.parametrize("name", ELEMENT_CASES)
def test_single_element_vs_golden(ctx, golden, name):
    g = golden(name)
    kid, p, nq, R = int(g["kid"]), int(g["p"]), int(g["nq"]), int(g["R"])
    info = system.kernel_info(kid)
    U = info["n_unknowns"]
    mesh = system.DeviceMesh(ctx, SingleElementMesh(p, g["verts"]), U)
    vo = (nq - 1) // p
    assert system.n_qps1d(p, vo) == nq
    mf = system.MatrixFreeSystem(mesh, kid, g.get("kparams"), asm_opts=(vo, 0, 0), n_rhs=R)
    if info["n_fields"]:
        mf.set_fields(dev(g["node_fields"].T))
    X = dev(g["x"].T)  # (R, Nd)
    Y = torch.full_like(X, 7.0)
    mf.apply(X, Y, 1.0, 0.0)
    torch.cuda.synchronize()
    assert rel_err(Y.cpu().numpy().T, g["y"]) < 1e-12
    # alpha / beta semantics (MatrixFreeSystem.hpp:1038, :511)
    Y0 = dev(np.random.default_rng(0).uniform(-1, 1, g["x"].T.shape))
    Y2 = Y0.clone()
    mf.apply(X, Y2, -0.5, 2.0)
    assert rel_err(Y2.cpu().numpy().T, -0.5 * g["y"] + 2.0 * Y0.cpu().numpy().T) < 1e-12
    # fewer columns than n_rhs (MatrixFreeSystem.hpp:1124-1138): column 0 alone
    if R > 1 and (kid, p, nq, 1) in system.instances():
        Y1 = torch.zeros_like(X[:1])
        mf.apply(X[:1], Y1)
        assert rel_err(Y1.cpu().numpy()[0], g["y"][:, 0]) < 1e-12
    # more columns than n_rhs is an error (MatrixFreeSystem.hpp:1035-1037)
    mf1 = system.MatrixFreeSystem(mesh, kid, g.get("kparams"), asm_opts=(vo, 0, 0), n_rhs=1)
    if info["n_fields"]:
        mf1.set_fields(dev(g["node_fields"].T))
    if R > 1:
        with pytest.raises(system.L3KError, match="columns"):
            mf1.apply(X, Y)


def random_fields(part, F, seed):
    rng = np.random.default_rng(seed)
    return rng.uniform(-1, 1, (F, part.n_local_nodes))


MESH_CASES = [
    # kid, ne, p, value_order, ncols, perturb
    (system.KERNEL_DIFFUSION3D, (3, 2, 2), 1, 1, 1, 0.1),
    (system.KERNEL_DIFFUSION3D, 3, 2, 1, 2, 0.1),
    (system.KERNEL_DIFFUSION3D, 4, 3, 1, 1, 0.0),
    # uniform meshes: every element a parallelepiped -> the affine variant of the one-wave kernel (one Jacobian per element)
    (system.KERNEL_DIFFUSION3D, (3, 2, 2), 6, 1, 1, 0.0),
    (system.KERNEL_DIFFUSION3D, (5, 4, 3), 4, 1, 1, 0.0),
    (system.KERNEL_ADVDIFF3D, 3, 4, 1, 1, 0.0),
    (system.KERNEL_DIFFUSION3D, (5, 4, 3), 4, 1, 1, 0.1),
    (system.KERNEL_DIFFUSION3D, 2, 5, 1, 1, 0.1),
    (system.KERNEL_DIFFUSION3D, 3, 6, 1, 1, 0.1),
    (system.KERNEL_DIFFUSION3D_VAR, 2, 3, 2, 2, 0.1),
    (system.KERNEL_DIFFUSION3D_VAR, 2, 3, 2, 1, 0.1),  # more quadrature points than nodes per direction, one column
    (system.KERNEL_DIFFUSION3D, 3, 3, 2, 1, 0.1),
    (system.KERNEL_DIFFUSION3D_VAR, 3, 4, 1, 1, 0.1),
    (system.KERNEL_ADVDIFF3D, 3, 2, 1, 2, 0.1),
    (system.KERNEL_ADVDIFF3D, 3, 4, 1, 1, 0.1),
    # orders above 6: the one-wave-per-element kernel does not fit (81 pencils > 64 lanes), the generic LDS kernel runs
    (system.KERNEL_DIFFUSION3D, 2, 7, 1, 1, 0.1),
    (system.KERNEL_DIFFUSION3D, 2, 8, 1, 1, 0.1),
    # 3 columns without a 3-column instantiation of this shape: column by column (MatrixFreeSystem.hpp:1124-1138)
    (system.KERNEL_DIFFUSION3D, 3, 4, 1, 3, 0.1),
]


@pytest.mark.parametrize("kid,ne,p,vo,ncols,perturb", MESH_CASES)
def test_mesh_apply_vs_oracle(ctx, kid, ne, p, vo, ncols, perturb):
    """Whole-mesh operator with Dirichlet rows (benchmarks/Diffusion3D.hpp:39-41: unknown 0 on all six sides)."""
    info = system.kernel_info(kid)
    U, F = info["n_unknowns"], info["n_fields"]
    part = system.CubePartition(ne, p, perturb=perturb)
    nq = system.n_qps1d(p, vo)
    mask = part.dirichlet_mask(U)
    mesh = system.DeviceMesh(ctx, part, U, mask)
    kpar = {0: [0.7, 1.0], 4: [0.7, 1.3, 0.5]}.get(kid)
    mf = system.MatrixFreeSystem(mesh, kid, kpar, asm_opts=(vo, 0, 0), n_rhs=ncols)
    fields = random_fields(part, F, 3) if F else None
    if F:
        mf.set_fields(dev(fields))
    x = part.synthetic_vector(U, ncols=ncols)
    y0 = np.random.default_rng(1).uniform(-1, 1, x.shape)
    om = oracle_mesh(part, nq, U, np.arange(U), mask, fields)
    y_ref = O.mf_apply(om, kid, x.T, np.asfortranarray(y0.T.copy()), alpha=1.5, beta=-0.25, kparams=kpar)
    X, Y = dev(x), dev(y0)
    mf.apply(X, Y, 1.5, -0.25)
    torch.cuda.synchronize()
    assert rel_err(Y.cpu().numpy().T, y_ref) < 1e-11


def test_subset_of_node_dofs_and_leading_dimension(ctx):
    """Operator on 4 of 6 per-node dofs through field_inds (detail::getDofs, MatrixFreeSystem.hpp:298-311) with padded
    leading dimensions."""
    p, U, dpn = 2, 4, 6
    part = system.CubePartition(3, p, perturb=0.1)
    fi = [4, 0, 5, 2]
    mask = np.zeros((part.n_local_nodes, dpn), np.uint8)
    mask[part.node_boundary != 0, 4] = 1
    mesh = system.DeviceMesh(ctx, part, dpn, mask.reshape(-1))
    mf = system.MatrixFreeSystem(mesh, system.KERNEL_DIFFUSION3D, field_inds=fi, n_rhs=2)
    assert "strided-dofs" in mf.route(2, 2) and "one per column" in mf.route(2, 2), mf.route(2, 2)  # (the test suite sets generic_below = 0)
    n = part.n_local_nodes * dpn
    pad = 37
    rng = np.random.default_rng(4)
    xb, yb = rng.uniform(-1, 1, (2, n + pad)), rng.uniform(-1, 1, (2, n + pad))
    X, Y = dev(xb), dev(yb)
    mf.apply(X[:, :n], Y[:, :n], 2.0, 1.0)
    om = oracle_mesh(part, p + 1, dpn, fi, mask.reshape(-1))
    y_ref = O.mf_apply(om, 0, xb[:, :n].T, np.asfortranarray(yb[:, :n].T.copy()), alpha=2.0, beta=1.0)
    out = Y.cpu().numpy()
    assert rel_err(out[:, :n].T, y_ref) < 1e-11
    assert np.array_equal(out[:, n:], yb[:, n:])  # padding untouched
    untouched = np.setdiff1d(np.arange(dpn), fi)
    for k in untouched:  # dofs outside field_inds only see beta
        np.testing.assert_array_equal(out[:, k:n:dpn], yb[:, k:n:dpn])


@pytest.mark.parametrize("kid,p,dpn,fi,kpar", [(system.KERNEL_DIFFUSION3D, 2, 6, [4, 0, 5, 2], [0.7, 1.3]), (system.KERNEL_DIFFUSION3D, 4, 5, [1, 2, 3, 4], [1.0, 1.0]),
                                               (system.KERNEL_DIFFUSION3D, 6, 6, [5, 3, 1, 0], [0.7, 1.3]), (system.KERNEL_DIVCURL3D, 4, 5, [3, 1, 4], [0.6]),
                                               (system.KERNEL_ADVECTION3D, 6, 3, [2], [0.05]), (system.KERNEL_ADVDIFF3D, 4, 7, [6, 0, 2, 3], None)])
def test_subset_of_node_dofs_on_the_single_wave_kernel(ctx, kid, p, dpn, fi, kpar):
    """Kernels whose unknowns are a SUBSET of the node's dofs (field_inds: detail::getDofs, MatrixFreeSystem.hpp:298-311 -- several
    kernels sharing one dof map) take the one-wave-per-element kernel too, in its strided-dof variant (8-byte gather / scatter, every
    node through the atomic path): route asserted; y <- alpha A x + beta y against the oracle incl. Dirichlet dofs, dofs outside
    field_inds only scaled by beta; then a rank with ghosts, ghost rows in buffers of their own (the owned-or-ghost select)."""
    info = system.kernel_info(kid)
    U, F = info["n_unknowns"], info["n_fields"]
    part = system.CubePartition(3, p, perturb=0.1)
    mask = np.zeros((part.n_local_nodes, dpn), np.uint8)
    mask[part.node_boundary != 0, fi[0]] = 1
    mask[::7, fi[-1]] = 1
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, dpn, mask.reshape(-1)), kid, kpar, field_inds=fi)
    fields = np.random.default_rng(2).uniform(-1, 1, (F, part.n_local_nodes)) if F else None
    if F:
        mf.set_fields(dev(fields))
    assert mf.route().startswith(f"sumfactFastKernel<p={p},") and "strided-dofs" in mf.route(), mf.route()
    n = part.n_local_nodes * dpn
    rng = np.random.default_rng(4)
    xb, yb = rng.uniform(-1, 1, (1, n)), rng.uniform(-1, 1, (1, n))
    X, Y = dev(xb), dev(yb)
    mf.apply(X, Y, 2.0, 0.5)
    om = oracle_mesh(part, p + 1, dpn, fi, mask.reshape(-1), fields=fields)
    y_ref = O.mf_apply(om, kid, xb.T, np.asfortranarray(yb.T.copy()), alpha=2.0, beta=0.5, kparams=kpar)
    out = Y.cpu().numpy()
    assert rel_err(out.T, y_ref) < 1e-11
    for k in np.setdiff1d(np.arange(dpn), fi):
        np.testing.assert_array_equal(out[:, k:n:dpn], 0.5 * yb[:, k:n:dpn])
    # the same kernel on the generic route agrees to rounding
    with ctx.tuning(generic_below=10 ** 9):
        assert mf.route().startswith("sumfactApplyKernel")
        Yg = dev(yb)
        mf.apply(X, Yg, 2.0, 0.5)
    assert rel_err(Yg.cpu().numpy().T, out.T) < 1e-12
    # a rank with ghosts: separate ghost buffers against ghost rows behind the owned rows
    part2 = system.CubePartition((4, 4, 2), p, parts=(2, 2, 1), rank=3, perturb=0.1)
    mask2 = np.zeros((part2.n_local_nodes, dpn), np.uint8)
    mask2[part2.node_boundary != 0, fi[0]] = 1
    mf2 = system.MatrixFreeSystem(system.DeviceMesh(ctx, part2, dpn, mask2.reshape(-1)), kid, kpar, field_inds=fi)
    if F:
        mf2.set_fields(dev(np.random.default_rng(3).uniform(-1, 1, (F, part2.n_local_nodes))))
    n_owned, n_ghost = part2.n_owned_nodes * dpn, part2.n_ghost_nodes * dpn
    x2 = dev(np.random.default_rng(5).uniform(-1, 1, (1, n_owned + n_ghost)))
    Y2 = torch.zeros((1, n_owned), dtype=torch.float64, device="cuda")
    YG = torch.zeros((1, n_ghost), dtype=torch.float64, device="cuda")
    mf2.apply_elems(2, x2[:, :n_owned].clone(), x2[:, n_owned:].clone(), Y2, YG, 1.5, 0.0)
    yc = torch.zeros_like(x2)
    mf2.apply_elems(2, x2[:, :n_owned], x2[:, n_owned:], yc[:, :n_owned], yc[:, n_owned:], 1.5, 0.0)
    with ctx.tuning(generic_below=10 ** 9):
        yg = torch.zeros_like(x2)
        mf2.apply_elems(2, x2[:, :n_owned], x2[:, n_owned:], yg[:, :n_owned], yg[:, n_owned:], 1.5, 0.0)
    torch.cuda.synchronize()
    scale = float(yg.abs().max())
    assert float((yc[:, :n_owned] - Y2).abs().max()) < 1e-12 * scale and float((yc[:, n_owned:] - YG).abs().max()) < 1e-12 * scale
    assert float((yc - yg).abs().max()) < 1e-12 * scale and float(YG.abs().max()) > 0.0


@pytest.mark.parametrize("ne,p", [(16, 4), (12, 6)])
def test_operator_properties_at_scale(ctx, ne, p):
    """Size-independent properties on a mesh too large for the oracle to be the first resort: symmetry
    <Ax, z> = <x, Az> (the least-squares operator is self-adjoint: apply ignores `mode`, MatrixFreeSystem.hpp:34-41),
    linearity, positive semi-definiteness, and agreement with the oracle on a random sample of rows."""
    U = 4
    part = system.CubePartition(ne, p, perturb=0.1)
    mask = part.dirichlet_mask(U)
    mesh = system.DeviceMesh(ctx, part, U, mask)
    mf = system.MatrixFreeSystem(mesh, system.KERNEL_DIFFUSION3D)
    n = part.n_local_nodes * U
    x = system.synthetic_vector_torch(part.node_grid_id, U, "cuda", seed=1)
    z = system.synthetic_vector_torch(part.node_grid_id, U, "cuda", seed=2)
    Ax, Az, Axz = torch.empty_like(x), torch.empty_like(z), torch.empty_like(x)
    mf.apply(x, Ax)
    mf.apply(z, Az)
    mf.apply(x + 2.0 * z, Axz)
    s1, s2 = torch.dot(Ax[0], z[0]).item(), torch.dot(x[0], Az[0]).item()
    assert abs(s1 - s2) < 1e-11 * max(abs(s1), 1.0)
    assert torch.dot(Ax[0], x[0]).item() > 0
    assert (Axz - (Ax + 2.0 * Az)).norm().item() < 1e-12 * Axz.norm().item()
    # oracle on the elements around a few rows only: compare the full vector but with a threaded oracle
    om = oracle_mesh(part, p + 1, U, np.arange(U), mask)
    y_ref = O.mf_apply(om, 0, x.cpu().numpy().T, nthreads=8)
    assert rel_err(Ax.cpu().numpy().T, y_ref) < 1e-11


DIAG_CASES = [
    # kid, ne, p, value_order, n_rhs
    (system.KERNEL_DIFFUSION3D, 3, 2, 1, 1),
    (system.KERNEL_DIFFUSION3D, 2, 3, 2, 3),
    (system.KERNEL_DIFFUSION3D, 2, 6, 1, 1),
    (system.KERNEL_DIFFUSION3D, 2, 8, 1, 1),
    (system.KERNEL_DIFFUSION3D_VAR, 2, 3, 2, 2),
    (system.KERNEL_ADVDIFF3D, 3, 4, 1, 1),
    (system.KERNEL_ADVDIFF3D, 3, 2, 1, 2),
]


@pytest.mark.parametrize("kid,ne,p,vo,R", DIAG_CASES)
def test_diag_and_lifted_rhs_vs_oracle(ctx, kid, ne, p, vo, R):
    """computeDiagAndRhs (algsys/MatrixFreeSystem.hpp:888-941): diag(A) (needed by the Jacobi preconditioner,
    solve/NativePreconditioners.hpp:36-96) and rhs = sum_e B^T W f - A[:, D] g_D with rhs[D] = g_D, diag[D] = 1."""
    info = system.kernel_info(kid)
    U, F = info["n_unknowns"], info["n_fields"]
    part = system.CubePartition(ne, p, perturb=0.1)
    nq = system.n_qps1d(p, vo)
    mask = part.dirichlet_mask(U)
    mesh = system.DeviceMesh(ctx, part, U, mask)
    kpar = {0: [0.7, 1.3], 4: [0.7, 1.3, 0.5]}.get(kid)
    mf = system.MatrixFreeSystem(mesh, kid, kpar, asm_opts=(vo, 0, 0), n_rhs=R)
    fields = random_fields(part, F, 5) if F else None
    if F:
        mf.set_fields(dev(fields))
    n = part.n_local_nodes * U
    g = np.random.default_rng(6).uniform(-1, 1, (R, n)) * mask[None, :]
    diag, rhs = mf.diag_rhs(dev(g))
    torch.cuda.synchronize()
    om = oracle_mesh(part, nq, U, np.arange(U), mask, fields)
    d_ref, r_ref = O.mf_diag_rhs(om, kid, R, np.asfortranarray(g.T), kparams=kpar)
    assert rel_err(diag.cpu().numpy(), d_ref) < 1e-11
    assert rel_err(rhs.cpu().numpy().T, r_ref) < 1e-11
    assert np.all(diag.cpu().numpy()[mask.astype(bool)] == 1.0)


class ThreadTransport:
    """In-process stand-in for the RCCL neighbour exchange: ranks are threads of this process sharing one GPU, messages
    go through per-(src, dst) queues.  Exercises everything of the multi-rank path except RCCL itself."""

    def __init__(self, rank, boxes):
        self.rank, self.boxes = rank, boxes

    def post(self, sends, recvs):
        torch.cuda.synchronize()  # the payload must be complete before another thread reads it
        for peer, t in sends:
            self.boxes[(self.rank, peer)].put(t.clone())
        return recvs

    def wait(self, recvs):
        for peer, t in recvs:
            t.copy_(self.boxes[(peer, self.rank)].get(timeout=120))
        torch.cuda.synchronize()


@pytest.mark.parametrize("ne,p,parts,ncols", [((4, 2, 2), 2, (2, 1, 1), 1), ((4, 4, 2), 4, (2, 2, 1), 1),
                                              ((2, 4, 4), 2, (1, 2, 2), 2), ((4, 4, 4), 6, (2, 2, 2), 1)])
def test_multi_rank_schedule_on_one_gpu(ctx, ne, p, parts, ncols):
    """DistributedOperator (pack -> import || interior -> border -> export -> unpack-add -> Dirichlet rows) with the HIP
    split-phase kernels, ranks emulated by threads: the assembled result equals the oracle on the whole mesh."""
    import queue
    import threading
    from l3ster_amd.distributed import DistributedOperator, HaloPlan
    world = int(np.prod(parts))
    U, kid = 4, system.KERNEL_DIFFUSION3D
    boxes = {(a, b): queue.Queue() for a in range(world) for b in range(world)}
    out, errors = {}, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            part = system.CubePartition(ne, p, parts, rank, perturb=0.1)
            mask = part.dirichlet_mask(U)
            c = system.Context(0, torch.cuda.current_stream().cuda_stream)
            mesh = system.DeviceMesh(c, part, U, mask)
            mf = system.MatrixFreeSystem(mesh, kid, [0.7, 1.0], n_rhs=ncols)
            n_owned = part.n_owned_nodes * U
            X = dev(part.synthetic_vector(U, ncols=ncols)[:, :n_owned])
            Y = dev(part.synthetic_vector(U, seed=7, ncols=ncols)[:, :n_owned])
            op = DistributedOperator(mf, HaloPlan(part, U, "cuda"), transport=ThreadTransport(rank, boxes))
            for _ in range(2):
                Yc = Y.clone()
                op.apply(X, Yc, 1.25, -0.5)
            torch.cuda.synchronize()
            out[rank] = (Yc.cpu().numpy(), part.node_grid_id[:part.n_owned_nodes].copy())
        except Exception as exc:  # pragma: no cover
            errors.append((rank, exc))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    whole = system.CubePartition(ne, p, perturb=0.1)
    mask = whole.dirichlet_mask(U)
    x, y0 = whole.synthetic_vector(U, ncols=ncols), whole.synthetic_vector(U, seed=7, ncols=ncols)
    y_ref = O.mf_apply(oracle_mesh(whole, p + 1, U, np.arange(U), mask), kid, x.T, np.asfortranarray(y0.T.copy()),
                       alpha=1.25, beta=-0.5, kparams=[0.7, 1.0], nthreads=4)
    row_of = {int(g): i for i, g in enumerate(whole.node_grid_id)}
    for r in range(world):
        y, gid = out[r]
        rows = np.array([row_of[int(g)] for g in gid])
        ref = y_ref.reshape(whole.n_local_nodes, U, ncols)[rows]
        got = y.reshape(ncols, len(rows), U).transpose(1, 2, 0)
        assert np.linalg.norm(got - ref) < 1e-11 * np.linalg.norm(ref)


@pytest.mark.parametrize("ne,p", [(64, 6), (64, 4)])
def test_operator_properties_at_full_size(ctx, ne, p):
    """BASELINE.json's single-GPU sizes (64^3 elements, orders 6 and 4: 228 M / 68 M dofs), through properties that do not
    need the oracle on the whole mesh: self-adjointness, linearity, positive semi-definiteness; the exact linear
    field T = x, q = (1, 0, 0) (in the discrete space whatever the distortion) is annihilated by the interior rows of the
    source-free operator; and the whole output vector agrees with the oracle on the whole mesh (rel L2 < 1e-12)."""
    U = 4
    part = system.CubePartition(ne, p, perturb=0.1)
    mask = part.dirichlet_mask(U)
    mesh = system.DeviceMesh(ctx, part, U, mask)
    mf = system.MatrixFreeSystem(mesh, system.KERNEL_DIFFUSION3D, [1.0, 0.0])
    x = system.synthetic_vector_torch(part.node_grid_id, U, "cuda", seed=1)
    z = system.synthetic_vector_torch(part.node_grid_id, U, "cuda", seed=2)
    Ax, Az, Axz = torch.empty_like(x), torch.empty_like(z), torch.empty_like(x)
    mf.apply(x, Ax)
    mf.apply(z, Az)
    s1, s2 = torch.dot(Ax[0], z[0]).item(), torch.dot(x[0], Az[0]).item()
    assert abs(s1 - s2) < 1e-11 * max(abs(s1), 1.0)
    assert torch.dot(Ax[0], x[0]).item() > 0
    z.mul_(2.0).add_(x)
    mf.apply(z, Axz)
    Az.mul_(2.0).add_(Ax)
    assert (Axz - Az).norm().item() < 1e-12 * Axz.norm().item()
    del Az, Axz, z
    # exact linear solution: B u = 0 at every quadrature point, so A u vanishes on all rows that are not Dirichlet rows
    # (the gather reads Dirichlet dofs as 0, so rows of elements touching the boundary are excluded)
    coords = torch.as_tensor(part.node_coords()[:, 0], device="cuda")
    u = torch.zeros((part.n_local_nodes, U), dtype=torch.float64, device="cuda")
    u[:, 0], u[:, 1] = coords, 1.0
    Au = torch.empty_like(x)
    mf.apply(u.view(1, -1), Au)
    touched = np.zeros(part.n_local_nodes, bool)
    touched[part.elem_nodes[part.elem_boundary != 0].reshape(-1)] = True
    inner = torch.as_tensor(~touched, device="cuda")
    assert Au.view(-1, U)[inner].abs().max().item() < 1e-9 * Ax.abs().max().item()
    # the WHOLE output vector of the production launch (dynamic XCD-chunked batch distribution with cross-XCD
    # continuation, which only carries real work at this size) against the oracle on the same mesh and x: every XCD chunk,
    # every Dirichlet-face brick and the last tickets are covered.  The oracle runs on all host cores (a few seconds).
    import os
    om = O.MeshView(3, p, p + 1, part.elem_nodes, part.elem_verts, part.n_local_nodes, U, np.arange(U), mask)
    ys = O.mf_apply(om, 0, x.cpu().numpy().T, kparams=[1.0, 0.0], nthreads=len(os.sched_getaffinity(0)))
    assert rel_err(Ax.cpu().numpy()[0], ys[:, 0]) < 1e-12
    # ... and against the static deal of the same kernel (another route through the element list)
    with ctx.tuning(static_deal=1):
        assert "static batches" in mf.route()
        Ay = torch.empty_like(x)
        mf.apply(x, Ay)
        torch.cuda.synchronize()
    assert "dynamic batches" in mf.route() and f"sumfactFastKernel<p={p},nq={p + 1},U=4,F=0>" in mf.route()
    assert (Ay - Ax).norm().item() < 1e-13 * Ax.norm().item()


def _free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


@pytest.mark.gpu
def test_rccl_transport_on_one_gpu():
    """The RCCL point-to-point transport of the partitioned apply, as far as one GPU allows (see the script's docstring)."""
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_self_exchange.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "RCCL self exchange ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_partitioned_apply_through_rccl_on_one_gpu():
    """8 logical ranks' partitioned apply with every message carried by RCCL (self send / receive on one GPU): see the
    script's docstring."""
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_self_partitioned_apply.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "RCCL partitioned apply ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("order", [6, 2])
def test_native_rccl_halo_self_exchange(order):
    """l3k_halo_* / l3k_mf_apply_dist (the RCCL neighbour exchange inside the library, behind the C ABI): a one-rank cube
    made periodic in x through the ghost machinery, import and export as ncclSend / ncclRecv of the rank to itself; against
    the oracle on the periodic mesh and against the Python-side schedule (see the script's docstring)."""
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_native_halo_periodic.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, script, "--order", str(order)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "native halo ok" in r.stdout, r.stdout + r.stderr
