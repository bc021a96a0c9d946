"""GPU parity of the kernel-definition API where the kernel READS ITS INPUT POINT: domain kernels whose operators and rhs depend on
in.point.space.{x,y,z} and in.point.time (the reference's examples do: examples/03-advection-2D/source.cpp:52-66 takes the velocity
from point.space.y(), examples/04-periodic-bc/source.cpp:60-95 reads point.time), on both apply routes (one wave per element /
generic LDS kernel), in diag / rhs and in LocalAssembly, at two times (l3k_mf_set_time), and in the two point modes:

  * default: the true point everywhere -- apply == the reference's LOCAL-ELEMENT path (algsys/AssembleLocalSystem.hpp:229-230);
  * l3k_ctx_set_reference_z0: the apply hands the kernel Point{x, y, 0.} as the reference's hex SUM-FACTORISATION path does
    (algsys/SumFactorization.hpp:732); diag / rhs / LocalAssembly keep the true point, as in the reference.

Also here: the kernels with an odd number of unknowns (scalar advection U = 1, div-curl U = 3).  HIP path through the C ABI vs the
golden fixtures (numpy restatement) and vs the CPU oracle; tolerances 1e-12 per element, 1e-11 per mesh (relative L2)."""
import numpy as np
import pytest

import oracle_lib as O
from helpers import HEX, SingleElementMesh, oracle_mesh, rel_err
from l3ster_amd import system

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

KPAR = {system.KERNEL_DIFFUSION3D_POINT: [0.8, 1.2], system.KERNEL_ADVECTION3D: [0.05], system.KERNEL_DIVCURL3D: [0.6]}


@pytest.fixture(scope="module")
def ctx():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    torch.cuda.set_device(0)
    return system.Context(0, torch.cuda.current_stream().cuda_stream)


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")


ROUTES = {"fast": 0, "generic": 10 ** 9}  # l3k_tuning::generic_below


@pytest.mark.parametrize("route", list(ROUTES))
@pytest.mark.parametrize("name", ["hex_p2_point", "hex_p4_point", "hex_p4_advection", "hex_p2_divcurl"])
def test_single_element_vs_golden(ctx, golden, name, route):
    """apply, diag, lifted rhs, K_e and F_e of one distorted element at t != 0 against the numpy restatement's fixture."""
    g = golden(name)
    kid, p, nq, R, t = int(g["kid"]), int(g["p"]), int(g["nq"]), int(g["R"]), float(g["time"])
    info = system.kernel_info(kid)
    U, F = info["n_unknowns"], info["n_fields"]
    vo = (nq - 1) // p
    assert system.n_qps1d(p, vo) == nq
    Nd = (p + 1) ** 3 * U
    mask = np.zeros(Nd, np.uint8)
    mask[g["dir_inds"]] = 1
    for dirichlet in (None, mask):
        mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, SingleElementMesh(p, g["verts"]), U, dirichlet), kid, g.get("kparams"),
                                     asm_opts=(vo, 0, 0), n_rhs=R)
        if F:
            mf.set_fields(dev(g["node_fields"].T))
        mf.set_time(t)
        with ctx.tuning(generic_below=ROUTES[route]):
            if dirichlet is None:
                X, Y = dev(g["x"].T), torch.full((R, Nd), 7.0, dtype=torch.float64, device="cuda")
                mf.apply(X, Y, 1.0, 0.0)
                assert rel_err(Y.cpu().numpy().T, g["y"]) < 1e-12
                K, Fe, _ = mf.local_assemble()
                assert np.abs(Fe.cpu().numpy()[0].T - g["F"]).max() < 1e-12 * max(1.0, np.abs(g["F"]).max())
                if "K" in g:
                    assert np.abs(K.cpu().numpy()[0] - g["K"]).max() < 1e-12 * np.abs(g["K"]).max()
            else:
                gd = np.zeros((R, Nd))
                gd[:, g["dir_inds"]] = g["dir_vals"].T
                diag, rhs = mf.diag_rhs(dev(gd), finalize=False)
                free = mask == 0
                np.testing.assert_allclose(diag.cpu().numpy(), g["diag"], rtol=1e-12, atol=1e-13)
                assert rel_err(rhs.cpu().numpy().T[free], g["rhs_lifted"][free]) < 1e-12


def fields_for(part, F, seed=3):
    return np.random.default_rng(seed).uniform(-1, 1, (F, part.n_local_nodes)) if F else None


MESH_CASES = [
    # kid, ne, p, value_order
    (system.KERNEL_DIFFUSION3D_POINT, 3, 2, 1),
    (system.KERNEL_DIFFUSION3D_POINT, 2, 2, 2),
    (system.KERNEL_DIFFUSION3D_POINT, (3, 2, 2), 4, 1),
    (system.KERNEL_DIFFUSION3D_POINT, 2, 6, 1),
    (system.KERNEL_ADVECTION3D, 3, 2, 1),
    (system.KERNEL_ADVECTION3D, (3, 2, 2), 4, 1),
    (system.KERNEL_ADVECTION3D, 2, 6, 1),
    (system.KERNEL_DIVCURL3D, 3, 2, 1),
    (system.KERNEL_DIVCURL3D, (3, 2, 2), 4, 1),
    (system.KERNEL_DIVCURL3D, 2, 6, 1),
]


@pytest.mark.parametrize("route", list(ROUTES))
@pytest.mark.parametrize("kid,ne,p,vo", MESH_CASES)
def test_mesh_apply_diag_rhs_at_two_times(ctx, kid, ne, p, vo, route):
    """Whole-mesh operator on a perturbed mesh, Dirichlet on unknown 0 of all sides, at two times through l3k_mf_set_time: apply
    (alpha, beta), diag and lifted rhs against the oracle's mesh-level functions (true point)."""
    info = system.kernel_info(kid)
    U, F = info["n_unknowns"], info["n_fields"]
    part = system.CubePartition(ne, p, perturb=0.15)
    nq = system.n_qps1d(p, vo)
    mask = part.dirichlet_mask(U)
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, U, mask), kid, KPAR[kid], asm_opts=(vo, 0, 0))
    fields = fields_for(part, F)
    if F:
        mf.set_fields(dev(fields))
    om = oracle_mesh(part, nq, U, np.arange(U), mask, fields)
    x = part.synthetic_vector(U)
    y0 = np.random.default_rng(1).uniform(-1, 1, x.shape)
    gd = np.random.default_rng(6).uniform(-1, 1, (1, part.n_local_nodes * U)) * mask[None, :]
    results = []
    with ctx.tuning(generic_below=ROUTES[route]):
        # the route under test -- kernels with an odd number of unknowns (U = 1, 3) take the one-wave-per-element kernel too
        assert mf.route().startswith(f"sumfactFastKernel<p={p},nq={nq},U={U},F={F}>" if route == "fast" else "sumfactApplyKernel"), mf.route()
        for t in (0.0, 0.85):
            mf.set_time(t)
            X, Y = dev(x), dev(y0)
            mf.apply(X, Y, 1.5, -0.25)
            y_ref = O.mf_apply(om, kid, x.T, np.asfortranarray(y0.T.copy()), alpha=1.5, beta=-0.25, kparams=KPAR[kid], time=t)
            assert rel_err(Y.cpu().numpy().T, y_ref) < 1e-11, t
            diag, rhs = mf.diag_rhs(dev(gd))
            d_ref, r_ref = O.mf_diag_rhs(om, kid, 1, np.asfortranarray(gd.T), kparams=KPAR[kid], time=t)
            assert rel_err(diag.cpu().numpy(), d_ref) < 1e-11 and rel_err(rhs.cpu().numpy().T, r_ref) < 1e-11, t
            results.append(Y.cpu().numpy())
    if kid == system.KERNEL_DIFFUSION3D_POINT:  # the time really enters the operator
        assert rel_err(results[0], results[1]) > 1e-3


@pytest.mark.parametrize("kid,p,vo", [(system.KERNEL_DIFFUSION3D_POINT, 2, 2), (system.KERNEL_DIFFUSION3D_POINT, 4, 1),
                                      (system.KERNEL_DIFFUSION3D_POINT, 6, 1), (system.KERNEL_ADVECTION3D, 4, 1),
                                      (system.KERNEL_DIVCURL3D, 4, 1), (system.KERNEL_DIVCURL3D, 6, 1)])
def test_local_assembly_vs_oracle_entrywise(ctx, kid, p, vo):
    """l3k_local_assemble of point-reading kernels (and of odd numbers of unknowns) on elements of a perturbed mesh at t = 0.6:
    K_e entry by entry and F_e against the oracle's assembleLocalSystem (true point, AssembleLocalSystem.hpp:229-230); the
    streaming checksum equals the stored matrix's."""
    info = system.kernel_info(kid)
    U, F = info["n_unknowns"], info["n_fields"]
    part = system.CubePartition((2, 2, 1), p, perturb=0.2)
    nq = system.n_qps1d(p, vo)
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, U), kid, KPAR[kid], asm_opts=(vo, 0, 0))
    fields = fields_for(part, F, 2)
    if F:
        mf.set_fields(dev(fields))
    mf.set_time(0.6)
    K, Fe, cs = mf.local_assemble(want_checksum=True)
    _, _, cs_stream = mf.local_assemble(want_K=False, want_F=False, want_checksum=True)
    torch.cuda.synchronize()
    K, Fe = K.cpu().numpy(), Fe.cpu().numpy()
    for e in (0, 3):
        nf = fields[:, part.elem_nodes[e]].T if F else None
        K_ref, F_ref = O.assemble_local(kid, p, nq, 1, part.elem_verts[e], nf, KPAR[kid], time=0.6)
        assert np.abs(K[e] - K_ref).max() < 1e-12 * np.abs(K_ref).max(), e
        assert np.abs(Fe[e].T - F_ref).max() < 1e-12 * max(1.0, np.abs(F_ref).max()), e
        assert np.array_equal(K[e], K[e].T)
    np.testing.assert_allclose(cs_stream.cpu().numpy(), cs.cpu().numpy(), rtol=1e-11)


@pytest.mark.parametrize("route", list(ROUTES))
@pytest.mark.parametrize("p", [2, 4, 6])
def test_reference_z0_mode(ctx, p, route):
    """The reference's two paths disagree about the point (D8).  Default: apply == the local-element operator (true z).  With
    l3k_ctx_set_reference_z0(1): apply == the reference's sum-factorisation path (z = 0, the oracle with the same switch), another
    operator for a kernel that reads z; diag / rhs and K_e stay on the true point as in the reference; switching back restores the
    default.  The route line names the mode."""
    kid, U = system.KERNEL_DIFFUSION3D_POINT, 4
    part = system.CubePartition(2 if p == 6 else 3, p, perturb=0.15)
    mask = part.dirichlet_mask(U)
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, U, mask), kid, KPAR[kid])
    mf.set_time(0.3)
    om = oracle_mesh(part, p + 1, U, np.arange(U), mask)
    x = part.synthetic_vector(U)
    X = dev(x)
    y_true = O.mf_apply(om, kid, x.T, kparams=KPAR[kid], time=0.3)
    try:
        O.set_reference_z0(True)
        y_z0 = O.mf_apply(om, kid, x.T, kparams=KPAR[kid], time=0.3)
    finally:
        O.set_reference_z0(False)
    assert rel_err(y_z0, y_true) > 1e-3  # not a rounding effect
    d_ref, r_ref = O.mf_diag_rhs(om, kid, 1, kparams=KPAR[kid], time=0.3)
    with ctx.tuning(generic_below=ROUTES[route]):
        Y = torch.zeros_like(X)
        mf.apply(X, Y)
        assert rel_err(Y.cpu().numpy().T, y_true) < 1e-11 and "reference z=0" not in mf.route()
        ctx.set_reference_z0(True)
        try:
            assert "reference z=0" in mf.route()
            Yz = torch.zeros_like(X)
            mf.apply(X, Yz)
            assert rel_err(Yz.cpu().numpy().T, y_z0) < 1e-11
            diag, rhs = mf.diag_rhs(None)  # the local-element path of the reference: true point in both modes
            assert rel_err(diag.cpu().numpy(), d_ref) < 1e-11 and rel_err(rhs.cpu().numpy().T, r_ref) < 1e-11
        finally:
            ctx.set_reference_z0(False)
        mf.apply(X, Y)
        assert rel_err(Y.cpu().numpy().T, y_true) < 1e-11


def test_z0_mode_leaves_kernels_that_ignore_z_alone(ctx):
    """Diffusion3D does not read the point: both modes give bitwise the same launch results on one element stream order
    (deterministic mode, so that the atomics do not blur the comparison)."""
    ctx2 = system.Context(0, torch.cuda.current_stream().cuda_stream)
    ctx2.set_deterministic(True)
    ctx2.set_tuning(generic_below=0)
    U, p = 4, 4
    part = system.CubePartition(3, p, perturb=0.1)
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx2, part, U, part.dirichlet_mask(U)), system.KERNEL_DIFFUSION3D, [0.7, 1.0])
    X = dev(part.synthetic_vector(U))
    Y0, Y1 = torch.zeros_like(X), torch.zeros_like(X)
    mf.apply(X, Y0)
    ctx2.set_reference_z0(True)
    mf.apply(X, Y1)
    torch.cuda.synchronize()
    assert torch.equal(Y0, Y1)


def test_update_solution_vs_oracle(ctx):
    """l3k_update_solution = MatrixFreeSystem::updateSolution (algsys/MatrixFreeSystem.hpp:1231-1273): dofs of the solution (two
    columns, a subset of the per-node dofs, owned rows + imported ghost rows) into SolutionManager-style SoA fields; the reference's
    index asserts come back as errors."""
    p, dpn = 3, 5
    part = system.CubePartition((4, 2, 2), p, parts=(2, 1, 1), rank=1, perturb=0.1)  # a rank with ghost nodes
    assert part.n_ghost_nodes > 0
    mesh = system.DeviceMesh(ctx, part, dpn)
    rng = np.random.default_rng(8)
    n_owned, n_ghost = part.n_owned_nodes * dpn, part.n_ghost_nodes * dpn
    x = rng.uniform(-1, 1, (2, n_owned + n_ghost))
    sol_inds, dest = [3, 0, 4], [5, 1, 0, 6, 2, 7]  # (index-major: (3, col 0) -> field 5, (3, col 1) -> field 1, (0, col 0) -> 0, ...)
    f0 = rng.uniform(-1, 1, (9, part.n_local_nodes))
    want = O.update_solution(oracle_mesh(part, p + 1, dpn, np.arange(dpn)), x.T, sol_inds, f0.copy(), dest)
    got = system.update_solution(mesh, dev(x[:, :n_owned]), sol_inds, dev(f0), dest, XG=dev(x[:, n_owned:]))
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(want[[3, 4, 8]], f0[[3, 4, 8]])  # fields that are no destination keep their values
    with pytest.raises(system.L3KError, match="Source index out of bounds"):
        system.update_solution(mesh, dev(x[:, :n_owned]), [5], dev(f0), [0, 1], XG=dev(x[:, n_owned:]))
    with pytest.raises(system.L3KError, match="Destination index out of bounds"):
        system.update_solution(mesh, dev(x[:, :n_owned]), [1], dev(f0), [0, 9], XG=dev(x[:, n_owned:]))
    with pytest.raises(system.L3KError, match="lengths must match"):
        system.update_solution(mesh, dev(x[:, :n_owned]), [1, 2], dev(f0), [0, 1, 2], XG=dev(x[:, n_owned:]))
    with pytest.raises(system.L3KError, match="ghost"):
        system.update_solution(mesh, dev(x[:, :n_owned]), [1], dev(f0), [0, 1])


def test_bdf3_advection_time_stepping_end_to_end(ctx):
    """The loop of examples/04-periodic-bc/source.cpp:97-140 in 3-D with the scalar advection kernel: per time step set the time,
    compute diag + rhs with the three previous solutions as the kernel's fields (BDF3), solve with Jacobi-PCG, updateSolution rotates
    the new solution into the field storage.  Device (fast scalar kernel, l3k_pcg_solve, l3k_update_solution) against the same loop
    on the CPU with the oracle's operator, diag / rhs and updateSolution driving the torch-op CG: solutions agree after 3 steps."""
    from l3ster_amd import solve
    kid, p, U, dt = system.KERNEL_ADVECTION3D, 2, 1, 0.05
    part = system.CubePartition((4, 3, 3), p, perturb=0.1)
    mask = part.dirichlet_mask(U, sides=[4])  # inflow: the x- side
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, U, mask), kid, [dt])
    xyz = part.node_coords()
    u0 = np.exp(-10.0 * ((xyz[:, 0] - 0.4) ** 2 + (xyz[:, 1] - 0.5) ** 2 + (xyz[:, 2] - 0.5) ** 2))
    fields_d = dev(np.stack([u0, u0, u0]))  # history: u^n, u^{n-1}, u^{n-2}
    fields_h = np.stack([u0, u0, u0])
    mf.set_fields(fields_d)
    n = part.n_local_nodes
    on_inflow = mask.astype(bool)
    for step in range(3):
        t = (step + 1) * dt
        g = np.zeros((1, n))  # homogeneous inflow value
        # ---- device
        mf.set_time(t)
        diag, rhs = mf.diag_rhs(dev(g))
        minv = solve.jacobi_inverse_native(ctx, diag)
        x_d = fields_d[0].clone()
        x_d[torch.as_tensor(on_inflow, device="cuda")] = 0.0
        r_d = solve.pcg(mf, rhs[0], x_d, minv, tol=1e-11, residual_scaling="rhs")
        # rotate the history: u^{n-1} -> u^{n-2}, u^n -> u^{n-1}, then the new solution into slot 0 through updateSolution
        fields_d[2].copy_(fields_d[1])
        fields_d[1].copy_(fields_d[0])
        system.update_solution(mf.mesh, x_d[None, :], [0], fields_d, [0])
        # ---- the same step on the CPU: oracle operator, diag / rhs, updateSolution; torch-op CG
        om = oracle_mesh(part, p + 1, U, np.arange(U), mask, fields_h)
        d_h, r_h = O.mf_diag_rhs(om, kid, 1, np.asfortranarray(g.T), kparams=[dt], time=t)
        x_h = torch.as_tensor(fields_h[0].copy())
        x_h[torch.as_tensor(on_inflow)] = 0.0
        apply_h = lambda v, out: out.copy_(torch.as_tensor(O.mf_apply(om, kid, v.numpy().reshape(-1, 1), kparams=[dt], time=t)[:, 0]))
        r_c = solve.cg(apply_h, torch.as_tensor(r_h[:, 0].copy()), x_h, torch.as_tensor(1.0 / d_h), tol=1e-11, residual_scaling="rhs")
        assert abs(r_c.num_iters - r_d.num_iters) <= 3, (step, r_c.num_iters, r_d.num_iters)
        fields_h[2], fields_h[1] = fields_h[1].copy(), fields_h[0].copy()
        O.update_solution(oracle_mesh(part, p + 1, U, np.arange(U)), x_h.numpy(), [0], fields_h, [0])
        err = np.abs(fields_d.cpu().numpy() - fields_h).max()
        assert err < 1e-8, (step, err)
    # the bump has moved: the solution changed, and stayed bounded
    assert 0.01 < np.abs(fields_h[0] - u0).max() < 1.0
