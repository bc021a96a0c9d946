"""Empty and degenerate inputs through the C ABI on the GPU (the reference's tests exercise empty domains / boundary
views and zero-sized ranges implicitly through its views; here they are explicit)."""
import numpy as np
import pytest

import helpers

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


class EmptyMesh:
    """A partition that owns nodes but no elements (possible for a rank of a badly balanced partition)."""
    dim, order = 3, 2
    n_elems = n_interior_elems = 0
    n_owned_nodes, n_ghost_nodes = 5, 0
    elem_nodes = np.zeros((0, 27), np.uint32)
    elem_verts = np.zeros((0, 8, 3))
    n_local_nodes = 5


def test_mesh_without_elements(S, ctx):
    U = 4
    mask = np.zeros(5 * U, np.uint8)
    mask[[0, 7]] = 1
    mesh = S.DeviceMesh(ctx, EmptyMesh(), U, mask)
    mf = S.MatrixFreeSystem(mesh, S.KERNEL_DIFFUSION3D, [1.0, 1.0])
    x = torch.arange(1.0, 21.0, dtype=torch.float64, device="cuda")[None, :]
    y = torch.full_like(x, 3.0)
    mf.apply(x, y, 2.0, 0.5)  # y <- 0.5*y + 2*x on Dirichlet rows (MatrixFreeSystem.hpp:1038,1087-1098)
    want = np.full(20, 1.5)
    want[[0, 7]] += 2.0 * np.array([1.0, 8.0])
    assert np.array_equal(y.cpu().numpy()[0], want)
    diag, rhs = mf.diag_rhs(None)
    assert diag.cpu().numpy()[[0, 7]].tolist() == [1.0, 1.0] and diag.sum().item() == 2.0 and rhs.abs().sum().item() == 0.0
    K, F, cs = mf.local_assemble(0, 0, want_checksum=True)
    assert K.shape[0] == 0 and cs.numel() == 0
    assert S.integrate(mesh, S.RESIDUAL_UNIT3D).tolist() == [0.0]


def test_empty_side_lists(S, ctx):
    part = S.CubePartition(2, 2)
    U = 4
    mesh = S.DeviceMesh(ctx, part, U)
    none = (np.zeros(0, np.int64), np.zeros(0, np.uint8))
    term = S.BoundaryTerm(mesh, S.KERNEL_ROBIN3D, *none, kernel_params=[1.0, 1.0])
    mf = S.MatrixFreeSystem(mesh, S.KERNEL_DIFFUSION3D, [1.0, 1.0])
    x = torch.as_tensor(part.synthetic_vector(U), device="cuda")
    y_plain = torch.zeros_like(x)
    mf.apply(x, y_plain)
    mf.attach_boundary(term)
    y_bnd = torch.zeros_like(x)
    mf.apply(x, y_bnd)
    assert helpers.rel_err(y_bnd.cpu().numpy(), y_plain.cpu().numpy()) < 1e-13  # an empty boundary adds nothing
    assert S.integrate(mesh, S.RESIDUAL_UNIT3D, asm_opts=(1, 0, 0), face_elem=none[0], face_side=none[1]).tolist() == [0.0]
    vals = torch.full((part.n_local_nodes * U,), 2.5, dtype=torch.float64, device="cuda")
    S.values_at_nodes(mesh, S.RESIDUAL_COORDX3D, [0], vals, face_elem=none[0], face_side=none[1])
    assert torch.all(vals == 2.5)


def test_argument_errors_are_reported(S, ctx):
    part = S.CubePartition(2, 2)
    mesh = S.DeviceMesh(ctx, part, 4)
    with pytest.raises(S.L3KError, match="outside the mesh"):
        S.BoundaryTerm(mesh, S.KERNEL_ROBIN3D, [99], [0], kernel_params=[1.0, 1.0])
    with pytest.raises(S.L3KError, match="outside the mesh"):
        S.integrate(mesh, S.RESIDUAL_UNIT3D, face_elem=[0], face_side=[6])
    with pytest.raises(S.L3KError, match="boundary equation kernel"):
        S.MatrixFreeSystem(mesh, S.KERNEL_ROBIN3D, [1.0, 1.0])
    with pytest.raises(S.L3KError, match="not a boundary"):
        S.BoundaryTerm(mesh, S.KERNEL_DIFFUSION3D, [0], [0], kernel_params=[1.0, 1.0])
    with pytest.raises(S.L3KError, match="no device instantiation"):
        S.integrate(mesh, S.RESIDUAL_DIFFUSION3D_ERROR, torch.zeros((4, part.n_local_nodes), dtype=torch.float64, device="cuda"),
                    asm_opts=(3, 0, 0))
    with pytest.raises(S.L3KError, match="16-byte|parameter block"):
        S.MatrixFreeSystem(mesh, S.KERNEL_DIFFUSION3D, [1.0])


def _single_column_shapes(S):
    """(order, nq, (value_order, derivative_order)) of every compiled single-column Diffusion3D instantiation"""
    out = []
    for kid, p, nq, nc in S.instances():
        if kid != S.KERNEL_DIFFUSION3D or nc != 1:
            continue
        opts = [(vo, do) for vo in range(1, 4) for do in range(0, 3) if S.n_qps1d(p, vo, do) == nq]
        if opts:
            out.append((p, nq, opts[0]))
    return sorted(set(out))


@pytest.mark.parametrize("seed", range(16))
def test_randomised_apply_and_diag_rhs_parity(S, ctx, seed):
    """Seeded random problems over every compiled single-column shape (orders 1-8, quadrature sizes equal to and larger
    than the node count): element counts, distortion, Dirichlet dofs anywhere (any unknown, interior nodes included),
    alpha / beta -- apply, diagonal and lifted rhs against the oracle."""
    import oracle_lib as O
    rng = np.random.default_rng(1000 + seed)
    shapes = _single_column_shapes(S)
    p, nq, (vo, do) = shapes[seed % len(shapes)]
    ne = tuple(int(v) for v in rng.integers(1, 3 if p > 5 else (4 if p > 3 else 5), size=3))
    U = 4
    part = S.CubePartition(ne, p, perturb=float(rng.uniform(0, 0.15)))
    mask = (rng.uniform(size=part.n_local_nodes * U) < rng.choice([0.0, 0.02, 0.2])).astype(np.uint8)
    use_mask = bool(mask.any() or seed % 2)
    mesh = S.DeviceMesh(ctx, part, U, mask if use_mask else None)
    kp = [float(rng.uniform(0.5, 2)), float(rng.uniform(-1, 1))]
    mf = S.MatrixFreeSystem(mesh, S.KERNEL_DIFFUSION3D, kp, asm_opts=(vo, do, 0))
    om = helpers.oracle_mesh(part, nq, U, np.arange(U), mask if use_mask else None)
    alpha, beta = float(rng.uniform(-2, 2)), float(rng.choice([0.0, 1.0, rng.uniform(-1, 1)]))
    x = rng.standard_normal((1, part.n_local_nodes * U))
    y0 = rng.standard_normal(x.shape)
    Y = torch.as_tensor(y0, device="cuda").clone()
    mf.apply(torch.as_tensor(x, device="cuda"), Y, alpha, beta)
    want = O.mf_apply(om, O.KERNEL_DIFFUSION3D, x.T, np.asfortranarray(y0.T.copy()), alpha=alpha, beta=beta, kparams=kp)
    assert helpers.rel_err(Y.cpu().numpy().T, want) < 1e-11, (p, nq, ne)
    g = np.where(mask != 0, rng.standard_normal(mask.size), 0.0)[None, :]
    diag, rhs = mf.diag_rhs(torch.as_tensor(g, device="cuda"))
    wd, wr = O.mf_diag_rhs(om, O.KERNEL_DIFFUSION3D, dirichlet_vals=np.asfortranarray(g.T), kparams=kp)
    assert helpers.rel_err(diag.cpu().numpy(), wd) < 1e-11 and helpers.rel_err(rhs.cpu().numpy().T, wr) < 1e-10, (p, nq, ne)


def test_small_launch_routing(S, ctx):
    """Default routing: a mesh below the small-launch threshold goes through the generic LDS kernel, above it through the
    one-wave-per-element kernel; both agree with the oracle and with each other."""
    import oracle_lib as O
    p, U = 6, 4
    part = S.CubePartition(3, p, perturb=0.1)  # 27 elements
    mask = part.dirichlet_mask(U)
    mesh = S.DeviceMesh(ctx, part, U, mask)
    mf = S.MatrixFreeSystem(mesh, S.KERNEL_DIFFUSION3D, [1.0, 1.0])
    x = part.synthetic_vector(U)
    want = O.mf_apply(helpers.oracle_mesh(part, p + 1, U, np.arange(U), mask), O.KERNEL_DIFFUSION3D, x.T, kparams=[1.0, 1.0])
    out = {}
    for name, value in (("generic", 1500), ("fast", 0)):  # (1500 = the product default, l3k_tuning::generic_below)
        with ctx.tuning(generic_below=value):
            assert ("sumfactApplyKernel" if name == "generic" else "sumfactFastKernel") in mf.route(), mf.route()
            Y = torch.zeros((1, part.n_local_nodes * U), dtype=torch.float64, device="cuda")
            mf.apply(torch.as_tensor(x, device="cuda"), Y)
            out[name] = Y.cpu().numpy()
        assert helpers.rel_err(out[name].T, want) < 1e-12, name
    assert not np.array_equal(out["generic"], out["fast"])  # different kernels: different rounding


@pytest.mark.parametrize("p", [2, 4, 6])
def test_ghost_rows_behind_the_owned_rows(S, ctx, p):
    """The element kernel comes in two variants: with an owned-or-ghost select per node (ghost rows in buffers of their
    own, the reference's import / export buffers) and without (no ghost buffers, or ghost rows directly behind the owned
    rows).  A rank with ghosts, all elements: both layouts must give the same result, and static batch distribution the
    same as dynamic."""
    U = 4
    part = S.CubePartition((4, 4, 2), p, parts=(2, 2, 1), rank=3, perturb=0.1)
    assert part.n_ghost_nodes > 0
    mesh = S.DeviceMesh(ctx, part, U, part.dirichlet_mask(U))
    mf = S.MatrixFreeSystem(mesh, S.KERNEL_DIFFUSION3D, [0.7, 1.0])
    n_owned, n_ghost = part.n_owned_nodes * U, part.n_ghost_nodes * U
    x = torch.as_tensor(part.synthetic_vector(U), device="cuda")  # (1, n_local_dofs): ghost values included
    # separate ghost buffers
    X, XG = x[:, :n_owned].clone(), x[:, n_owned:].clone()
    Y = torch.zeros((1, n_owned), dtype=torch.float64, device="cuda")
    YG = torch.zeros((1, n_ghost), dtype=torch.float64, device="cuda")
    mf.apply_elems(2, X, XG, Y, YG, 1.5, 0.0)
    # one allocation: ghost rows directly behind the owned rows
    xc = x.clone()
    yc = torch.zeros_like(xc)
    mf.apply_elems(2, xc[:, :n_owned], xc[:, n_owned:], yc[:, :n_owned], yc[:, n_owned:], 1.5, 0.0)
    torch.cuda.synchronize()
    scale = float(Y.abs().max())
    assert float((yc[:, :n_owned] - Y).abs().max()) < 1e-12 * scale
    assert float((yc[:, n_owned:] - YG).abs().max()) < 1e-12 * scale
    assert float(YG.abs().max()) > 0.0


def test_static_deal_fallback():
    """L3K_FAST_STATIC=1 in the environment of a fresh process (the environment initialises the context's l3k_tuning once, at
    l3k_ctx_create): the static deal of the batches instead of the per-XCD counters still agrees with the oracle."""
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "static_deal_check.py")
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=300, env=dict(os.environ, L3K_FAST_STATIC="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "static deal ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("p,ne", [(6, 12), (4, 14), (2, 6)])
def test_deterministic_mode_is_bitwise_reproducible(p, ne):
    """SURVEY.md 5 (race detection): the reference's scatter uses relaxed atomics (algsys/MatrixFreeSystem.hpp:513) and its
    global results are not bitwise reproducible; this build's deterministic mode (l3k_ctx_set_deterministic, element
    launches colour by colour) is.  Two applies of the same x are bitwise equal, agree with the atomic mode to rounding
    and with the oracle; diag / rhs likewise; and a PCG solve takes exactly the same number of iterations and ends in
    exactly the same vector in three runs.  (Orders 6 and 4 at sizes that run the one-wave kernel, order 2 the generic.)"""
    import torch
    import oracle_lib as O
    from helpers import oracle_mesh, rel_err
    from l3ster_amd import solve, system
    torch.cuda.set_device(0)
    U, kid, kpar = 4, system.KERNEL_DIFFUSION3D, [1.0, 1.0]
    part = system.CubePartition(ne, p, perturb=0.1)
    mask = part.dirichlet_mask(U)
    ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
    ctx.set_deterministic(True)
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, U, mask), kid, kpar)
    x = system.synthetic_vector_torch(part.node_grid_id, U, "cuda", seed=3)
    ys = []
    for _ in range(3):
        y = torch.full_like(x, 0.25)
        mf.apply(x, y, 1.5, -0.5)
        ys.append(y)
    torch.cuda.synchronize()
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])
    ctx2 = system.Context(0, torch.cuda.current_stream().cuda_stream)  # the atomic mode on the same mesh
    mf2 = system.MatrixFreeSystem(system.DeviceMesh(ctx2, part, U, mask), kid, kpar)
    y2 = torch.full_like(x, 0.25)
    mf2.apply(x, y2, 1.5, -0.5)
    assert (ys[0] - y2).norm().item() < 1e-13 * y2.norm().item()
    if part.n_elems <= 3000:
        om = oracle_mesh(part, p + 1, U, np.arange(U), mask)
        y_ref = O.mf_apply(om, kid, x.cpu().numpy().T, np.full((x.shape[1], 1), 0.25, order="F"), alpha=1.5, beta=-0.5, kparams=kpar, nthreads=8)
        assert rel_err(ys[0].cpu().numpy()[0], y_ref[:, 0]) < 1e-11
    runs = []
    for _ in range(3):
        diag, rhs = mf.diag_rhs(None)
        sol = torch.zeros_like(diag)
        res = solve.pcg(mf, rhs[0], sol, solve.jacobi_inverse_native(ctx, diag), tol=1e-9, residual_scaling="rhs", max_iters=4000)
        runs.append((res.num_iters, diag.clone(), rhs.clone(), sol.clone()))
    torch.cuda.synchronize()
    for r in runs[1:]:
        assert r[0] == runs[0][0]
        assert torch.equal(r[1], runs[0][1]) and torch.equal(r[2], runs[0][2]) and torch.equal(r[3], runs[0][3])


def test_deterministic_mode_with_boundary_terms():
    """Boundary equation kernels in deterministic mode: the element sides are coloured like the elements (sides of one colour
    share no node) and launched colour by colour behind the domain kernel.  Applies and diag / rhs of a system with an
    attached boundary term are bitwise equal run to run and agree with the oracle."""
    import torch
    import oracle_lib as O
    from helpers import oracle_mesh, rel_err
    from l3ster_amd import system
    torch.cuda.set_device(0)
    p, ne, U, kid, kpar = 2, 5, 4, system.KERNEL_DIFFUSION3D, [1.0, 1.0]
    part = system.CubePartition(ne, p, perturb=0.1)
    mask = part.dirichlet_mask(U, sides=(4, 5))
    ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
    ctx.set_deterministic(True)
    mesh = system.DeviceMesh(ctx, part, U, mask)
    mf = system.MatrixFreeSystem(mesh, kid, kpar)
    fe, fs = part.boundary_sides([0, 1, 2, 3])  # four walls: sides of corner elements share edges
    mf.attach_boundary(system.BoundaryTerm(mesh, system.KERNEL_ADIABATIC3D, fe, fs))
    x = system.synthetic_vector_torch(part.node_grid_id, U, "cuda", seed=3)
    ys, drs = [], []
    for _ in range(3):
        y = torch.full_like(x, 0.25)
        mf.apply(x, y, 1.5, -0.5)
        ys.append(y)
        drs.append(tuple(t.clone() for t in mf.diag_rhs(None)))
    torch.cuda.synchronize()
    for k in (1, 2):
        assert torch.equal(ys[0], ys[k]) and torch.equal(drs[0][0], drs[k][0]) and torch.equal(drs[0][1], drs[k][1])
    om = oracle_mesh(part, p + 1, U, np.arange(U), mask)
    xh = x.cpu().numpy()
    want = O.mf_apply(om, kid, xh.T, np.full((xh.shape[1], 1), 0.25, order="F"), alpha=1.5, beta=-0.5, kparams=kpar)
    O.bnd_apply(om, O.KERNEL_ADIABATIC3D, fe, fs, np.asfortranarray(xh.T), want, alpha=1.5)
    assert rel_err(ys[0].cpu().numpy().T, want) < 1e-12
    wd, wr = O.mf_diag_rhs(om, kid, kparams=kpar, finalize=False)
    O.bnd_diag_rhs(om, O.KERNEL_ADIABATIC3D, fe, fs, wd, wr)
    wd[mask != 0] = 1.0
    wr[mask != 0] = 0.0
    assert rel_err(drs[0][0].cpu().numpy(), wd) < 1e-12 and rel_err(drs[0][1].cpu().numpy().T, wr) < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("ne", [(3, 3, 3), (4, 3, 1)])
def test_deterministic_mode_with_derivative_boundary_kernel(ne):
    """A boundary kernel that fills A1..A3 scatter-adds over EVERY node of its element (the normal derivative couples all of
    them), so two sides may share a launch only if their elements share no node.  All six cube sides, and a mesh one element
    thick (every element carries two to five sides): applies and diag / rhs bitwise equal run to run and equal to the oracle."""
    import torch
    import oracle_lib as O
    from helpers import oracle_mesh, rel_err
    from l3ster_amd import system
    torch.cuda.set_device(0)
    p, U, kp = 2, 4, [1.5, 0.4, 0.3]
    part = system.CubePartition(ne, p, perturb=0.1)
    mask = np.zeros(part.n_local_nodes * U, dtype=np.uint8)
    ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
    ctx.set_deterministic(True)
    mesh = system.DeviceMesh(ctx, part, U, mask)
    fe, fs = part.boundary_sides([0, 1, 2, 3, 4, 5])
    term = system.BoundaryTerm(mesh, system.KERNEL_NORMALFLUX3D, fe, fs, kernel_params=kp)
    x = system.synthetic_vector_torch(part.node_grid_id, U, "cuda", seed=5)
    ys, drs = [], []
    for _ in range(4):
        y = torch.full_like(x, 0.125)
        term.apply(x, y, alpha=0.75)
        diag = torch.zeros(mask.size, dtype=torch.float64, device="cuda")
        rhs = torch.zeros((1, mask.size), dtype=torch.float64, device="cuda")
        term.diag_rhs(diag, rhs)
        ys.append(y)
        drs.append((diag, rhs))
    torch.cuda.synchronize()
    for k in (1, 2, 3):
        assert torch.equal(ys[0], ys[k]) and torch.equal(drs[0][0], drs[k][0]) and torch.equal(drs[0][1], drs[k][1])
    om = oracle_mesh(part, p + 1, U, np.arange(U), mask)
    xh = x.cpu().numpy()
    want = np.full((xh.shape[1], 1), 0.125, order="F")
    O.bnd_apply(om, O.KERNEL_NORMALFLUX3D, fe, fs, np.asfortranarray(xh.T), want, alpha=0.75, kparams=kp)
    assert rel_err(ys[0].cpu().numpy().T, want) < 1e-12
    wd, wr = np.zeros(mask.size), np.zeros((mask.size, 1), order="F")
    O.bnd_diag_rhs(om, O.KERNEL_NORMALFLUX3D, fe, fs, wd, wr, kparams=kp)
    assert rel_err(drs[0][0].cpu().numpy(), wd) < 1e-12 and rel_err(drs[0][1].cpu().numpy().T, wr) < 1e-11


@pytest.mark.gpu
def test_bench_gpus_2_as_typed():
    """`python bench.py --gpus 2 ...` typed WITHOUT a launcher (it used to exit with "launch with torch.distributed.run"):
    the script starts its two ranks itself as child processes (l3ster_amd/launch.py), relays rank 0's single JSON line and returns 0.
    On the one-GPU box under L3K_BENCH_REHEARSAL=1 (both ranks on GPU 0, gloo, host-staged messages); the partitioned apply is
    checked inside against one rank's apply on the whole mesh."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["L3K_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--ne", "8", "--steps", "2", "--warmup", "1"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip() and not ln.startswith("[Gloo]")]  # (gloo announces itself on stdout)
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rehearsal"] is True and out["steps"] == 2
    assert out["rehearsal_check"]["xAx_and_yy_vs_one_rank_rel"] < 1e-11
    assert "sumfactFastKernel" in out["roofline"]["kernel"]
    # a failing rank fails the command (and does not leave the other rank waiting in a collective)
    env["L3K_BENCH_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--ne", "8", "--steps", "2", "--warmup", "1"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "rank 1 of 2" in (r.stdout + r.stderr)
