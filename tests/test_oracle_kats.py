"""Pins the CPU oracle (oracle/oracle.cpp) to the reference's own known-answer tests, re-expressed (SURVEY.md §8c
K1..K5), and to the golden fixtures from the independent numpy/mpmath restatement.  CPU only."""
import numpy as np
import pytest

import oracle_lib as O

# tests/LocalOperatorCommon.hpp:17-59
QUAD = np.array([[1, 1, 0], [2, 1, 0], [1, 3, 0], [3, 4, 0]], float)
HEX = np.array([[1, 1, 0], [2, 1, 0], [1, 3, 0], [3, 4, 0], [1, 1, 1], [2, 1, 1.5], [1, 3, 2], [3, 4, 3.5]], float)


def boundary_nodes(dim, p):
    n = p + 1
    return np.array([b for b in range(n ** dim) if any(((b // n ** a) % n) in (0, n - 1) for a in range(dim))])


def apply_dirichlet(A, b, phi, bnd, U):
    """tests/LocalOperatorCommon.hpp:190-205"""
    dofs = bnd * U
    b = b - A[:, dofs] @ phi[bnd]
    b[dofs] = phi[bnd]
    A = A.copy()
    A[dofs, :] = 0.0
    A[:, dofs] = 0.0
    A[dofs, dofs] = 1.0
    return A, b


# ------------------------------------------------------------------------------------------------ K5: tables
def test_gll_closed_forms():
    """tests/MathTests.cpp:168-214"""
    assert O.gll_nodes(2).tolist() == [-1.0, 1.0]
    assert O.gll_nodes(3).tolist() == [-1.0, 0.0, 1.0]
    a = 0.2 * np.sqrt(5.0)
    np.testing.assert_allclose(O.gll_nodes(4), [-1, -a, a, 1], atol=1e-14, rtol=0)
    a = np.sqrt(21.0) / 7.0
    np.testing.assert_allclose(O.gll_nodes(5), [-1, -a, 0, a, 1], atol=1e-14, rtol=0)
    a14 = np.sqrt((7.0 + 2 * np.sqrt(7.0)) / 21.0)
    a23 = np.sqrt((7.0 - 2 * np.sqrt(7.0)) / 21.0)
    np.testing.assert_allclose(O.gll_nodes(6), [-1, -a14, -a23, a23, a14, 1], atol=1e-14, rtol=0)


def test_gauss_legendre_small_rules():
    """tests/QuadratureTests.cpp:10-61; size rule quad/ReferenceQuadrature.hpp:13-22"""
    x, w = O.gl_rule(1)
    assert abs(x[0]) < 1e-10 and abs(w[0] - 2) < 1e-10
    x, w = O.gl_rule(2)
    np.testing.assert_allclose(x, [-0.57735026919, 0.57735026919], atol=1e-10)
    np.testing.assert_allclose(w, [1, 1], atol=1e-10)
    x, w = O.gl_rule(3)
    np.testing.assert_allclose(x, [-0.77459666924, 0, 0.77459666924], atol=1e-10)
    np.testing.assert_allclose(w, [0.55555555556, 0.88888888889, 0.55555555556], atol=1e-10)
    assert O.n_qps1d(6) == 7 and O.n_qps1d(4) == 5 and O.n_qps1d(3, 2) == 7 and O.n_qps1d(4, 1, 1) == 8


@pytest.mark.parametrize("nq", range(1, 10))
def test_gl_exactness(nq):
    """tests/QuadratureTests.cpp (polynomial exactness): an nq-point rule integrates x^k exactly for k <= 2nq-1"""
    x, w = O.gl_rule(nq)
    for k in range(2 * nq):
        exact = 0.0 if k % 2 else 2.0 / (k + 1)
        assert abs(np.dot(w, x ** k) - exact) < 1e-13


def test_jacobi_matrix_entries():
    """tests/MappingTests.cpp:99-135 (elements :25-47)"""
    quad = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [2, 2, 0]], float)
    J = O.jacobi_mat(2, quad, [0.5, 0.5])
    np.testing.assert_allclose(J, [[7 / 8, 3 / 8], [3 / 8, 7 / 8]], atol=1e-13)
    hexa = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0, 0, 1], [1, 0, 1.5], [0, 1, 1.5], [1, 1, 2]], float)
    J = O.jacobi_mat(3, hexa, [0.5, 0.5, 0.5])
    np.testing.assert_allclose(J, [[0.5, 0, 3 / 16], [0, 0.5, 3 / 16], [0, 0, 7 / 8]], atol=1e-13)


def test_map_to_physical():
    """tests/MappingTests.cpp:49-96"""
    quad = np.array([[1, -1, 0], [2, -1, 0], [1, 1, 1], [2, 1, 1]], float)
    np.testing.assert_allclose(O.map_to_physical(2, quad, [0.5, -0.5]), [1.75, -0.5, 0.25], atol=1e-15)
    hexa = np.array([[.5, .5, .5], [1, .5, .5], [.5, 1, .5], [1, 1, .5], [.5, .5, 1], [1, .5, 1], [.5, 1, 1], [1, 1, 1]])
    np.testing.assert_allclose(O.map_to_physical(3, hexa, [0, 0, 0]), [0.75, 0.75, 0.75], atol=1e-15)


def test_reference_basis_partition_of_unity():
    """tests/MappingTests.cpp:405-427: hex p=4, QO=4 (3 points): sum phi = 1, sum d phi = 0 at every QP"""
    vals, ders, w, pts = O.ref_basis_at_qps(3, 4, 3)
    np.testing.assert_allclose(vals.sum(axis=1), 1.0, atol=1e-13)
    np.testing.assert_allclose(ders.sum(axis=2), 0.0, atol=1e-13)
    assert abs(w.sum() - 8.0) < 1e-13
    # QP ordering: xi slowest (quad/GenerateQuadrature.hpp:64-71)
    x1, _ = O.gl_rule(3)
    assert np.allclose(pts[1], [x1[0], x1[0], x1[1]]) and np.allclose(pts[9], [x1[1], x1[0], x1[0]])


def test_tables_vs_mpmath_golden(golden):
    g = golden("tables")
    for p in range(1, 9):
        np.testing.assert_allclose(O.gll_nodes(p + 1), g[f"gll_{p}"], atol=1e-15, rtol=0)
        for nq in sorted({p + 1, 2 * p + 1}):
            x, w = O.gl_rule(nq)
            np.testing.assert_allclose(x, g[f"qx_{nq}"], atol=2e-16, rtol=0)
            np.testing.assert_allclose(w, g[f"qw_{nq}"], atol=1e-15, rtol=0)
            I, D = O.basis_1d(p, nq)
            np.testing.assert_allclose(I, g[f"I_{p}_{nq}"], atol=2e-15, rtol=0)
            np.testing.assert_allclose(D, g[f"D_{p}_{nq}"], atol=1e-13, rtol=0)


# ------------------------------------------------------------------------------------------------ K1
@pytest.mark.parametrize("dim,kid,p,verts,U", [(2, O.KERNEL_DIFFUSION2D, 4, QUAD, 3), (3, O.KERNEL_DIFFUSION3D, 3, HEX, 4)])
def test_k1_single_element_least_squares_solve(dim, kid, p, verts, U):
    """tests/LocalAssemblyTests.cpp:3-43: assemble, Dirichlet on unknown 0 at boundary nodes with phi = x_d (dim rhs),
    dense solve reproduces phi at ALL nodes, rel 1e-6."""
    nq = O.n_qps1d(p, value_order=2)
    R = dim
    K, F = O.assemble_local(kid, p, nq, R, verts, kparams=[1.0, 0.0] if kid == O.KERNEL_DIFFUSION3D else None)
    N = (p + 1) ** dim
    phi = np.array([O.node_location(dim, p, verts, n)[:dim] for n in range(N)])  # makeSolution, :207-219
    A, b = apply_dirichlet(K, np.array(F), phi, boundary_nodes(dim, p), U)
    x = np.linalg.solve(A, b)
    np.testing.assert_allclose(x[::U], phi, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(K, K.T, atol=0)
    assert np.linalg.eigvalsh(K).min() > -1e-10  # least-squares operator is PSD


# ------------------------------------------------------------------------------------------------ K2
@pytest.mark.parametrize("dim,kid,p,verts,U,R", [(2, O.KERNEL_DIFFUSION2D, 4, QUAD, 3, 2), (3, O.KERNEL_DIFFUSION3D, 3, HEX, 4, 3)])
def test_k2_local_operator_equals_matrix(dim, kid, p, verts, U, R):
    """tests/LocalOperatorTests.cpp:3-95"""
    nq = O.n_qps1d(p, value_order=2)
    kpar = [1.0, 0.0] if kid == O.KERNEL_DIFFUSION3D else None
    K, F = O.assemble_local(kid, p, nq, R, verts, kparams=kpar)
    N = (p + 1) ** dim
    phi = np.zeros((N, R))
    phi[:, :dim] = np.array([O.node_location(dim, p, verts, n)[:dim] for n in range(N)])
    bnd = boundary_nodes(dim, p)
    dofs = bnd * U
    A, b = apply_dirichlet(K, np.array(F), phi, bnd, U)
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, (N * U, R))
    x_bc = x.copy()
    x_bc[dofs] = 0.0
    diag, rhs = O.diag_rhs_local(kid, p, nq, R, verts, dofs, phi[bnd], kparams=kpar)
    y = O.apply_local(kid, p, nq, verts, x_bc, kparams=kpar)
    y[dofs] = x[dofs]
    diag[dofs] = 1.0
    rhs[dofs] = phi[bnd]
    assert np.linalg.norm(y - A @ x) < 1e-8
    assert np.linalg.norm(np.diag(A) - diag) < 1e-8
    assert np.linalg.norm(rhs - b) < 1e-8


# ------------------------------------------------------------------------------------------------ K3
@pytest.mark.parametrize("dim,kid,p,verts,U", [(2, O.KERNEL_DIFFUSION2D_VAR, 4, QUAD, 3), (3, O.KERNEL_DIFFUSION3D_VAR, 3, HEX, 4)])
@pytest.mark.parametrize("odd_even", [False, True])
def test_k3_sumfact_equals_local_element(dim, kid, p, verts, U, odd_even):
    """tests/SumFactorizationTests.cpp:3-53: variable-coefficient kernel (F=1), 2 rhs, random field in U(-1,1)"""
    nq = O.n_qps1d(p, value_order=2)
    N = (p + 1) ** dim
    rng = np.random.default_rng(5)
    nf = rng.uniform(-1, 1, (N, 1))
    x = rng.uniform(-1, 1, (N * U, 2))
    y_le = O.apply_local(kid, p, nq, verts, x, nf)
    y_sf = O.apply_sumfact(kid, p, nq, verts, x, nf, odd_even=odd_even)
    assert np.linalg.norm(y_le - y_sf) < 1e-8
    assert np.linalg.norm(y_le - y_sf) < 1e-11 * np.linalg.norm(y_le)  # the build's own, tighter bar


# ------------------------------------------------------------------------------------------------ K4
@pytest.mark.parametrize("EO", [3, 4])
@pytest.mark.parametrize("QO", [3, 4])
def test_k4_odd_even_equals_standard(EO, QO):
    """tests/SumFactorizationTests.cpp:55-129: 33 columns, (EO,QO) in {3,4}^2; SumFactParams.quad_order = QO ->
    n_qps1d = QO/2+1"""
    nq = QO // 2 + 1
    rng = np.random.default_rng(EO * 10 + QO)
    err = O.oddeven_check(EO, nq, 33, rng.uniform(-1, 1, (EO + 1) * 33), rng.uniform(-1, 1, nq * 33))
    assert err.max() < 1e-8 and err.max() < 1e-13


# ------------------------------------------------------------------------------------------------ golden fixtures
CASES = ["hex_p3_diff", "hex_p3_var", "quad_p4_diff", "quad_p4_var", "hex_p4_diff", "hex_p6_diff", "hex_p4_advdiff",
         "hex_p2_advdiff",
         # round 4: kernels reading point.space / point.time (the sum-factorised path is checked with the true z here; the
         # reference's z = 0 of SumFactorization.hpp:732 in test_reference_z0_is_what_the_sumfact_path_passes), odd U, NS3D
         "hex_p2_point", "hex_p4_point", "hex_p4_advection", "hex_p2_divcurl", "hex_p2_ns3d"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_vs_golden(golden, name):
    g = golden(name)
    kid, p, nq, R = int(g["kid"]), int(g["p"]), int(g["nq"]), int(g["R"])
    nf = g.get("node_fields")
    kpar = g.get("kparams")
    scale = np.linalg.norm(g["y"])
    t = float(g.get("time", 0.0))
    for fn in (lambda: O.apply_local(kid, p, nq, g["verts"], g["x"], nf, kpar, time=t),
               lambda: O.apply_sumfact(kid, p, nq, g["verts"], g["x"], nf, kpar, time=t, true_z=True),
               lambda: O.apply_sumfact(kid, p, nq, g["verts"], g["x"], nf, kpar, time=t, odd_even=True, true_z=True)):
        assert np.linalg.norm(fn() - g["y"]) < 1e-12 * scale
    diag, rhs = O.diag_rhs_local(kid, p, nq, R, g["verts"], g["dir_inds"], g["dir_vals"], nf, kpar, time=t)
    np.testing.assert_allclose(diag, g["diag"], rtol=1e-12, atol=1e-13)
    assert np.linalg.norm(rhs - g["rhs_lifted"]) < 1e-12 * max(1.0, np.linalg.norm(g["rhs_lifted"]))
    if "K" in g or p <= 4:
        K, F = O.assemble_local(kid, p, nq, R, g["verts"], nf, kpar, time=t)
        np.testing.assert_allclose(F, g["F"], rtol=0, atol=1e-12 * max(1.0, np.abs(g["F"]).max()))
        if "K" in g:
            np.testing.assert_allclose(K, g["K"], rtol=0, atol=1e-12 * np.abs(g["K"]).max())


def test_reference_z0_is_what_the_sumfact_path_passes(golden):
    """evalAtHexQPs hands the kernel Point{x, y, 0.} (algsys/SumFactorization.hpp:732), the local-element path the
    true point (algsys/AssembleLocalSystem.hpp:229-230).  The oracle's switch reproduces both: with the reference's z = 0 a
    z-reading kernel gives ANOTHER operator than the local-element path; on an element lying in z in [0, ...] whose kernel
    does not read z the two agree.  (Element-level statement; the mesh-level switch is orc_set_reference_z0.)"""
    g = golden("hex_p2_point")
    kid, p, nq, t = int(g["kid"]), int(g["p"]), int(g["nq"]), float(g["time"])
    y_true = O.apply_sumfact(kid, p, nq, g["verts"], g["x"], None, g["kparams"], time=t, true_z=True)
    y_z0 = O.apply_sumfact(kid, p, nq, g["verts"], g["x"], None, g["kparams"], time=t, true_z=False)
    assert np.linalg.norm(y_true - g["y"]) < 1e-12 * np.linalg.norm(g["y"])
    assert np.linalg.norm(y_z0 - g["y"]) > 1e-3 * np.linalg.norm(g["y"])  # the divergence is not a rounding effect
    # z = 0 equals the true-z operator of a kernel evaluated on the element's footprint in the plane z = 0: flatten the
    # element's point dependence by hand -- same element, kernel evaluated at (x, y, 0) -- through the numpy restatement
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_np as P
    pp = P.phys_point
    try:
        P.phys_point = lambda dim, verts, pt: pp(dim, verts, pt) * np.array([1.0, 1.0, 0.0])
        K0, _ = P.assemble(kid, p, nq, 1, g["verts"], None, g["kparams"], t)
    finally:
        P.phys_point = pp
    assert np.linalg.norm(y_z0 - K0 @ g["x"]) < 1e-12 * np.linalg.norm(g["y"])
    # the mesh-level switch
    from helpers import SingleElementMesh, oracle_mesh
    om = oracle_mesh(SingleElementMesh(p, g["verts"]), nq, 4, np.arange(4))
    try:
        O.set_reference_z0(True)
        y_mesh = O.mf_apply(om, kid, g["x"], kparams=g["kparams"], time=t)
    finally:
        O.set_reference_z0(False)
    assert np.linalg.norm(y_mesh - y_z0) < 1e-13 * np.linalg.norm(g["y"])
    assert np.linalg.norm(O.mf_apply(om, kid, g["x"], kparams=g["kparams"], time=t) - y_true) < 1e-13 * np.linalg.norm(g["y"])


def test_update_solution_restatement():
    """MatrixFreeSystem::updateSolution (algsys/MatrixFreeSystem.hpp:1231-1273): dofs sol_inds of column r -> field
    sol_man_inds[i * n_rhs + r]; the reference's index asserts."""
    from helpers import SingleElementMesh, oracle_mesh
    m = oracle_mesh(SingleElementMesh(2, HEX), 3, 3, np.arange(3))
    x = np.arange(27 * 3 * 2, dtype=float).reshape(2, 81).T
    f = np.full((5, 27), -1.0)
    O.update_solution(m, x, [2, 0], f, [3, 1, 0, 2])
    assert np.array_equal(f[3], x[2::3, 0]) and np.array_equal(f[1], x[2::3, 1])
    assert np.array_equal(f[0], x[0::3, 0]) and np.array_equal(f[2], x[0::3, 1]) and np.all(f[4] == -1.0)
    with pytest.raises(RuntimeError, match="Source index out of bounds"):
        O.update_solution(m, x, [3], f, [0, 1])
    with pytest.raises(RuntimeError, match="Destination index out of bounds"):
        O.update_solution(m, x, [1], f, [0, 5])


def test_degenerate_element_is_reported():
    """algsys/AssembleLocalSystem.hpp:249, EvaluateLocalOperator.hpp:229: detJ <= 0 -> error"""
    bad = HEX.copy()
    bad[[0, 1]] = bad[[1, 0]]  # swap two vertices: inverted element
    with pytest.raises(RuntimeError, match="degenerate"):
        O.assemble_local(O.KERNEL_DIFFUSION3D, 2, 3, 1, bad)
