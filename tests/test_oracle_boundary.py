"""Known-answer tests of the oracle's boundary terms and post-processing integrals, re-expressed
from the reference's own tests: boundary normals and side areas (tests/MappingTests.cpp:137-218,555-610) and the
end-to-end 2-D diffusion problem K6 (tests/Diffusion2D.hpp:17-121)."""
import numpy as np
import pytest

import oracle_lib as O

# tests/MappingTests.cpp:25-47
QUAD = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [2, 2, 0]], float)
HEXM = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0, 0, 1], [1, 0, 1.5], [0, 1, 1.5], [1, 1, 2]], float)


def test_boundary_normals_quad():
    # tests/MappingTests.cpp:154-178
    s5 = np.sqrt(5.0)
    for side, pt, expect in [(0, (0, -1), (0, -1)), (1, (0, 1), (-1 / s5, 2 / s5)), (2, (-1, 0), (-1, 0)),
                             (3, (1, 0), (2 / s5, -1 / s5))]:
        n, _ = O.boundary_geometry(2, QUAD, pt, side)
        assert np.allclose(n, expect, atol=1e-13)


def test_boundary_normals_hex():
    # tests/MappingTests.cpp:179-217
    a, b = np.sqrt(1 / 6), np.sqrt(2 / 3)
    for side, pt, expect in [(0, (0, 0, -1), (0, 0, -1)), (1, (0, 0, 1), (-a, -a, b)), (2, (0, -1, 0), (0, -1, 0)),
                             (3, (0, 1, 0), (0, 1, 0)), (4, (-1, 0, 0), (-1, 0, 0)), (5, (1, 0, 0), (1, 0, 0))]:
        n, _ = O.boundary_geometry(3, HEXM, pt, side)
        assert np.allclose(n, expect, atol=1e-13)


def test_side_areas():
    # tests/MappingTests.cpp:555-610 (quadrature order 10 -> 6 points per direction)
    nq = 6
    for side, area in enumerate([1.0, np.sqrt(5.0), 1.0, np.sqrt(5.0)]):
        assert O.integrate_local(side, O.RESIDUAL_UNIT2D, 1, nq, QUAD, None)[0] == pytest.approx(area, abs=1e-14)
    for side, area in enumerate([1.0, np.sqrt(1.5), 1.25, 1.75, 1.25, 1.75]):
        assert O.integrate_local(side, O.RESIDUAL_UNIT3D, 1, nq, HEXM, None)[0] == pytest.approx(area, abs=1e-14)


@pytest.mark.parametrize("dim,p,nq", [(2, 2, 3), (2, 4, 5), (3, 2, 3), (3, 3, 4)])
def test_side_basis_properties(dim, p, nq):
    n = p + 1
    for side in range(2 * dim):
        vals, ders, w, pts = O.side_basis_at_qps(dim, p, nq, side)
        assert np.allclose(vals.sum(axis=1), 1.0, atol=1e-13)  # partition of unity on the side
        assert np.allclose(ders.sum(axis=2), 0.0, atol=1e-11)
        assert w.sum() == pytest.approx(2.0 ** (dim - 1), abs=1e-13)
        # points lie on the side (mesh/ElementTraits.hpp:84-95: hex z-,z+,y-,y+,x-,x+; quad y-,y+,x-,x+)
        axis = (dim - 1) - side // 2
        assert np.allclose(pts[:, axis], -1.0 if side % 2 == 0 else 1.0, atol=1e-15)
        assert np.all(np.abs(pts) <= 1 + 1e-15)
        # only the nodes of the side have non-zero VALUES there
        idx = np.arange(n ** dim)
        coord = (idx // n ** axis) % n
        on_side = coord == (0 if side % 2 == 0 else p)
        assert np.abs(vals[:, ~on_side]).max() < 1e-13
        # the set of side points equals the tensor Gauss rule on the two tangential axes
        x, _ = O.gl_rule(nq)
        tang = [a for a in range(dim) if a != axis]
        got = sorted(map(tuple, np.round(pts[:, tang], 12)))
        want = sorted(map(tuple, np.round(np.array(np.meshgrid(*[x] * (dim - 1))).reshape(dim - 1, -1).T, 12)))
        assert got == want


def _square_mesh(ne, p):
    """makeSquareMesh(linspace(0,1,ne+1)) at order p (mesh/primitives/SquareMesh.hpp): lexicographic GLL node grid,
    boundary ids bottom 1, top 2, left 3, right 4 -> sides 0, 1, 2, 3 of the adjacent elements."""
    gll = O.gll_nodes(p + 1)
    n1 = ne * p + 1
    xs = np.concatenate([[0.0]] + [(e + (gll[1:] + 1) / 2) / ne for e in range(ne)])
    elem_nodes, elem_verts = [], []
    for ey in range(ne):
        for ex in range(ne):
            ids = [(ey * p + j) * n1 + (ex * p + i) for j in range(p + 1) for i in range(p + 1)]
            elem_nodes.append(ids)
            elem_verts.append([[(ex + i) / ne, (ey + j) / ne, 0.0] for j in range(2) for i in range(2)])
    X, Y = np.meshgrid(xs, xs)  # node (iy, ix)
    coords = np.stack([X.ravel(), Y.ravel()], axis=1)
    faces = {1: [(ex, 0) for ex in range(ne)], 2: [((ne - 1) * ne + ex, 1) for ex in range(ne)],
             3: [(ey * ne, 2) for ey in range(ne)], 4: [(ey * ne + ne - 1, 3) for ey in range(ne)]}
    return np.array(elem_nodes, np.uint32), np.array(elem_verts, float), coords, faces


@pytest.mark.parametrize("ne", [4, 32])
def test_k6_diffusion2d_end_to_end(ne):
    """K6 (tests/Diffusion2D.hpp): ne x ne quads of order 2 on [0,1]^2, first-order diffusion system (T, qx, qy),
    Dirichlet T = x on left/right, adiabatic q.n = 0 boundary kernel on top/bottom; the discrete solution reproduces
    T = x, q = (1, 0): L2 error of (T - x, dT/dx - 1, dT/dy) < 1e-8 on the domain and on the boundary.  ne = 4 is the
    reference's test; ne = 32 is BASELINE.json configs[0] as stated (examples/02-diffusion-2D, quad mesh 32 x 32, order 2,
    assembled path on the CPU: 12 675 dofs), assembled sparse and solved directly."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    p, U = 2, 3
    nq = O.n_qps1d(p, 1, 0)
    elem_nodes, elem_verts, coords, faces = _square_mesh(ne, p)
    n_nodes = coords.shape[0]
    nd = n_nodes * U
    assert (ne, n_nodes, nd) in ((4, 81, 243), (32, 4225, 12675))
    rows, cols, vals = [], [], []
    F = np.zeros(nd)

    def add(nodes, Ke, Fe):
        dofs = (nodes.astype(np.int64)[:, None] * U + np.arange(U)[None, :]).ravel()
        rows.append(np.repeat(dofs, len(dofs)))
        cols.append(np.tile(dofs, len(dofs)))
        vals.append(Ke.ravel())
        F[dofs] += Fe[:, 0]

    for e in range(ne * ne):
        add(elem_nodes[e], *O.assemble_local(O.KERNEL_DIFFUSION2D, p, nq, 1, elem_verts[e]))
    for bid in (1, 2):  # adiabatic: bottom, top
        for e, side in faces[bid]:
            add(elem_nodes[e], *O.assemble_local_side(side, O.KERNEL_ADIABATIC2D, p, nq, 1, elem_verts[e]))
    K = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nd, nd)).tocsr()
    assert abs(K - K.T).max() < 1e-13
    # Dirichlet T = x on left/right (unknown 0)
    dir_nodes = np.where((coords[:, 0] == 0.0) | (coords[:, 0] == 1.0))[0]
    dir_dofs = dir_nodes * U
    g = coords[dir_nodes, 0]
    free = np.setdiff1d(np.arange(nd), dir_dofs)
    x = np.zeros(nd)
    x[dir_dofs] = g
    x[free] = spla.spsolve(K[free][:, free].tocsc(), F[free] - K[free][:, dir_dofs] @ g)
    sol = x.reshape(n_nodes, U)
    assert np.abs(sol[:, 0] - coords[:, 0]).max() < 1e-9

    fields = np.ascontiguousarray(sol.T)  # SoA [F][n_nodes]
    mesh = O.MeshView(2, p, nq, elem_nodes, elem_verts, n_nodes, U, [0, 1, 2], fields=fields)
    nq2 = O.n_qps1d(p, 2, 0)  # computeNormL2 doubles the quadrature orders (post/NormL2.hpp:11-19)
    err = np.sqrt(O.mf_integrate(mesh, O.RESIDUAL_LINEAR2D_ERROR, nq2, square=True))
    fe = [e for b in (2, 1, 3, 4) for e, _ in faces[b]]
    fs = [s for b in (2, 1, 3, 4) for _, s in faces[b]]
    berr = np.sqrt(O.mf_integrate(mesh, O.RESIDUAL_LINEAR2D_ERROR, nq2, square=True, face_elem=fe, face_side=fs))
    assert np.linalg.norm(err) < 1e-8 and np.linalg.norm(berr) < 1e-8  # tests/Diffusion2D.hpp:117-119
    # the perimeter through the same path: a sanity check that all 4 * ne sides were visited once
    per = O.mf_integrate(mesh, O.RESIDUAL_UNIT2D, nq2, face_elem=fe, face_side=fs)
    assert per[0] == pytest.approx(4.0, abs=1e-13)


def test_boundary_matrix_free_consistency():
    """orc_bnd_apply / orc_bnd_diag_rhs on a small distorted hex mesh == dense side matrices."""
    rng = np.random.default_rng(5)
    p, U = 2, 4
    nq = O.n_qps1d(p, 1, 0)
    import helpers
    verts = helpers.HEX
    N = (p + 1) ** 3
    elem_nodes = np.arange(N, dtype=np.uint32).reshape(1, N)
    mask = np.zeros(N * U, np.uint8)
    mask[rng.choice(N, 6, replace=False) * U] = 1
    mesh = O.MeshView(3, p, nq, elem_nodes, verts.reshape(1, 8, 3), N, U, [0, 1, 2, 3], dirichlet=mask)
    kp = [2.0, 0.7]
    x = rng.standard_normal(N * U)
    y = np.zeros((N * U, 1), order="F")
    sides = [1, 3, 4]
    O.bnd_apply(mesh, O.KERNEL_ROBIN3D, [0] * 3, sides, x, y, alpha=1.5, kparams=kp)
    Ks = sum(O.assemble_local_side(s, O.KERNEL_ROBIN3D, p, nq, 1, verts, kparams=kp)[0] for s in sides)
    Fs = sum(O.assemble_local_side(s, O.KERNEL_ROBIN3D, p, nq, 1, verts, kparams=kp)[1] for s in sides)
    xm = np.where(mask, 0.0, x)
    want = np.where(mask, 0.0, 1.5 * (Ks @ xm))
    assert np.allclose(y[:, 0], want, rtol=1e-12, atol=1e-12)
    g = np.where(mask, rng.standard_normal(N * U), 0.0)
    diag = np.zeros(N * U)
    rhs = np.zeros((N * U, 1), order="F")
    O.bnd_diag_rhs(mesh, O.KERNEL_ROBIN3D, [0] * 3, sides, diag, rhs, dirichlet_vals=g, kparams=kp)
    assert np.allclose(diag, np.diag(Ks), rtol=1e-12, atol=1e-13)
    assert np.allclose(rhs[:, 0], Fs[:, 0] - Ks @ g, rtol=1e-11, atol=1e-12)


def test_values_at_nodes_dirichlet_kernel():
    """setDirichletBCValues of K6 (tests/Diffusion2D.hpp:49-62): the boundary residual kernel out[0] = x evaluated at the
    nodes of the left / right sides and averaged gives T = x there; nothing else is touched; nodes shared by two boundary
    elements receive two contributions (algsys/ComputeValuesAtNodes.hpp:508-594,112-154)."""
    en, ev, coords, faces = _square_mesh(4, 2)
    U = 3
    mesh = O.MeshView(2, 2, 3, en, ev, coords.shape[0], U, [0, 1, 2])
    fe = [e for b in (3, 4) for e, _ in faces[b]]
    fs = [s for b in (3, 4) for _, s in faces[b]]
    s, c = O.values_at_nodes(mesh, O.RESIDUAL_COORDX2D, [0], fe, fs)
    on = (coords[:, 0] == 0.0) | (coords[:, 0] == 1.0)
    c2, s2 = c.reshape(-1, U), s.reshape(-1, U)
    assert np.all(c2[~on] == 0) and np.all(c2[:, 1:] == 0)
    assert set(np.unique(c2[on, 0])) == {1.0, 2.0}
    assert np.allclose(s2[on, 0] / c2[on, 0], coords[on, 0], atol=1e-15)


def test_values_at_nodes_normal_and_derivatives():
    """A domain evaluation reproduces nodal fields and their gradients: the Linear2D error kernel on the exact fields
    (T = x, dT/dx = 1, dT/dy = 0 as nodal fields) vanishes at every node; visit counts are 1, 2 or 4 per node."""
    en, ev, coords, _ = _square_mesh(3, 3)
    n = coords.shape[0]
    fields = np.stack([coords[:, 0], np.ones(n), np.zeros(n)])
    mesh = O.MeshView(2, 3, 4, en, ev, n, 3, [0, 1, 2], fields=fields)
    s, c = O.values_at_nodes(mesh, O.RESIDUAL_LINEAR2D_ERROR, [0, 1, 2])
    assert np.abs(s).max() < 1e-14 and set(np.unique(c)) == {1.0, 2.0, 4.0}
