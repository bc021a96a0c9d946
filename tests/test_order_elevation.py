"""Order elevation of an order-1 hex mesh (mesh::convertMeshToOrder, mesh/ConvertMeshToOrder.hpp:51-104).

CPU: the numpy restatement of the device algorithm's specification (oracle/oracle_np.py:elevate_order) against the
reference's own known answers (tests/MeshTests.cpp:244-279: 44745 nodes for its gmsh cube at order 2; (5*2+1)^3 nodes with
contiguous ids for the procedural cube) and against geometry: every node must sit at one physical location whichever
element it is reached through, and different nodes at different locations -- also after every element's local frame has
been rotated at random.  GPU: l3k_elevate_order bit-exact against the restatement, and the matrix-free apply on an
elevated mesh against the structured generator's mesh of the same cube."""
import itertools
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import oracle_np as ONP  # noqa: E402
from l3ster_amd import capi, system  # noqa: E402


def gmsh_cube():
    d = np.load(os.path.join(HERE, "golden", "gmsh_cube_hexes.npz"))
    return d["verts"], d["conn"], int(d["order2_node_count"])


def cube_conn(n):
    idx = lambda i, j, k: i + (n + 1) * (j + (n + 1) * k)
    conn = [[idx(i + a, j + b, k + c) for c in (0, 1) for b in (0, 1) for a in (0, 1)]
            for k in range(n) for j in range(n) for i in range(n)]
    g = np.linspace(0., 1., n + 1)
    verts = np.array([[g[i], g[j], g[k]] for k in range(n + 1) for j in range(n + 1) for i in range(n + 1)])
    return verts, np.array(conn, dtype=np.uint32)


def hex_rotations():
    """Vertex permutations of the 24 proper rotations of the reference hex (local vertex v = i + 2j + 4k)."""
    perms = []
    for axes in itertools.permutations(range(3)):
        for signs in itertools.product((-1, 1), repeat=3):
            R = np.zeros((3, 3), int)
            for r in range(3):
                R[r, axes[r]] = signs[r]
            if round(np.linalg.det(R)) != 1:
                continue
            perm = []
            for v in range(8):
                c = np.array([2 * (v & 1) - 1, 2 * ((v >> 1) & 1) - 1, 2 * (v >> 2) - 1])
                d = R @ c
                perm.append(int((d[0] > 0) + 2 * (d[1] > 0) + 4 * (d[2] > 0)))
            perms.append(perm)
    assert len(perms) == 24
    return np.array(perms)


def rotate_elements(conn, seed):
    rot = hex_rotations()
    pick = np.random.default_rng(seed).integers(0, 24, conn.shape[0])
    return np.take_along_axis(conn, rot[pick], axis=1)


def node_locations(verts, conn, elem_nodes, p):
    """Per (element, local node) physical location from that element's own tri-linear map."""
    g = system.gll_nodes(p + 1)
    l = np.stack([(1 - g) / 2, (1 + g) / 2], axis=1)
    shape = np.einsum("xi,yj,zk->zyxkji", l, l, l).reshape((p + 1) ** 3, 8)
    return np.einsum("nv,evs->ens", shape, verts[conn.astype(np.int64)])


def check_topology(verts, conn, elem_nodes, n_nodes, p):
    loc = node_locations(verts, conn, elem_nodes, p).reshape(-1, 3)
    ids = elem_nodes.reshape(-1).astype(np.int64)
    assert np.array_equal(np.unique(ids), np.arange(n_nodes))  # contiguous, every id used (MeshTests.cpp:256-258)
    lo = np.full((n_nodes, 3), np.inf)
    hi = np.full((n_nodes, 3), -np.inf)
    np.minimum.at(lo, ids, loc)
    np.maximum.at(hi, ids, loc)
    assert np.max(hi - lo) < 1e-12  # one location per node, whichever element it is reached through
    centre = 0.5 * (lo + hi)
    assert np.unique(np.round(centre, 9), axis=0).shape[0] == n_nodes  # no location carries two nodes


def test_restatement_meets_reference_node_counts():
    verts, conn, want = gmsh_cube()
    en, n_nodes, n_nonint = ONP.elevate_order(conn, verts.shape[0], 2)
    assert n_nodes == want == 44745  # tests/MeshTests.cpp:255
    assert n_nodes - n_nonint == conn.shape[0]  # one internal node per hex at order 2
    v2, c2 = cube_conn(5)
    en, n_nodes, _ = ONP.elevate_order(c2, v2.shape[0], 2)
    assert n_nodes == 11 ** 3  # tests/MeshTests.cpp:274-276
    assert np.array_equal(np.unique(en), np.arange(n_nodes))  # :277-278


@pytest.mark.parametrize("p", [1, 2, 3, 5])
def test_restatement_topology_on_unstructured_and_rotated_meshes(p):
    verts, conn, _ = gmsh_cube()
    en, n_nodes, n_nonint = ONP.elevate_order(conn, verts.shape[0], p)
    check_topology(verts, conn, en, n_nodes, p)
    # internal nodes last, contiguous per element, in local lexicographic order (LocalMeshView.hpp:425-458 class)
    n1 = p + 1
    interior = [i + n1 * (j + n1 * k) for k in range(1, p) for j in range(1, p) for i in range(1, p)]
    if interior:
        want = n_nonint + np.arange(conn.shape[0])[:, None] * len(interior) + np.arange(len(interior))[None, :]
        assert np.array_equal(en[:, interior], want)
    rc = rotate_elements(conn, seed=p)
    en_r, n_r, nn_r = ONP.elevate_order(rc, verts.shape[0], p)
    assert (n_r, nn_r) == (n_nodes, n_nonint)
    check_topology(verts, rc, en_r, n_r, p)


def scrambled_mesh(seed, p):
    """A random subset of the gmsh cube's hexes with randomly rotated local frames and randomly relabelled vertices (the
    canonical edge / face frames follow the vertex ids, so relabelling moves every orientation decision)."""
    rng = np.random.default_rng(seed)
    verts, conn, _ = gmsh_cube()
    keep = rng.random(conn.shape[0]) < 0.4
    conn = rotate_elements(conn[keep], seed)
    relabel = rng.permutation(verts.shape[0])
    new_verts = np.empty_like(verts)
    new_verts[relabel] = verts
    return new_verts, relabel[conn.astype(np.int64)].astype(np.uint32)


@pytest.mark.parametrize("seed", range(6))
def test_restatement_topology_on_scrambled_meshes(seed):
    p = 2 + seed % 4
    verts, conn = scrambled_mesh(seed, p)
    en, n_nodes, n_nonint = ONP.elevate_order(conn, verts.shape[0], p)
    # vertices no element of the subset touches keep their ids but carry no element node: check the used ids only
    used = np.unique(en)
    assert used.max() == n_nodes - 1
    loc = node_locations(verts, conn, en, p).reshape(-1, 3)
    ids = en.reshape(-1)
    lo = np.full((n_nodes, 3), np.inf)
    hi = np.full((n_nodes, 3), -np.inf)
    np.minimum.at(lo, ids, loc)
    np.maximum.at(hi, ids, loc)
    assert np.max((hi - lo)[used]) < 1e-12
    assert np.unique(np.round(0.5 * (lo[used] + hi[used]), 9), axis=0).shape[0] == used.size


# ---------------------------------------------------------------------------------------------------------- device
@pytest.fixture(scope="module")
def ctx():
    import torch
    return system.Context(0, torch.cuda.current_stream().cuda_stream)


@pytest.mark.gpu
@pytest.mark.parametrize("p", [1, 2, 3, 6])
def test_device_elevation_is_bit_exact(ctx, p):
    verts, conn, want = gmsh_cube()
    for c in (conn, rotate_elements(conn, seed=10 + p)):
        en, n_nodes, n_nonint = system.elevate_order(ctx, c, verts.shape[0], p)
        en_o, n_o, nn_o = ONP.elevate_order(c, verts.shape[0], p)
        assert (n_nodes, n_nonint) == (n_o, nn_o)
        assert np.array_equal(en.astype(np.int64), en_o)
    if p == 2:
        assert n_nodes == want


@pytest.mark.gpu
def test_device_elevation_edge_cases(ctx):
    en, n_nodes, n_nonint = system.elevate_order(ctx, np.zeros((0, 8), np.uint32), 7, 3)
    assert en.shape == (0, 64) and n_nodes == 7 and n_nonint == 7  # no elements: the vertices remain
    v, c = cube_conn(1)
    en, n_nodes, _ = system.elevate_order(ctx, c, 8, 4)  # a single element: nothing is shared
    assert n_nodes == 125 and np.array_equal(np.sort(en[0]), np.arange(125))
    with pytest.raises(capi.L3KError):
        system.elevate_order(ctx, np.full((1, 8), 9, np.uint32), 8, 2)  # vertex id outside [0, n_vertices)
    with pytest.raises(capi.L3KError):
        system.elevate_order(ctx, c, 8, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("p,rotate", [(2, False), (4, True), (6, True)])
def test_apply_on_elevated_mesh_matches_structured_generator(ctx, p, rotate):
    """Same cube, same operator, two meshes: the structured generator's and the device-elevated order-1 connectivity (with
    randomly rotated local frames).  x is a function of the node location; y must agree node by node."""
    import torch
    ne, U = 4, 4
    verts, conn = cube_conn(ne)
    if rotate:
        conn = rotate_elements(conn, seed=p)
    mesh_e = system.ElevatedHexMesh(ctx, verts, conn, p)
    part = system.CubePartition(ne, p)
    assert mesh_e.n_owned_nodes == part.n_owned_nodes == (ne * p + 1) ** 3

    def run(mesh):
        xyz = mesh.node_coords()
        x = np.stack([np.sin(3 * xyz[:, 0] + u) * np.cos(2 * xyz[:, 1] - u) + xyz[:, 2] ** 2 for u in range(U)], axis=1)
        on_bnd = np.any((np.abs(xyz) < 1e-12) | (np.abs(xyz - 1) < 1e-12), axis=1)
        mask = np.zeros((mesh.n_local_nodes, U), np.uint8)
        mask[on_bnd, 0] = 1
        dm = system.DeviceMesh(ctx, mesh, U, mask.reshape(-1))
        mf = system.MatrixFreeSystem(dm, system.KERNEL_DIFFUSION3D, [1.0, 1.0])
        X = torch.as_tensor(x.reshape(1, -1), device="cuda")
        Y = torch.zeros_like(X)
        mf.apply(X, Y)
        torch.cuda.synchronize()
        return xyz, Y.cpu().numpy().reshape(-1, U)

    xyz_e, y_e = run(mesh_e)
    xyz_c, y_c = run(part)
    key = lambda xyz: [tuple(r) for r in np.round(xyz * (ne * 64), 6)]  # noqa: E731  (exact grid locations)
    pos = {k: i for i, k in enumerate(key(xyz_c))}
    idx = np.array([pos[k] for k in key(xyz_e)])
    assert np.array_equal(np.sort(idx), np.arange(idx.size))
    err = np.linalg.norm(y_e - y_c[idx]) / np.linalg.norm(y_c)
    assert err < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_device_elevation_on_scrambled_meshes(ctx, seed):
    p = 2 + seed % 4
    verts, conn = scrambled_mesh(seed, p)
    en, n_nodes, n_nonint = system.elevate_order(ctx, conn, verts.shape[0], p)
    en_o, n_o, nn_o = ONP.elevate_order(conn, verts.shape[0], p)
    assert (n_nodes, n_nonint) == (n_o, nn_o)
    assert np.array_equal(en.astype(np.int64), en_o)
