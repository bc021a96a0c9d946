"""Native mesh files (post/NativeIO.hpp:75-108, mesh/MeshUtils.hpp:318-360): the C-ABI writer / reader (host code, no GPU)
against the numpy restatement of the format in oracle/oracle_np.py, in both directions; the single-rank order-1 cube
against the restated makeCubeMesh (same nodes, vertices, element ids and boundary quads as the reference generates); parts
written by several ranks in any order; a partition rebuilt from the file has the halo lists of the one that was saved;
error behaviour (tests/SaveLoadTests.cpp saves and reloads a mesh the same way)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import oracle_np as ONP  # noqa: E402
from l3ster_amd import capi, mesh_file, system  # noqa: E402


def as_restated(fp):
    """MeshFilePart -> the domains dict of ONP.mesh_part_bytes."""
    return {d: [by_type.get(t) for t in mesh_file.TYPES] for d, by_type in fp.domains.items()}


def sorted_by_id(el):
    o = np.argsort(el[2], kind="stable")
    return el[0][o], el[1][o], el[2][o]


def test_writer_matches_format_restatement(tmp_path):
    parts = [system.CubePartition((3, 2, 2), 2, parts=(2, 1, 1), rank=r, perturb=0.1) for r in range(2)]
    fps = [mesh_file.part_of(p) for p in parts]
    sizes = [fp.n_bytes() for fp in fps]
    path = tmp_path / "a.mesh"
    for r in (1, 0):  # any order
        fps[r].save(path, sizes, r, comment="two\nlines")
    blobs = [ONP.mesh_part_bytes(2, as_restated(fp), fp.nodes_begin, fp.n_owned_nodes, fp.boundary_ids) for fp in fps]
    assert [len(b) for b in blobs] == sizes
    assert path.read_bytes() == ONP.mesh_file_bytes(blobs, "two\nlines")  # bit-exact
    comment, restated = ONP.mesh_file_parse(path.read_bytes(), 2)
    assert comment == "two lines" and len(restated) == 2
    for (doms, begin, n_owned, bnd), fp, p in zip(restated, fps, parts):
        assert (begin, n_owned) == (p.global_node_base, p.n_owned_nodes) and list(bnd) == [1, 2, 3, 4, 5, 6]
        assert sorted(doms) == sorted(fp.domains)
        for d in doms:
            for k, t in enumerate(mesh_file.TYPES):
                assert (doms[d][k] is None) == (t not in fp.domains[d])
                if doms[d][k] is not None:
                    for a, b in zip(doms[d][k], fp.domains[d][t]):
                        assert np.array_equal(a, b)


def test_reader_reads_restated_file(tmp_path):
    rng = np.random.default_rng(11)
    order = 3

    def elems(dim, n, id0):
        return (rng.integers(0, 1 << 40, (n, (order + 1) ** dim)).astype(np.uint64), rng.standard_normal((n, 2 ** dim, 3)),
                np.arange(id0, id0 + n, dtype=np.uint64))

    doms0 = {7: [elems(3, 5, 0), elems(2, 2, 5), elems(1, 3, 7)], 2: [None, elems(2, 4, 10), None], 65535: [None, None, None]}
    doms1 = {7: [elems(3, 1, 14), None, None]}
    blobs = [ONP.mesh_part_bytes(order, doms0, 0, 100, [2, 65535]), ONP.mesh_part_bytes(order, doms1, 100, 0, [])]
    path = tmp_path / "b.mesh"
    path.write_bytes(ONP.mesh_file_bytes(blobs, "made by the restatement"))
    assert mesh_file.info(path) == [len(b) for b in blobs]
    p0, p1 = mesh_file.load(path, 0, order), mesh_file.load(path, 1, order)
    assert list(p0.domains) == [2, 7, 65535] and (p0.nodes_begin, p0.n_owned_nodes) == (0, 100)
    assert list(p0.boundary_ids) == [2, 65535] and p1.boundary_ids.size == 0 and (p1.nodes_begin, p1.n_owned_nodes) == (100, 0)
    for d, els in doms0.items():
        for k, t in enumerate(mesh_file.TYPES):
            assert (els[k] is None) == (t not in p0.domains[d])
            if els[k] is not None:
                for a, b in zip(els[k], p0.domains[d][t]):
                    assert np.array_equal(a, b)
    # written back through the C ABI: the same bytes (domains given in any order are stored in ascending id)
    shuffled = mesh_file.MeshFilePart(order, {d: p0.domains[d] for d in (65535, 2, 7)}, 0, 100, [2, 65535])
    out = tmp_path / "b2.mesh"
    shuffled.save(out, [len(b) for b in blobs], 0, "made by the restatement")
    p1.save(out, [len(b) for b in blobs], 1, "made by the restatement", write_header=False)  # same comment: it fixes the offsets
    assert out.read_bytes() == path.read_bytes()
    # loadUnifiedMesh: elements of a domain in part order, boundary ids of the first part that has any
    u = mesh_file.load_unified(path, order)
    assert np.array_equal(u.domains[7]["hex"][2], np.r_[np.arange(5), 14]) and list(u.boundary_ids) == [2, 65535]


@pytest.mark.parametrize("ne", [1, 3])
def test_order1_cube_is_the_reference_cube_mesh(tmp_path, ne):
    """A single-rank order-1 CubePartition saved as a mesh file holds exactly the elements makeCubeMesh(dist) creates:
    node ids, vertices, element ids, boundary quads with their orientation (compared element by element after sorting by
    element id: the traversal order inside a domain is the partition's, not the generator's)."""
    part = system.CubePartition(ne, 1)
    fp = mesh_file.part_of(part)
    path = tmp_path / "cube.mesh"
    fp.save(path, [fp.n_bytes()], 0)
    got = mesh_file.load(path, 0, 1)
    ref, n_nodes = ONP.make_cube_mesh(np.linspace(0.0, 1.0, ne + 1))
    assert (got.nodes_begin, got.n_owned_nodes) == (0, n_nodes) and list(got.boundary_ids) == [1, 2, 3, 4, 5, 6]
    assert sorted(got.domains) == list(range(7))
    # the partition's node numbering at order 1 is a permutation of the generator's x-fastest one: compare through the
    # nodes' coordinates (vertices are stored per element) and through the partition-independent grid id
    gid_of_global = np.empty(n_nodes, dtype=np.int64)
    gid_of_global[mesh_file.local_to_global(part).astype(np.int64)] = part.node_grid_id
    for d in range(7):
        t = "hex" if d == 0 else "quad"
        g_nodes, g_verts, g_ids = sorted_by_id(got.domains[d][t])
        r_nodes, r_verts, r_ids = sorted_by_id(ref[d][0 if d == 0 else 1])
        assert np.array_equal(g_ids, r_ids)
        assert np.allclose(g_verts, r_verts, atol=1e-15)
        assert np.array_equal(gid_of_global[g_nodes.astype(np.int64)], r_nodes.astype(np.int64))


def test_high_order_boundary_quads_lie_on_the_element_sides():
    part = system.CubePartition((2, 3, 2), 3, perturb=0.05)
    fp = mesh_file.part_of(part)
    coords = part.node_coords()
    gid = mesh_file.local_to_global(part).astype(np.int64)
    assert np.array_equal(gid, np.arange(part.n_local_nodes))  # single rank: local == global
    hex_nodes, _, hex_ids = fp.domains[0]["hex"]
    assert np.array_equal(np.sort(hex_ids), np.arange(12))
    n_quads, all_ids = 0, [hex_ids]
    for d, (axis, val) in zip(range(1, 7), ((2, 0.0), (2, 1.0), (1, 0.0), (1, 1.0), (0, 0.0), (0, 1.0))):
        nodes, verts, ids = fp.domains[d]["quad"]
        n_quads += ids.size
        all_ids.append(ids)
        assert nodes.shape[1] == 16 and np.allclose(coords[nodes.astype(np.int64).reshape(-1), axis], val)
        assert np.allclose(verts[:, :, axis], val)
        # corner nodes of the quad's node array sit at the quad's vertices, in order
        assert np.allclose(coords[nodes[:, [0, 3, 12, 15]].astype(np.int64)], verts, atol=1e-14)
    assert n_quads == 2 * (2 * 3 + 2 * 2 + 3 * 2)
    assert np.array_equal(np.sort(np.concatenate(all_ids)), np.arange(12 + n_quads))  # unique ids, generator's range


def test_partition_rebuilt_from_the_file_has_the_saved_halo(tmp_path):
    parts_xyz, order, ne = (2, 2, 1), 2, (4, 4, 2)
    world = 4
    saved = [system.CubePartition(ne, order, parts=parts_xyz, rank=r, perturb=0.1) for r in range(world)]
    fps = [mesh_file.part_of(p) for p in saved]
    sizes = [fp.n_bytes() for fp in fps]
    path = tmp_path / "p.mesh"
    for r in (2, 0, 3, 1):
        fps[r].save(path, sizes, r)
    for r in range(world):
        a, b = saved[r], mesh_file.FilePartition(path, r, order)
        assert (b.n_elems, b.n_interior_elems, b.n_owned_nodes, b.n_ghost_nodes, b.global_node_base, b.n_global_nodes) == \
               (a.n_elems, a.n_interior_elems, a.n_owned_nodes, a.n_ghost_nodes, a.global_node_base, a.n_global_nodes)
        assert np.array_equal(b.ghost_global_id, a.ghost_global_id)
        assert b.nbr_rank == a.nbr_rank and b.ghost_ranges == a.ghost_ranges
        for x, y in zip(a.send_nodes, b.send_nodes):
            assert np.array_equal(x, y)
        # same elements (the order inside the interior / border groups may differ): compare as sets of node tuples
        assert sorted(map(tuple, a.elem_nodes)) == sorted(map(tuple, b.elem_nodes))
        assert np.array_equal(b.dirichlet_mask(1, domain_ids=[1, 6]), a.dirichlet_mask(1, sides=[0, 5]))
        fe, fs = b.boundary_sides([3, 4])
        ae, as_ = a.boundary_sides([2, 3])
        key = lambda part, e, s: sorted((tuple(part.elem_nodes[i]), int(j)) for i, j in zip(e, s))
        assert key(b, fe, fs) == key(a, ae, as_)
    u = mesh_file.load_unified(path, order)
    assert u.domains[0]["hex"][2].size == 32 and np.array_equal(np.sort(u.domains[0]["hex"][2]), np.arange(32))
    assert u.n_owned_nodes == saved[0].n_global_nodes


def test_errors(tmp_path):
    part = system.CubePartition(2, 2)
    fp = mesh_file.part_of(part)
    path = tmp_path / "e.mesh"
    with pytest.raises(capi.L3KError, match="size table"):
        fp.save(path, [fp.n_bytes() + 1], 0)
    with pytest.raises(capi.L3KError):
        fp.save(path, [fp.n_bytes()], 1)  # part index outside the table
    fp.save(path, [fp.n_bytes()], 0)
    with pytest.raises(capi.L3KError, match="part 1 of 1"):
        mesh_file.load(path, 1, 2)
    with pytest.raises(capi.L3KError, match="order 3"):
        mesh_file.load(path, 0, 3)  # the order is a template argument of the reference's loader, not stored
    data = path.read_bytes()
    (tmp_path / "t.mesh").write_bytes(data[:-9])
    with pytest.raises(capi.L3KError, match="truncated"):
        mesh_file.info(tmp_path / "t.mesh")
    (tmp_path / "h.mesh").write_bytes(b"L3STER mesh file\nv1.0\n")
    with pytest.raises(capi.L3KError, match="header"):
        mesh_file.info(tmp_path / "h.mesh")
    with pytest.raises(capi.L3KError):
        mesh_file.info(tmp_path / "missing.mesh")
    dup = mesh_file.MeshFilePart(2, {0: fp.domains[0]}, 0, 1, [])
    d = dup._desc()
    d.n_domains = 2  # the same domain twice
    doms = (capi.MeshFileDomain * 2)(d.domains[0], d.domains[0])
    d.domains = doms
    import ctypes as C
    out = C.c_size_t()
    assert capi.load().l3k_meshfile_part_bytes(C.byref(d), C.byref(out)) != 0


def test_save_load_round_trip_of_mesh_and_results(tmp_path):
    """tests/SaveLoadTests.cpp: fields set from analytic functions of the node positions are saved next to the mesh, then
    loaded into a mesh read back from the file (here: by another rank count's worth of readers, each through its own
    node ids) and compared with the functions."""
    from l3ster_amd import native_io
    order, ne, world = 3, (4, 2, 2), 2
    saved = [system.CubePartition(ne, order, parts=(2, 1, 1), rank=r, perturb=0.07) for r in range(world)]
    f = lambda c: np.stack([np.sin(c[:, 0]) + c[:, 1] * c[:, 2], c[:, 0] - 2.0 * c[:, 2]])
    mesh_path, res_path = tmp_path / "m.mesh", tmp_path / "m.res"
    fps = [mesh_file.part_of(p) for p in saved]
    sizes = [fp.n_bytes() for fp in fps]
    for r in (1, 0):
        fps[r].save(mesh_path, sizes, r, "saved mesh")
        vals = f(saved[r].node_coords()[:saved[r].n_owned_nodes])
        native_io.save(res_path, vals, saved[r].n_global_nodes, saved[r].global_node_base, "saved results", write_header=(r == 0))
    for r in range(world):
        part = mesh_file.FilePartition(mesh_path, r, order)
        coords = part.node_coords()  # from the vertices stored in the file
        gids = part.node_grid_id  # owned and ghost nodes by the file's global ids
        for k in range(2):
            assert np.allclose(native_io.load(res_path, k, node_ids=gids), f(coords)[k], atol=1e-14)


def _save_worker(rank, world, port, path, order):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        part = system.CubePartition((4, 2, 2), order, parts=(world, 1, 1), rank=rank, perturb=0.1)
        sizes = mesh_file.save_partition(path, part, comment="saved by two ranks")  # the sizes travel through all_gather_object
        assert len(sizes) == world
        dist.barrier()
        again = mesh_file.FilePartition(path, rank, order)
        assert again.n_elems == part.n_elems and again.n_owned_nodes == part.n_owned_nodes
        assert np.array_equal(again.ghost_global_id, part.ghost_global_id)
    finally:
        dist.destroy_process_group()


def test_ranks_save_their_parts_through_torch_distributed(tmp_path):
    """save(comm, mesh, path, comment) of the reference: every rank serialises its part, the sizes are gathered, rank 0 writes
    the header (post/NativeIO.hpp:75-108) -- here with the gloo backend, two processes."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    path = str(tmp_path / "two.mesh")
    mp.spawn(_save_worker, args=(2, port, path, 2), nprocs=2, join=True)
    assert len(mesh_file.info(path)) == 2
    comment, parts = ONP.mesh_file_parse(open(path, "rb").read(), 2)
    assert comment == "saved by two ranks" and len(parts) == 2
    whole = system.CubePartition((4, 2, 2), 2, perturb=0.1)
    assert sum(p[2] for p in parts) == whole.n_global_nodes  # owned node counts add up to the mesh
