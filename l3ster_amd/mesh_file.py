"""Native mesh files of the reference (post/NativeIO.hpp:75-108 save(comm, mesh, path, comment); :185-232
loadUnifiedMesh / loadPartitionedMesh), through the C ABI (l3k_meshfile_*, host code in csrc/host/mesh_file.cpp).

A file holds one serialised MeshPartition<order> per rank: per domain the hex / quad / line elements with GLOBAL node
ids, vertices and element ids, then the rank's owned node range and the boundary domain ids.  part_of() turns one rank's
partition (system.CubePartition, partition.PartitionedMesh) into that description, with the boundary quads of
makeCubeMesh (mesh/primitives/CubeMesh.hpp:66-138) for the cube; FilePartition turns a saved part back into what
DeviceMesh / HaloPlan / DistributedOperator take, keeping the file's global numbering (which the native results file is
indexed by).
"""
import ctypes as C

import numpy as np

from . import capi

TYPES = ("hex", "quad", "line")
# local node indices of the six hex sides (mesh/ElementTraits.hpp:84-95: 0 z=-1, 1 z=+1, 2 y=-1, 3 y=+1, 4 x=-1, 5 x=+1),
# the two remaining axes in ascending order, lower axis fastest: the orientation makeCubeMesh gives its boundary quads
# (CubeMesh.hpp:72-81, 97-110, 120-133)


def side_nodes(order, side):
    n = order + 1
    a = np.arange(n)
    fixed = 0 if side % 2 == 0 else order
    if side < 2:
        return (a[None, :] + n * a[:, None] + n * n * fixed).reshape(-1)
    if side < 4:
        return (a[None, :] + n * fixed + n * n * a[:, None]).reshape(-1)
    return (fixed + n * a[None, :] + n * n * a[:, None]).reshape(-1)


class MeshFilePart:
    """One rank's share of a mesh file: domains[id][type] = (nodes uint64 [n][nn], verts float64 [n][nv][3], ids uint64
    [n]) for the types present."""

    def __init__(self, order, domains, nodes_begin, n_owned_nodes, boundary_ids):
        self.order, self.nodes_begin, self.n_owned_nodes = int(order), int(nodes_begin), int(n_owned_nodes)
        self.boundary_ids = np.asarray(boundary_ids, dtype=np.uint16).reshape(-1)
        self.domains = {}
        for dom_id, by_type in domains.items():
            self.domains[int(dom_id)] = {
                t: (np.ascontiguousarray(v[0], dtype=np.uint64), np.ascontiguousarray(v[1], dtype=np.float64),
                    np.ascontiguousarray(v[2], dtype=np.uint64)) for t, v in by_type.items()}

    def _desc(self):
        doms = (capi.MeshFileDomain * max(1, len(self.domains)))()
        for i, (dom_id, by_type) in enumerate(self.domains.items()):
            doms[i].id = dom_id
            for t in TYPES:
                e = getattr(doms[i], t)
                if t in by_type:
                    nodes, verts, ids = by_type[t]
                    e.n = ids.size
                    e.nodes = nodes.ctypes.data_as(capi.c_uint64_p)
                    e.verts = verts.ctypes.data_as(capi.c_double_p)
                    e.ids = ids.ctypes.data_as(capi.c_uint64_p)
        d = capi.MeshFilePartDesc(self.order, len(self.domains), doms, self.nodes_begin, self.n_owned_nodes,
                                  self.boundary_ids.size, self.boundary_ids.ctypes.data_as(capi.c_uint16_p))
        d._keep = doms
        return d

    def n_bytes(self):
        out = C.c_size_t()
        capi.check(capi.load().l3k_meshfile_part_bytes(C.byref(self._desc()), C.byref(out)))
        return out.value

    def save(self, path, part_sizes, part, comment="", write_header=None):
        """part_sizes: n_bytes() of every part (the caller gathers them; the reference: comm.gather, NativeIO.hpp:83)."""
        sizes = (C.c_size_t * len(part_sizes))(*[int(s) for s in part_sizes])
        capi.check(capi.load().l3k_meshfile_save(str(path).encode(), comment.encode(), len(part_sizes), sizes, int(part),
                                                 C.byref(self._desc()),
                                                 int(part == 0 if write_header is None else bool(write_header))))


def info(path):
    """Sizes in bytes of the parts a file holds."""
    lib = capi.load()
    n = C.c_size_t()
    capi.check(lib.l3k_meshfile_info(str(path).encode(), C.byref(n), None, 0))
    sizes = (C.c_size_t * max(1, n.value))()
    capi.check(lib.l3k_meshfile_info(str(path).encode(), C.byref(n), sizes, n.value))
    return [sizes[i] for i in range(n.value)]


def load(path, part, order):
    """Part `part` of the file as a mesh of the given element order (loadPartitionedMesh, NativeIO.hpp:219-232)."""
    lib = capi.load()
    h = C.c_void_p()
    capi.check(lib.l3k_meshfile_load(str(path).encode(), int(part), int(order), C.byref(h)))
    try:
        d = capi.MeshFilePartDesc()
        capi.check(lib.l3k_meshfile_part_get(h, C.byref(d)))
        doms = {}
        for i in range(d.n_domains):
            by_type = {}
            for dim, t in zip((3, 2, 1), TYPES):
                e = getattr(d.domains[i], t)
                if e.n:
                    nn, nv = (order + 1) ** dim, 2 ** dim
                    by_type[t] = (np.ctypeslib.as_array(e.nodes, shape=(e.n, nn)).copy(),
                                  np.ctypeslib.as_array(e.verts, shape=(e.n, nv, 3)).copy(),
                                  np.ctypeslib.as_array(e.ids, shape=(e.n,)).copy())
            doms[d.domains[i].id] = by_type
        bnd = np.ctypeslib.as_array(d.boundary_ids, shape=(d.n_boundary_ids,)).copy() if d.n_boundary_ids else \
            np.zeros(0, np.uint16)
        return MeshFilePart(order, doms, d.nodes_begin, d.n_owned_nodes, bnd)
    finally:
        lib.l3k_meshfile_part_destroy(h)


def load_unified(path, order):
    """All parts merged into one (loadUnifiedMesh, NativeIO.hpp:185-216): the domains' elements in part order, the
    boundary ids of the first part that has any; the merged mesh owns every node."""
    parts = [load(path, i, order) for i in range(len(info(path)))]
    doms, bnd = {}, np.zeros(0, np.uint16)
    for p in parts:
        for dom_id, by_type in p.domains.items():
            for t, v in by_type.items():
                doms.setdefault(dom_id, {}).setdefault(t, []).append(v)
        if bnd.size == 0:
            bnd = p.boundary_ids
    merged = {d: {t: tuple(np.concatenate([v[k] for v in vs]) for k in range(3)) for t, vs in by_type.items()}
              for d, by_type in sorted(doms.items())}
    n_nodes = max((int(v[0].max()) + 1 for by_type in merged.values() for v in by_type.values() if v[0].size), default=0)
    return MeshFilePart(order, merged, 0, n_nodes, bnd)


def local_to_global(part):
    """Global node id of every local node of a partition ([owned | ghosts])."""
    return np.concatenate([part.global_node_base + np.arange(part.n_owned_nodes, dtype=np.int64),
                           np.asarray(part.ghost_global_id, dtype=np.int64)]).astype(np.uint64)


def cube_element_ids(part):
    """Element ids of a CubePartition's hexes and boundary quads exactly as makeCubeMesh numbers them (CubeMesh.hpp:39-138:
    hexes x fastest, then the z faces back/front interleaved, the y faces bottom/top, the x faces left/right).  Returns
    (hex_ids [n_elems], quad_id(elem_index_array, side) -> ids)."""
    p, (ex, ey, ez) = part.order, part.ne
    NX, NY = ex * p + 1, ey * p + 1
    g = part.node_grid_id[part.elem_nodes[:, 0]]  # the element's lowest corner on the global node grid
    ix, iy, iz = (g % NX) // p, ((g // NX) % NY) // p, (g // (NX * NY)) // p
    hex_ids = (ix + ex * (iy + ey * iz)).astype(np.uint64)
    n_hex = ex * ey * ez
    base = (n_hex, n_hex + 2 * ex * ey, n_hex + 2 * ex * ey + 2 * ex * ez)

    def quad_ids(elems, side):
        a, b, c = ix[elems], iy[elems], iz[elems]
        k = (b * ex + a, c * ex + a, c * ey + b)[side // 2]
        return (base[side // 2] + 2 * k + side % 2).astype(np.uint64)

    return hex_ids, quad_ids


def part_of(part, boundary_sides=None, domain_id=0, hex_ids=None, quad_ids=None, side_domain=lambda s: s + 1):
    """The mesh-file description of one rank's partition.  boundary_sides: (elem index [n], side [n]) of the element
    sides that carry a boundary quad (default for a CubePartition: all six cube sides, domains 1..6 = back, front,
    bottom, top, left, right as CubeMeshIds, CubeMesh.hpp:8-11); side_domain maps a side tag to the boundary domain id.
    For a partition that is not a cube give hex_ids (global element ids) and quad_ids(elems, side)."""
    p = part.order
    gid = local_to_global(part)
    en = gid[np.asarray(part.elem_nodes, dtype=np.int64)]
    is_cube = hasattr(part, "ne") and hasattr(part, "elem_boundary")
    if hex_ids is None:
        if is_cube:
            hex_ids, quad_ids = cube_element_ids(part)
        else:
            hex_ids = np.asarray(part.elem_global, dtype=np.uint64)
    if boundary_sides is None:
        boundary_sides = part.boundary_sides() if is_cube else (np.zeros(0, np.int64), np.zeros(0, np.uint8))
    fe, fs = (np.asarray(x) for x in boundary_sides)
    domains = {domain_id: {"hex": (en, part.elem_verts, hex_ids)}} if part.n_elems else {}
    corner = side_nodes(1, 0), side_nodes(1, 1), side_nodes(1, 2), side_nodes(1, 3), side_nodes(1, 4), side_nodes(1, 5)
    bnd_ids = sorted({int(side_domain(int(s))) for s in (range(6) if is_cube else np.unique(fs))})
    for s in np.unique(fs):
        e = fe[fs == s]
        if quad_ids is None:
            raise ValueError("quad_ids(elems, side) is needed for the ids of the boundary elements")
        domains[int(side_domain(int(s)))] = {"quad": (en[e][:, side_nodes(p, int(s))],
                                                      np.asarray(part.elem_verts)[e][:, corner[int(s)]],
                                                      quad_ids(e, int(s)))}
    return MeshFilePart(p, dict(sorted(domains.items())), part.global_node_base if part.n_owned_nodes else 0,
                        part.n_owned_nodes, bnd_ids)


def save_partition(path, part, world=None, comment="", sizes=None, **kw):
    """Saves this rank's partition.  The sizes of all parts come from `sizes`, or are all-gathered over torch.distributed
    when a process group is up, or are just this part's for a single rank."""
    fp = part_of(part, **kw)
    mine = fp.n_bytes()
    if sizes is None:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            sizes = [None] * dist.get_world_size()
            dist.all_gather_object(sizes, mine)
        else:
            sizes = [mine]
    fp.save(path, sizes, part.rank, comment)
    return sizes


class FilePartition:
    """One rank's part of a saved mesh as DeviceMesh / HaloPlan / DistributedOperator take it: local numbering [owned by
    global id | ghosts by global id], interior elements first, neighbour lists from the other parts of the same file
    (every rank reads the node tables of the other parts: no communication).  Dirichlet / boundary queries go by
    boundary domain id."""
    dim = 3

    def __init__(self, path, rank, order, domain_id=None):
        sizes = info(path)
        world = len(sizes)
        parts = [load(path, r, order) for r in range(world)]
        me = parts[rank]
        self.order, self.rank, self.parts, self.file_part = order, rank, (world, 1, 1), me
        vol = [d for d, by_type in me.domains.items() if "hex" in by_type] if domain_id is None else [domain_id]
        N = (order + 1) ** 3
        nodes = np.concatenate([me.domains[d]["hex"][0] for d in vol]) if vol else np.zeros((0, N), np.uint64)
        verts = np.concatenate([me.domains[d]["hex"][1] for d in vol]) if vol else np.zeros((0, 8, 3))
        ids = np.concatenate([me.domains[d]["hex"][2] for d in vol]) if vol else np.zeros(0, np.uint64)
        b0, b1 = me.nodes_begin, me.nodes_begin + me.n_owned_nodes
        begins = np.array([q.nodes_begin for q in parts], dtype=np.int64)
        counts = np.array([q.n_owned_nodes for q in parts], dtype=np.int64)
        self.global_node_base, self.n_owned_nodes = b0, me.n_owned_nodes
        self.n_global_nodes = int(counts.sum())
        E = nodes.astype(np.int64)
        is_owned = (E >= b0) & (E < b1)
        ghosts = np.unique(E[~is_owned])
        self.ghost_global_id, self.n_ghost_nodes = ghosts, int(ghosts.size)
        local = np.where(is_owned, E - b0, self.n_owned_nodes + np.searchsorted(ghosts, E))
        interior = is_owned.all(axis=1)
        perm = np.concatenate([np.nonzero(interior)[0], np.nonzero(~interior)[0]])
        self.elem_nodes = np.ascontiguousarray(local[perm].astype(np.uint32))
        self.elem_verts = np.ascontiguousarray(verts[perm])
        self.elem_global = ids[perm].astype(np.int64)
        self.n_elems, self.n_interior_elems = int(E.shape[0]), int(interior.sum())
        self.node_grid_id = np.concatenate([b0 + np.arange(self.n_owned_nodes, dtype=np.int64), ghosts])
        # owner of a ghost: the part whose range holds it (util/SegmentedOwnership.hpp:11-45)
        owners_sorted = np.argsort(begins, kind="stable")
        nonempty = owners_sorted[counts[owners_sorted] > 0]
        ghost_owner = nonempty[np.searchsorted(begins[nonempty], ghosts, side="right") - 1] if ghosts.size else \
            np.zeros(0, np.int64)
        shared = {}
        for q in range(world):
            if q == rank:
                continue
            touched = [by_type["hex"][0].reshape(-1).astype(np.int64) for by_type in parts[q].domains.values()
                       if "hex" in by_type]
            t = np.unique(np.concatenate(touched)) if touched else np.zeros(0, np.int64)
            t = t[(t >= b0) & (t < b1)]
            if t.size:
                shared[q] = (t - b0).astype(np.int32)
        nbrs = sorted(set(ghost_owner.tolist()) | set(shared))
        self.nbr_rank, self.send_nodes, self.ghost_ranges = [], [], []
        if np.any(np.diff(ghost_owner) < 0):  # ghosts are sorted by global id: the owners' ranges must ascend with the rank
            raise ValueError("mesh file: the parts' node ranges are not in rank order")
        cursor = 0
        for q in nbrs:
            self.nbr_rank.append(int(q))
            self.send_nodes.append(shared.get(q, np.zeros(0, np.int32)))
            n_from_q = int(np.count_nonzero(ghost_owner == q))
            self.ghost_ranges.append((cursor, cursor + n_from_q))  # the ranges tile the ghosts in neighbour order
            cursor += n_from_q

    @property
    def n_local_nodes(self):
        return self.n_owned_nodes + self.n_ghost_nodes

    def _local(self, gids):
        g = np.asarray(gids, dtype=np.int64)
        b0 = self.global_node_base
        owned = (g >= b0) & (g < b0 + self.n_owned_nodes)
        pos = np.searchsorted(self.ghost_global_id, g)
        pos_c = np.minimum(pos, max(0, self.n_ghost_nodes - 1))
        is_ghost = ~owned & (self.n_ghost_nodes > 0) & (self.ghost_global_id[pos_c] == g if self.n_ghost_nodes else False)
        return np.where(owned, g - b0, np.where(is_ghost, self.n_owned_nodes + pos, -1))

    def boundary_nodes(self, domain_ids):
        """Local ids of the nodes of the boundary elements in the listed domains (what BCDefinition::defineDirichlet
        marks)."""
        out = []
        for d in domain_ids:
            for t in ("quad", "line"):
                if d in self.file_part.domains and t in self.file_part.domains[d]:
                    l = self._local(self.file_part.domains[d][t][0].reshape(-1))
                    out.append(l[l >= 0])
        return np.unique(np.concatenate(out)) if out else np.zeros(0, np.int64)

    def dirichlet_mask(self, dofs_per_node, unknowns=(0,), domain_ids=None):
        ids = self.file_part.boundary_ids if domain_ids is None else domain_ids
        mask = np.zeros((self.n_local_nodes, dofs_per_node), dtype=np.uint8)
        on = self.boundary_nodes([int(d) for d in ids])
        for u in unknowns:
            mask[on, u] = 1
        return mask.reshape(-1)

    def boundary_sides(self, domain_ids):
        """(element index, side) of the element sides the boundary quads of the listed domains lie on: the quad's four
        corner nodes matched against the corners of every side (what MeshPartition's boundary views hold)."""
        p, n = self.order, self.order + 1
        cpos = np.array([0, p, p * n, p * n + p])  # corners of a quad's node array
        gid = self.node_grid_id
        key = {}
        for s in range(6):
            c = np.sort(gid[self.elem_nodes[:, side_nodes(p, s)[cpos]].astype(np.int64)], axis=1)
            for e, k in enumerate(map(tuple, c)):
                key[k] = (e, s)
        fe, fs = [], []
        for d in domain_ids:
            q = self.file_part.domains.get(int(d), {}).get("quad")
            if q is None:
                continue
            for k in map(tuple, np.sort(q[0][:, cpos].astype(np.int64), axis=1)):
                e, s = key[k]
                fe.append(e)
                fs.append(s)
        return np.asarray(fe, dtype=np.int64), np.asarray(fs, dtype=np.uint8)

    def node_coords(self):
        from . import system
        return system.CubePartition.node_coords(self)

    def synthetic_vector(self, dofs_per_node, seed=42, ncols=1):
        from . import system
        return system.CubePartition.synthetic_vector(self, dofs_per_node, seed, ncols)
