"""One rank's share of an order-p hex mesh given an element partition vector: ownership, global and local numbering, ghost
lists, neighbour exchange lists, interior / border split -- the work of comm::distributeMesh + mesh::LocalMeshView in the
reference (comm/DistributeMesh.hpp, mesh/LocalMeshView.hpp:425-458, util/SegmentedOwnership.hpp:11-45,
comm/ImportExport.hpp:29-72) for a mesh every rank holds in full (the reference reads the mesh on rank 0, partitions it with
METIS and sends every rank its part; here every rank elevates the global order-1 mesh on its own GPU -- system.elevate_order,
tens of milliseconds -- and extracts its part without any communication, so there is no serial step and no mesh traffic).

Conventions (the ones the device path wants): a node is owned by the lowest part that touches it; every rank owns a
contiguous global range, ranks in ascending order; inside a rank the non-internal nodes come first, then the
element-internal nodes contiguous per element; local numbering [owned | ghosts sorted by global id]; elements whose nodes
are all owned ("interior") first.  The result has the attributes of system.CubePartition that DeviceMesh, HaloPlan and
DistributedOperator use.  Host code (numpy); sized for meshes up to ~10^8 nodes.
"""
import numpy as np

from . import system


class PartitionedMesh:
    dim = 3

    def __init__(self, elem_nodes, elem_verts, n_noninternal, elem_part, rank, world, order):
        """elem_nodes: [n_elems][(order+1)^3] global node ids numbered [non-internal | internal, contiguous per element]
        (what system.elevate_order returns); elem_verts [n_elems][8][3]; elem_part [n_elems] in [0, world)."""
        en = np.ascontiguousarray(elem_nodes, dtype=np.int64)
        part = np.ascontiguousarray(elem_part, dtype=np.int64)
        n_elems, N = en.shape
        n_nodes = int(en.max()) + 1 if n_elems else 0
        self.order, self.rank, self.parts = order, rank, (world, 1, 1)
        # ownership: lowest part touching the node (SegmentedOwnership / the METIS-based distribution's rule)
        owner = np.full(n_nodes, world, dtype=np.int64)
        for q in range(world - 1, -1, -1):  # descending: the lowest part writes last
            owner[en[part == q].reshape(-1)] = q
        # new global ids: rank-major, inside a rank ascending old id (non-internal ids are all below the internal ones)
        order_idx = np.argsort(owner, kind="stable")
        new_gid = np.empty(n_nodes, dtype=np.int64)
        new_gid[order_idx] = np.arange(n_nodes)
        counts = np.bincount(owner, minlength=world + 1)[:world]
        base = np.concatenate([[0], np.cumsum(counts)])
        self.global_node_base, self.n_global_nodes = int(base[rank]), int(n_nodes)
        self.n_owned_nodes = int(counts[rank])
        mine = part == rank
        E = new_gid[en[mine]]  # my elements in new global ids
        touched = np.unique(E)
        ghosts = touched[(touched < base[rank]) | (touched >= base[rank + 1])]  # sorted by global id
        self.n_ghost_nodes = int(ghosts.size)
        # local ids: owned = gid - base, ghosts behind them in global-id order
        loc = np.searchsorted(ghosts, E)
        is_owned = (E >= base[rank]) & (E < base[rank + 1])
        local = np.where(is_owned, E - base[rank], self.n_owned_nodes + loc)
        interior = is_owned.all(axis=1)
        perm = np.concatenate([np.nonzero(interior)[0], np.nonzero(~interior)[0]])  # interior elements first, stable
        self.elem_nodes = np.ascontiguousarray(local[perm].astype(np.uint32))
        self.elem_verts = np.ascontiguousarray(np.asarray(elem_verts, dtype=np.float64)[mine][perm])
        self.elem_global = np.nonzero(mine)[0][perm]  # index of each local element in the global mesh
        self.n_elems, self.n_interior_elems = int(mine.sum()), int(interior.sum())
        # partition-independent node id of every local node (the OLD global id): for synthetic data and comparisons
        old_of_new = order_idx
        self.node_grid_id = np.concatenate([old_of_new[base[rank]:base[rank + 1]], old_of_new[ghosts]])
        # neighbours.  Import receive / export send: my ghosts, grouped by owner (contiguous: global ids are rank-major)
        ghost_owner = np.searchsorted(base, ghosts, side="right") - 1
        # import send / export receive: my owned nodes that elements of other parts touch, per part, ascending local id
        other = ~mine
        en_o = new_gid[en[other]]
        sel = (en_o >= base[rank]) & (en_o < base[rank + 1])
        q_of = np.repeat(part[other], N).reshape(en_o.shape)[sel]
        pairs = np.unique(np.stack([q_of, en_o[sel] - base[rank]], axis=1), axis=0) if sel.any() else np.zeros((0, 2), np.int64)
        nbrs = sorted(set(ghost_owner.tolist()) | set(pairs[:, 0].tolist()))
        self.nbr_rank, self.send_nodes, self.ghost_ranges = [], [], []
        for q in nbrs:
            self.nbr_rank.append(int(q))
            self.send_nodes.append(pairs[pairs[:, 0] == q, 1].astype(np.int32))
            g = np.nonzero(ghost_owner == q)[0]
            self.ghost_ranges.append((int(g[0]), int(g[-1]) + 1) if g.size else (0, 0))

    n_local_nodes = system.CubePartition.n_local_nodes
    node_coords = system.CubePartition.node_coords
    synthetic_vector = system.CubePartition.synthetic_vector
