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
DistributedOperator use.  The index work is torch array code that runs on the GPU.
"""
import numpy as np
import torch

from . import system


class PartitionedMesh:
    dim = 3

    def __init__(self, elem_nodes, elem_verts, n_noninternal, elem_part, rank, world, order, device=None):
        """elem_nodes: [n_elems][(order+1)^3] global node ids numbered [non-internal | internal, contiguous per element]
        (what system.elevate_order returns); elem_verts [n_elems][8][3]; elem_part [n_elems] in [0, world).  The index
        work (sorts, uniques, searches over the global node table) runs in torch on `device` (default: the GPU if there
        is one); the attributes are numpy arrays like CubePartition's."""
        dev = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        en = torch.as_tensor(np.ascontiguousarray(elem_nodes).astype(np.int64), device=dev)
        part = torch.as_tensor(np.ascontiguousarray(elem_part).astype(np.int64), device=dev)
        n_elems, N = en.shape
        n_nodes = int(en.max()) + 1 if n_elems else 0
        self.order, self.rank, self.parts = order, rank, (world, 1, 1)
        # ownership: lowest part touching the node (SegmentedOwnership / the METIS-based distribution's rule)
        owner = torch.full((n_nodes,), world, dtype=torch.int64, device=dev)
        for q in range(world - 1, -1, -1):  # descending: the lowest part writes last
            owner[en[part == q].reshape(-1)] = q
        # new global ids: rank-major, inside a rank ascending old id (non-internal ids are all below the internal ones)
        order_idx = torch.sort(owner, stable=True).indices
        new_gid = torch.empty(n_nodes, dtype=torch.int64, device=dev)
        new_gid[order_idx] = torch.arange(n_nodes, device=dev)
        counts = torch.bincount(owner, minlength=world + 1)[:world]
        base = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(counts, 0)])
        b0, b1 = int(base[rank]), int(base[rank + 1])
        self.global_node_base, self.n_global_nodes, self.n_owned_nodes = b0, int(n_nodes), b1 - b0
        mine = part == rank
        E = new_gid[en[mine]]  # my elements in new global ids
        touched = torch.unique(E)
        ghosts = touched[(touched < b0) | (touched >= b1)]  # sorted by global id
        self.n_ghost_nodes = int(ghosts.numel())
        self.ghost_global_id = ghosts.cpu().numpy()
        # local ids: owned = gid - base, ghosts behind them in global-id order
        is_owned = (E >= b0) & (E < b1)
        local = torch.where(is_owned, E - b0, self.n_owned_nodes + torch.searchsorted(ghosts, E))
        interior = is_owned.all(dim=1)
        perm = torch.cat([torch.nonzero(interior).reshape(-1), torch.nonzero(~interior).reshape(-1)])  # interior first, stable
        self.elem_nodes = np.ascontiguousarray(local[perm].cpu().numpy().astype(np.uint32))
        mine_idx = torch.nonzero(mine).reshape(-1)[perm].cpu().numpy()
        self.elem_verts = np.ascontiguousarray(np.asarray(elem_verts, dtype=np.float64)[mine_idx])
        self.elem_global = mine_idx  # index of each local element in the global mesh
        self.n_elems, self.n_interior_elems = int(mine.sum()), int(interior.sum())
        # partition-independent node id of every local node (the OLD global id): for synthetic data and comparisons
        self.node_grid_id = torch.cat([order_idx[b0:b1], order_idx[ghosts]]).cpu().numpy()
        # neighbours.  Import receive / export send: my ghosts, grouped by owner (contiguous: global ids are rank-major)
        ghost_owner = (torch.searchsorted(base, ghosts, right=True) - 1).cpu().numpy()
        # import send / export receive: my owned nodes that elements of other parts touch, per part, ascending local id
        other = ~mine
        en_o = new_gid[en[other]]
        sel = (en_o >= b0) & (en_o < b1)
        q_of = part[other][:, None].expand(-1, N)[sel]
        pairs = torch.unique(torch.stack([q_of, en_o[sel] - b0], dim=1), dim=0).cpu().numpy() if bool(sel.any()) else np.zeros((0, 2), np.int64)
        nbrs = sorted(set(ghost_owner.tolist()) | set(pairs[:, 0].tolist()))
        self.nbr_rank, self.send_nodes, self.ghost_ranges = [], [], []
        cursor = 0
        for q in nbrs:
            self.nbr_rank.append(int(q))
            self.send_nodes.append(pairs[pairs[:, 0] == q, 1].astype(np.int32))
            n_from_q = int(np.count_nonzero(ghost_owner == q))
            self.ghost_ranges.append((cursor, cursor + n_from_q))  # global ids are rank-major: the ranges tile the ghosts
            cursor += n_from_q

    n_local_nodes = system.CubePartition.n_local_nodes
    node_coords = system.CubePartition.node_coords
    synthetic_vector = system.CubePartition.synthetic_vector


def rcb_partition(elem_verts, n_parts):
    """Recursive coordinate bisection of the element centroids into n_parts parts of (nearly) equal element counts: the
    stand-in for mesh::partitionMesh (mesh/PartitionMesh.hpp:142-183, METIS_PartMeshNodal) when no partition vector comes
    from outside.  Splits along the longest extent of the current set, proportionally to the parts on either side, so any
    n_parts works.  Returns int64 [n_elems]."""
    c = np.asarray(elem_verts, dtype=np.float64).mean(axis=1)
    out = np.zeros(c.shape[0], dtype=np.int64)

    def split(idx, first, n):
        if n == 1:
            out[idx] = first
            return
        pts = c[idx]
        axis = int(np.argmax(pts.max(axis=0) - pts.min(axis=0)))
        n_left = n // 2
        k = int(round(idx.size * n_left / n))
        order = np.argsort(pts[:, axis], kind="stable")
        split(idx[order[:k]], first, n_left)
        split(idx[order[k:]], first + n_left, n - n_left)

    split(np.arange(c.shape[0]), 0, int(n_parts))
    return out
