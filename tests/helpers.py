"""Shared helpers of the test-suite (CPU + GPU)."""
import numpy as np

import oracle_lib as O

HEX = np.array([[1, 1, 0], [2, 1, 0], [1, 3, 0], [3, 4, 0], [1, 1, 1], [2, 1, 1.5], [1, 3, 2], [3, 4, 3.5]], float)


class SingleElementMesh:
    """A one-element 'partition' with identity numbering (duck-types l3ster_amd.system.CubePartition)."""

    def __init__(self, order, verts):
        N = (order + 1) ** 3
        self.dim, self.order = 3, order
        self.n_elems = self.n_interior_elems = 1
        self.n_owned_nodes, self.n_ghost_nodes = N, 0
        self.elem_nodes = np.arange(N, dtype=np.uint32).reshape(1, N)
        self.elem_verts = np.asarray(verts, dtype=np.float64).reshape(1, 8, 3)

    @property
    def n_local_nodes(self):
        return self.n_owned_nodes


def oracle_mesh(part, nq, dofs_per_node, field_inds, dirichlet=None, fields=None):
    return O.MeshView(3, part.order, nq, part.elem_nodes, part.elem_verts, part.n_local_nodes, dofs_per_node, field_inds,
                      dirichlet, fields)


def rel_err(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


class PeriodicXPartition:
    """A one-rank cube partition made periodic in x through the ghost machinery: the nodes of the face x = 1 become ghost
    nodes owned by ... the same rank (their partners on the face x = 0), so import / export run as a self exchange --
    the only neighbour exchange one GPU can carry through RCCL.  Duck-types l3ster_amd.system.CubePartition.
    `merged` is the connectivity with the identification carried out (for the oracle on the periodic mesh)."""

    def __init__(self, part):
        ne, p = part.ne, part.order
        nx = ne[0] * p + 1
        gid = part.node_grid_id.astype(np.int64)
        is_ghost = (gid % nx) == nx - 1
        old_owned = np.nonzero(~is_ghost)[0]
        old_ghost = np.nonzero(is_ghost)[0]
        new_of_old = np.empty(part.n_local_nodes, np.int64)
        new_of_old[old_owned] = np.arange(len(old_owned))
        new_of_old[old_ghost] = len(old_owned) + np.arange(len(old_ghost))
        row_of_gid = {int(g): i for i, g in enumerate(gid)}
        partner_old = np.array([row_of_gid[int(g) - (nx - 1)] for g in gid[old_ghost]])
        en = new_of_old[part.elem_nodes.astype(np.int64)]
        touches = (en >= len(old_owned)).any(axis=1)
        order = np.concatenate([np.nonzero(~touches)[0], np.nonzero(touches)[0]])  # interior elements first
        self.dim, self.order, self.ne = 3, p, ne
        self.n_elems, self.n_interior_elems = part.n_elems, int((~touches).sum())
        self.n_owned_nodes, self.n_ghost_nodes = len(old_owned), len(old_ghost)
        self.elem_nodes = en[order].astype(np.uint32)
        self.elem_verts = part.elem_verts[order]
        perm = np.concatenate([old_owned, old_ghost])
        self.node_grid_id = gid[perm]
        self.node_boundary = part.node_boundary[perm]
        self.elem_boundary = part.elem_boundary[order]
        self.nbr_rank = [0]
        self.send_nodes = [new_of_old[partner_old].astype(np.int32)]
        self.ghost_ranges = [(0, len(old_ghost))]
        merged_of_new = np.arange(self.n_owned_nodes + self.n_ghost_nodes)
        merged_of_new[self.n_owned_nodes:] = self.send_nodes[0]
        self.merged = merged_of_new[self.elem_nodes.astype(np.int64)].astype(np.uint32)

    @property
    def n_local_nodes(self):
        return self.n_owned_nodes + self.n_ghost_nodes
