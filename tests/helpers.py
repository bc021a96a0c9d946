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
