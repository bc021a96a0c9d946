"""Native results files of the reference (post/NativeIO.hpp): save(comm, mesh, solution_manager, path, inds, comment) and
Loader::loadResults, through the C ABI (l3k_results_save / _info / _load, host code in csrc/host/native_io.cpp).

The file holds field-major arrays indexed by global node id; a rank passes the rows of its owned nodes.  Device vectors
are node-interleaved ([node][dof]): save_solution() de-interleaves the owned rows with torch before handing them over.
"""
import ctypes as C

import numpy as np

from . import capi


def save(path, fields, n_global_nodes, node_begin=0, comment="", write_header=True):
    """fields: host array [n_fields][n_local_nodes] (values of this rank's owned nodes, ascending global id)."""
    f = np.ascontiguousarray(fields, dtype=np.float64)
    if f.ndim != 2:
        raise ValueError("fields must be [n_fields][n_local_nodes]")
    lib = capi.load()
    capi.check(lib.l3k_results_save(str(path).encode(), comment.encode(), f.shape[0], int(n_global_nodes), int(node_begin),
                                    f.shape[1], f.ctypes.data_as(capi.c_double_p), max(1, f.shape[1]), int(bool(write_header))))


def info(path):
    lib = capi.load()
    nf, nn = C.c_size_t(), C.c_size_t()
    capi.check(lib.l3k_results_info(str(path).encode(), C.byref(nf), C.byref(nn)))
    return nf.value, nn.value


def load(path, field, node_ids=None, node_begin=0, n=None):
    """Values of one field at the given global node ids (Loader::loadResultsImpl), or of the contiguous range
    [node_begin, node_begin + n)."""
    lib = capi.load()
    if node_ids is not None:
        ids = np.ascontiguousarray(node_ids, dtype=np.int64)
        out = np.empty(ids.size, dtype=np.float64)
        capi.check(lib.l3k_results_load(str(path).encode(), int(field), ids.size, ids.ctypes.data_as(capi.c_int64_p), 0,
                                        out.ctypes.data_as(capi.c_double_p)))
        return out
    if n is None:
        n = info(path)[1] - node_begin
    out = np.empty(int(n), dtype=np.float64)
    capi.check(lib.l3k_results_load(str(path).encode(), int(field), int(n), None, int(node_begin),
                                    out.ctypes.data_as(capi.c_double_p)))
    return out


def save_solution(path, x, part, dofs_per_node, dof_inds=None, comment="", write_header=None):
    """Saves the owned rows of a node-interleaved solution vector (torch tensor or numpy array of n_owned_nodes *
    dofs_per_node entries) of a CubePartition rank: one file field per entry of dof_inds (default: all dofs)."""
    n_owned = int(part.n_owned_nodes)
    if hasattr(x, "detach"):
        rows = x.detach()[:n_owned * dofs_per_node].reshape(n_owned, dofs_per_node).t().contiguous().cpu().numpy()
    else:
        rows = np.asarray(x)[:n_owned * dofs_per_node].reshape(n_owned, dofs_per_node).T
    if dof_inds is not None:
        rows = rows[list(dof_inds)]
    save(path, rows, part.n_global_nodes, part.global_node_base, comment,
         write_header=(part.rank == 0) if write_header is None else write_header)
