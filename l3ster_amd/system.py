"""Host-side mirror of the reference's interface for the hot path, on top of the C ABI (include/l3k.h).

Mirrors (names, argument meaning, error behaviour) for this path only:
  * tables            -- math::getLobattoRuleAbsc, quad::getReferenceQuadrature, SumFactorization tables
  * CubePartition     -- generateAndDistributeMesh for structured cubes (comm/DistributeMesh.hpp:284-299)
  * MatrixFreeSystem  -- algsys::MatrixFreeSystem: assembleProblem -> Operator.apply(X, Y, alpha, beta)
                         (algsys/MatrixFreeSystem.hpp:24-89,1020-1140)
Vectors are torch tensors of shape (ncols, n_owned_dofs), i.e. column-major [row][col] multivectors with owned rows
only -- the host-view layout of the Tpetra multivectors of the reference.  torch is plumbing here (device memory,
streams); all compute goes through libl3k.so.
"""
import contextlib
import ctypes as C

import numpy as np

from . import capi
from .capi import L3KError, check

KERNEL_DIFFUSION3D = 0
KERNEL_DIFFUSION3D_VAR = 1
KERNEL_ADVDIFF3D = 4
KERNEL_MASS3D = 8  # A0 = I: known answers for w * detJ
KERNEL_DIFFUSION3D_POINT = 10  # operators and rhs read point.space.{x,y,z} and point.time
KERNEL_ADVECTION3D = 11  # scalar BDF3 advection, U = E = 1, F = 3, velocity from the point
KERNEL_DIVCURL3D = 12  # div-curl system, U = 3, E = 4
KERNEL_ADIABATIC3D = 6  # boundary equation kernels
KERNEL_ROBIN3D = 7
KERNEL_NORMALFLUX3D = 9  # boundary kernel with derivative operators (A1..A3)
KERNEL_ROBINPOINT3D = 14  # boundary kernel whose coefficients read point.space and point.time
RESIDUAL_DIFFUSION3D_ERROR = 0
RESIDUAL_LINEAR3D_ERROR = 2
RESIDUAL_UNIT3D = 4
RESIDUAL_COORDX3D = 6


# ------------------------------------------------------------------------------------------------------- tables
def gll_nodes(n):
    x = np.zeros(n)
    check(capi.load().l3k_gll_nodes(n, x.ctypes.data_as(capi.c_double_p)))
    return x


def gl_rule(nq):
    x, w = np.zeros(nq), np.zeros(nq)
    check(capi.load().l3k_gl_rule(nq, x.ctypes.data_as(capi.c_double_p), w.ctypes.data_as(capi.c_double_p)))
    return x, w


def n_qps1d(p, value_order=1, derivative_order=0):
    return capi.load().l3k_n_qps1d(p, value_order, derivative_order)


def basis_1d(p, nq):
    I, D = np.zeros((p + 1, nq)), np.zeros((p + 1, nq))
    check(capi.load().l3k_basis_1d(p, nq, I.ctypes.data_as(capi.c_double_p), D.ctypes.data_as(capi.c_double_p)))
    return I, D


def colloc_deriv(nq):
    Cm = np.zeros((nq, nq))
    check(capi.load().l3k_colloc_deriv(nq, Cm.ctypes.data_as(capi.c_double_p)))
    return Cm


def kernel_info(kernel_id):
    kp, name, nbytes = capi.KParams(), C.c_char_p(), C.c_size_t()
    check(capi.load().l3k_kernel_info(kernel_id, C.byref(kp), C.byref(name), C.byref(nbytes)))
    return dict(dimension=kp.dimension, n_equations=kp.n_equations, n_unknowns=kp.n_unknowns, n_fields=kp.n_fields,
                name=name.value.decode(), param_bytes=nbytes.value)


def residual_info(residual_id):
    kp, name, nbytes = capi.KParams(), C.c_char_p(), C.c_size_t()
    check(capi.load().l3k_residual_info(residual_id, C.byref(kp), C.byref(name), C.byref(nbytes)))
    return dict(dimension=kp.dimension, n_equations=kp.n_equations, n_fields=kp.n_fields, name=name.value.decode(),
                param_bytes=nbytes.value)


def instances():
    lib = capi.load()
    out = []
    for i in range(lib.l3k_instance_count()):
        v = [C.c_int() for _ in range(4)]
        check(lib.l3k_instance_info(i, *[C.byref(x) for x in v]))
        out.append(tuple(x.value for x in v))
    return out


# ------------------------------------------------------------------------------------------------------- mesh
class CubePartition:
    """One rank's block of a structured order-p hex mesh of [0,1]^3 (host arrays, reference numbering conventions)."""

    def __init__(self, ne, order, parts=(1, 1, 1), rank=0, perturb=0.0):
        lib = capi.load()
        ne = (ne,) * 3 if np.isscalar(ne) else tuple(ne)
        self.ne, self.order, self.parts, self.rank = ne, order, tuple(parts), rank
        h = C.c_void_p()
        check(lib.l3k_cube_partition_create((C.c_int * 3)(*ne), order, (C.c_int * 3)(*parts), rank, perturb,
                                            C.byref(h)))
        try:
            v = capi.HostMeshView()
            check(lib.l3k_hostmesh_view_get(h, C.byref(v)))
            N = (order + 1) ** 3
            as_np = lambda ptr, shape: np.ctypeslib.as_array(ptr, shape=shape).copy() if np.prod(shape) else \
                np.zeros(shape, dtype=np.ctypeslib.as_ctypes_type(ptr._type_))
            self.dim = 3
            self.n_elems, self.n_interior_elems = v.n_elems, v.n_interior_elems
            self.n_owned_nodes, self.n_ghost_nodes = v.n_owned_nodes, v.n_ghost_nodes
            self.global_node_base, self.n_global_nodes = v.global_node_base, v.n_global_nodes
            self.elem_nodes = as_np(v.elem_nodes, (v.n_elems, N))
            self.elem_verts = as_np(v.elem_verts, (v.n_elems, 8, 3))
            nl = v.n_owned_nodes + v.n_ghost_nodes
            self.node_grid_id = as_np(v.node_grid_id, (nl,))
            self.node_boundary = as_np(v.node_boundary, (nl,))
            self.elem_boundary = as_np(v.elem_boundary, (v.n_elems,))
            self.ghost_global_id = as_np(v.ghost_global_id, (v.n_ghost_nodes,))
            nn = v.n_nbrs
            self.nbr_rank = [v.nbr_rank[i] for i in range(nn)]
            so = [v.send_offsets[i] for i in range(nn + 1)]
            go = [v.ghost_offsets[i] for i in range(nn + 1)]
            send = as_np(v.send_nodes, (so[-1],)) if so[-1] else np.zeros(0, np.int32)
            self.send_nodes = [send[so[i]:so[i + 1]] for i in range(nn)]
            self.ghost_ranges = [(go[i], go[i + 1]) for i in range(nn)]
        finally:
            lib.l3k_hostmesh_destroy(h)

    @property
    def n_local_nodes(self):
        return self.n_owned_nodes + self.n_ghost_nodes

    def dirichlet_mask(self, dofs_per_node, unknowns=(0,), sides=range(6)):
        """Byte mask over local dofs: the listed unknowns on the listed cube sides (BCDefinition::defineDirichlet,
        benchmarks/Diffusion3D.hpp:39-41)."""
        side_bits = 0
        for s in sides:
            side_bits |= 1 << s
        on = (self.node_boundary & side_bits) != 0
        mask = np.zeros((self.n_local_nodes, dofs_per_node), dtype=np.uint8)
        for u in unknowns:
            mask[on, u] = 1
        return mask.reshape(-1)

    def boundary_sides(self, sides=range(6)):
        """(element index, side) of this rank's element sides on the listed cube sides: the BoundaryViews of
        makeCubeMesh (mesh/primitives/CubeMesh.hpp:66-138).  Returns (int64[n], uint8[n])."""
        fe, fs = [], []
        for s in sides:
            e = np.nonzero(self.elem_boundary & (1 << s))[0]
            fe.append(e.astype(np.int64))
            fs.append(np.full(e.size, s, dtype=np.uint8))
        if not fe:
            return np.zeros(0, np.int64), np.zeros(0, np.uint8)
        return np.concatenate(fe), np.concatenate(fs)

    def node_coords(self):
        """Physical location of every local node (mesh/NodePhysicalLocation.hpp: tri-linear map of the element vertices
        at the GLL reference positions).  Returns float64 [n_local_nodes, 3]."""
        g = gll_nodes(self.order + 1)
        l = np.stack([(1 - g) / 2, (1 + g) / 2], axis=1)  # [n][2] linear shape functions at the GLL points
        n = self.order + 1
        # shape[(ix,iy,iz) lexicographic, v = i + 2j + 4k]
        shape = np.einsum("xi,yj,zk->zyxkji", l, l, l).reshape(n ** 3, 8)
        coords = np.zeros((self.n_local_nodes, 3))
        for e0 in range(0, self.n_elems, 65536):
            ev = self.elem_verts[e0:e0 + 65536]
            coords[self.elem_nodes[e0:e0 + 65536].reshape(-1)] = np.einsum("nv,evs->ens", shape, ev).reshape(-1, 3)
        return coords

    def synthetic_vector(self, dofs_per_node, seed=42, ncols=1):
        """x ~ U(-1,1) as a function of (partition-independent grid node id, dof, column): every partition of the same
        mesh sees the same global vector.  Counter-based splitmix64.  Shape (ncols, n_local_dofs)."""
        key = (self.node_grid_id.astype(np.uint64)[:, None] * np.uint64(dofs_per_node) +
               np.arange(dofs_per_node, dtype=np.uint64)[None, :]).reshape(-1)
        out = np.empty((ncols, key.size))
        for c in range(ncols):
            with np.errstate(over="ignore"):
                z = key + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(c) * np.uint64(0xD1B54A32D192ED03)
                z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
                z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
                z = z ^ (z >> np.uint64(31))
            out[c] = (z >> np.uint64(11)).astype(np.float64) * (2.0 / (1 << 53)) - 1.0
        return out


def synthetic_vector_torch(node_grid_id, dofs_per_node, device, seed=42, ncols=1):
    """Same function as CubePartition.synthetic_vector, evaluated with torch on `device` (int64 two's-complement
    arithmetic == uint64 arithmetic modulo 2^64; logical shifts emulated by masking)."""
    import torch

    def s64(v):  # python int -> signed 64-bit representative
        v &= (1 << 64) - 1
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)

    ids = torch.as_tensor(node_grid_id, dtype=torch.int64, device=device)
    key = (ids[:, None] * dofs_per_node + torch.arange(dofs_per_node, dtype=torch.int64, device=device)[None, :]).reshape(-1)
    out = torch.empty((ncols, key.numel()), dtype=torch.float64, device=device)
    for c in range(ncols):
        z = key + s64(seed * 0x9E3779B97F4A7C15 + c * 0xD1B54A32D192ED03)
        z = (z ^ lsr(z, 30)) * s64(0xBF58476D1CE4E5B9)
        z = (z ^ lsr(z, 27)) * s64(0x94D049BB133111EB)
        z = z ^ lsr(z, 31)
        out[c] = lsr(z, 11).to(torch.float64) * (2.0 / (1 << 53)) - 1.0
    return out


# ------------------------------------------------------------------------------------------------------- device
class Context:
    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        check(capi.load().l3k_ctx_create(device, C.c_void_p(stream or 0), C.byref(self._h)))
        self.device = device

    def set_stream(self, stream):
        check(capi.load().l3k_ctx_set_stream(self._h, C.c_void_p(stream or 0)))

    def set_deterministic(self, on=True):
        """Bitwise-reproducible element launches (colour by colour) for the meshes created from now on."""
        check(capi.load().l3k_ctx_set_deterministic(self._h, int(bool(on))))

    def set_reference_z0(self, on=True):
        """Applies pass Point{x, y, 0.} to domain kernels as the reference's hex sum-factorisation path does
        (algsys/SumFactorization.hpp:732); default off: the true point (l3k_ctx_set_reference_z0 in include/l3k.h)."""
        check(capi.load().l3k_ctx_set_reference_z0(self._h, int(bool(on))))

    def get_tuning(self):
        """The context's launch-route settings as a dict (l3k_tuning in include/l3k.h)."""
        t = capi.Tuning()
        check(capi.load().l3k_ctx_get_tuning(self._h, C.byref(t)))
        return {name: getattr(t, name) for name, _ in capi.Tuning._fields_}

    def set_tuning(self, **fields):
        t = capi.Tuning()
        check(capi.load().l3k_ctx_get_tuning(self._h, C.byref(t)))
        for name, value in fields.items():
            if name not in dict(capi.Tuning._fields_):
                raise L3KError(f"l3k_tuning has no field {name!r}")
            setattr(t, name, int(value))
        check(capi.load().l3k_ctx_set_tuning(self._h, C.byref(t)))

    @contextlib.contextmanager
    def tuning(self, **fields):
        """with ctx.tuning(generic_below=0): ...  -- the settings inside the block, the previous ones afterwards."""
        before = self.get_tuning()
        self.set_tuning(**fields)
        try:
            yield self
        finally:
            self.set_tuning(**before)

    def synchronize(self):
        check(capi.load().l3k_ctx_synchronize(self._h))

    def __del__(self):
        if getattr(self, "_h", None) and capi is not None:  # (module globals may be gone at interpreter shutdown)
            capi.load().l3k_ctx_destroy(self._h)
            self._h = None


def elevate_order(ctx, conn, n_vertices, order):
    """Order elevation of an order-1 hex mesh on the device (l3k_elevate_order; mesh::convertMeshToOrder +
    LocalMeshView's numbering): conn [n_elems][8] vertex ids (local vertex i + 2j + 4k) -> (elem_nodes
    [n_elems][(order+1)^3] uint32, n_nodes, n_noninternal)."""
    lib = capi.load()
    c = np.ascontiguousarray(conn, dtype=np.uint32).reshape(-1, 8)
    out = np.empty((c.shape[0], (order + 1) ** 3), dtype=np.uint32)
    nn, nni = C.c_int64(), C.c_int64()
    capi.check(lib.l3k_elevate_order(ctx._h, c.shape[0], c.ctypes.data_as(capi.c_uint32_p), int(n_vertices), int(order),
                                     out.ctypes.data_as(capi.c_uint32_p), C.byref(nn), C.byref(nni)))
    return out, nn.value, nni.value


class ElevatedHexMesh:
    """A single-rank order-p hex mesh made from an unstructured order-1 one on the device: the mesh object DeviceMesh
    and the operators take (same attributes as CubePartition; no ghosts, no neighbours)."""

    def __init__(self, ctx, verts, conn, order):
        verts = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 3)
        conn = np.ascontiguousarray(conn, dtype=np.uint32).reshape(-1, 8)
        self.dim, self.order, self.rank, self.parts = 3, order, 0, (1, 1, 1)
        self.elem_nodes, n_nodes, self.n_noninternal = elevate_order(ctx, conn, verts.shape[0], order)
        self.elem_verts = verts[conn.astype(np.int64)]  # [n_elems][8][3]
        self.n_elems = self.n_interior_elems = conn.shape[0]
        self.n_owned_nodes, self.n_ghost_nodes = n_nodes, 0
        self.global_node_base, self.n_global_nodes = 0, n_nodes
        self.nbr_rank, self.send_nodes, self.ghost_ranges = [], [], []

    n_local_nodes = CubePartition.n_local_nodes
    node_coords = CubePartition.node_coords


class DeviceMesh:
    def __init__(self, ctx, part, dofs_per_node, dirichlet=None):
        self.ctx, self.part, self.dofs_per_node = ctx, part, dofs_per_node
        d = capi.MeshDesc()
        d.dim, d.order = part.dim, part.order
        d.n_elems, d.n_interior_elems = part.n_elems, part.n_interior_elems
        en = np.ascontiguousarray(part.elem_nodes, dtype=np.uint32)
        ev = np.ascontiguousarray(part.elem_verts, dtype=np.float64)
        d.elem_nodes = en.ctypes.data_as(capi.c_uint32_p)
        d.elem_verts = ev.ctypes.data_as(capi.c_double_p)
        d.n_owned_nodes, d.n_ghost_nodes = part.n_owned_nodes, part.n_ghost_nodes
        d.dofs_per_node = dofs_per_node
        if dirichlet is not None:
            dm = np.ascontiguousarray(dirichlet, dtype=np.uint8)
            if dm.size != part.n_local_nodes * dofs_per_node:
                raise L3KError("dirichlet mask must cover every local dof")
            d.dirichlet = dm.ctypes.data_as(capi.c_uint8_p)
        self._h = C.c_void_p()
        check(capi.load().l3k_mesh_create(ctx._h, C.byref(d), C.byref(self._h)))
        self.n_owned_dofs = part.n_owned_nodes * dofs_per_node
        self.n_ghost_dofs = part.n_ghost_nodes * dofs_per_node

    def __del__(self):
        if getattr(self, "_h", None) and capi is not None:
            capi.load().l3k_mesh_destroy(self._h)
            self._h = None


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _blob(kernel_params):
    if kernel_params is None:
        return None, 0, None
    arr = np.ascontiguousarray(kernel_params, dtype=np.float64)
    return arr.ctypes.data_as(C.c_void_p), arr.nbytes, arr


def _sides(face_elem, face_side):
    fe = np.ascontiguousarray(face_elem, dtype=np.int64)
    fs = np.ascontiguousarray(face_side, dtype=np.uint8)
    if fe.shape != fs.shape or fe.ndim != 1:
        raise L3KError("face_elem and face_side must be 1-D arrays of the same length")
    return fe, fs


class BoundaryTerm:
    """A boundary equation kernel on a list of element sides: assembleProblem(kernel, boundary_ids) of the reference
    (algsys/MatrixFreeSystem.hpp:58-68).  Attach it to a MatrixFreeSystem, or use apply / diag_rhs directly."""

    def __init__(self, mesh, kernel_id, face_elem, face_side, kernel_params=None, asm_opts=(1, 0, 0), field_inds=None,
                 n_rhs=1):
        self.mesh, self.ctx, self.kernel_id, self.n_rhs = mesh, mesh.ctx, kernel_id, n_rhs
        self.info = kernel_info(kernel_id)
        blob, nbytes, self._keep = _blob(kernel_params)
        opts = capi.AsmOpts(*asm_opts)
        fi = None if field_inds is None else (C.c_int * len(field_inds))(*field_inds)
        fe, fs = _sides(face_elem, face_side)
        self.n_faces = fe.size
        self._h = C.c_void_p()
        check(capi.load().l3k_bnd_create(self.ctx._h, mesh._h, kernel_id, blob, nbytes, C.byref(opts), fi, n_rhs, fe.size,
                                         fe.ctypes.data_as(capi.c_int64_p), fs.ctypes.data_as(capi.c_uint8_p),
                                         C.byref(self._h)))
        self._fields = None

    def __del__(self):
        if getattr(self, "_h", None) and capi is not None:
            capi.load().l3k_bnd_destroy(self._h)
            self._h = None

    def set_fields(self, fields):
        if fields is not None and (fields.dim() != 2 or not fields.is_contiguous()):
            raise L3KError("fields must be a contiguous (n_fields, n_local_nodes) tensor")
        self._fields = fields
        check(capi.load().l3k_bnd_set_fields(self._h, _ptr(fields), 0 if fields is None else fields.shape[1]))

    def set_time(self, t):
        check(capi.load().l3k_bnd_set_time(self._h, float(t)))

    def apply(self, X, Y, alpha=1.0, which=2, XG=None, YG=None):
        """Y += alpha * A_b X"""
        nc, ldx = MatrixFreeSystem._cols(X)
        _, ldy = MatrixFreeSystem._cols(Y)
        ldxg = XG.shape[1] if XG is not None else 0
        ldyg = YG.shape[1] if YG is not None else 0
        check(capi.load().l3k_bnd_apply(self._h, which, _ptr(X), ldx, _ptr(XG), ldxg, _ptr(Y), ldy, _ptr(YG), ldyg, nc,
                                        alpha))
        return Y

    def diag_rhs(self, diag, rhs, dirichlet_vals=None, which=2, diag_ghost=None, rhs_ghost=None):
        """diag += diag(A_b), rhs += B_b^T W (f_b - B_b g) (accumulating; no Dirichlet finalisation)"""
        g = dirichlet_vals
        ldg = 0 if g is None else g.shape[1]
        ldrg = 0 if rhs_ghost is None else rhs_ghost.shape[1]
        check(capi.load().l3k_bnd_diag_rhs(self._h, which, _ptr(g), ldg, _ptr(diag), _ptr(rhs), rhs.shape[1],
                                           _ptr(diag_ghost), _ptr(rhs_ghost), ldrg))
        return diag, rhs


def integrate(mesh, residual_id, fields=None, kernel_params=None, asm_opts=(1, 0, 0), time=0.0, square=False,
              face_elem=None, face_side=None):
    """evalLocalIntegral (post/Integral.hpp:54-111): this rank's integral of a residual kernel over all elements, or over
    the listed element sides.  fields: contiguous (n_fields, n_local_nodes) device tensor.  Returns a numpy array
    [n_equations]; multi-rank callers all-reduce it (computeIntegral :113-128)."""
    info = residual_info(residual_id)
    blob, nbytes, keep = _blob(kernel_params)
    opts = capi.AsmOpts(*asm_opts)
    if fields is not None and (fields.dim() != 2 or not fields.is_contiguous()):
        raise L3KError("fields must be a contiguous (n_fields, n_local_nodes) tensor")
    out = np.zeros(info["n_equations"])
    if face_elem is None:
        nf, fe_p, fs_p = -1, None, None
    else:
        fe, fs = _sides(face_elem, face_side)
        nf, fe_p, fs_p = fe.size, fe.ctypes.data_as(capi.c_int64_p), fs.ctypes.data_as(capi.c_uint8_p)
    check(capi.load().l3k_integrate(mesh.ctx._h, mesh._h, residual_id, blob, nbytes, C.byref(opts), _ptr(fields),
                                    0 if fields is None else fields.shape[1], float(time), int(square), nf, fe_p, fs_p,
                                    out.ctypes.data_as(capi.c_double_p)))
    return out


def values_at_nodes(mesh, residual_id, dof_inds, values, fields=None, kernel_params=None, time=0.0, face_elem=None,
                    face_side=None):
    """computeValuesAtNodes (algsys/ComputeValuesAtNodes.hpp) on one rank -- the engine of setDirichletBCValues / setValues:
    evaluates the residual kernel at the nodes of the listed element sides (or of all elements), averages the
    contributions per dof and writes them into `values` (1-D device tensor over the local dofs); other entries keep their
    content.  Returns `values`."""
    import torch
    info = residual_info(residual_id)
    if len(dof_inds) != info["n_equations"]:
        raise L3KError("one dof index per equation of the kernel")
    blob, nbytes, keep = _blob(kernel_params)
    di = (C.c_int * len(dof_inds))(*dof_inds)
    s = torch.zeros_like(values)
    c = torch.zeros_like(values)
    if face_elem is None:
        nf, fe_p, fs_p = -1, None, None
    else:
        fe, fs = _sides(face_elem, face_side)
        nf, fe_p, fs_p = fe.size, fe.ctypes.data_as(capi.c_int64_p), fs.ctypes.data_as(capi.c_uint8_p)
    lib = capi.load()
    check(lib.l3k_values_at_nodes(mesh.ctx._h, mesh._h, residual_id, blob, nbytes, _ptr(fields),
                                  0 if fields is None else fields.shape[1], float(time), nf, fe_p, fs_p, di, _ptr(s), _ptr(c)))
    check(lib.l3k_average_values(mesh.ctx._h, _ptr(s), _ptr(c), values.numel(), _ptr(values)))
    return values


def update_solution(mesh, X, sol_inds, fields, sol_man_inds, XG=None):
    """MatrixFreeSystem::updateSolution (algsys/MatrixFreeSystem.hpp:1231-1273): the solution's per-node dofs `sol_inds` of every
    column of X (ncols, n_owned_dofs) -- ghost rows from XG (ncols, n_ghost_dofs), the imported values -- into the fields
    `sol_man_inds` (one per (index, column), index-major) of the SoA field storage `fields` (n_fields, n_local_nodes): what a
    kernel's FieldAccess reads at the next assembly (time stepping, Newton iterations)."""
    nc, ldx = MatrixFreeSystem._cols(X)
    if fields.dim() != 2 or not fields.is_contiguous():
        raise L3KError("fields must be a contiguous (n_fields, n_local_nodes) tensor")
    if len(sol_man_inds) != len(sol_inds) * nc:
        raise L3KError("Source and destination indices lengths must match")  # MatrixFreeSystem.hpp:1237-1238
    si = (C.c_int * len(sol_inds))(*[int(i) for i in sol_inds])
    di = (C.c_int * len(sol_man_inds))(*[int(i) for i in sol_man_inds])
    check(capi.load().l3k_update_solution(mesh.ctx._h, mesh._h, _ptr(X), ldx, _ptr(XG), 0 if XG is None else XG.shape[1], nc,
                                          len(sol_inds), si, di, _ptr(fields), fields.shape[1], fields.shape[0]))
    return fields


def norm_l2(mesh, residual_id, fields=None, kernel_params=None, asm_opts=(1, 0, 0), time=0.0, face_elem=None,
            face_side=None, group=None):
    """computeNormL2 (post/NormL2.hpp:31-62): sqrt of the integral of the squared residual with doubled quadrature
    orders; all-reduced over `group` when torch.distributed is initialised."""
    doubled = (2 * asm_opts[0], 2 * asm_opts[1], asm_opts[2])
    sq = integrate(mesh, residual_id, fields, kernel_params, doubled, time, True, face_elem, face_side)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        import torch
        t = torch.from_numpy(sq)
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, group=group)
        sq = t.cpu().numpy()
    return np.sqrt(sq)


class MatrixFreeSystem:
    """algsys::MatrixFreeSystem for one rank: kernel + mesh -> operator.  apply() is Operator::apply
    (Y <- alpha*A*X + beta*Y, algsys/MatrixFreeSystem.hpp:34-41,1038)."""

    def __init__(self, mesh, kernel_id, kernel_params=None, asm_opts=(1, 0, 0), field_inds=None, n_rhs=1):
        self.mesh, self.ctx, self.kernel_id, self.n_rhs = mesh, mesh.ctx, kernel_id, n_rhs
        self.info = kernel_info(kernel_id)
        blob = None
        nbytes = 0
        if kernel_params is not None:
            arr = np.ascontiguousarray(kernel_params, dtype=np.float64)
            blob, nbytes = arr.ctypes.data_as(C.c_void_p), arr.nbytes
        opts = capi.AsmOpts(*asm_opts)
        fi = None
        if field_inds is not None:
            fi = (C.c_int * len(field_inds))(*field_inds)
        self._h = C.c_void_p()
        check(capi.load().l3k_mf_create(self.ctx._h, mesh._h, kernel_id, blob, nbytes, C.byref(opts), fi, n_rhs,
                                        C.byref(self._h)))
        self._fields = None
        self._boundary_terms = []

    def __del__(self):
        if getattr(self, "_h", None) and capi is not None:
            capi.load().l3k_mf_destroy(self._h)
            self._h = None

    def attach_boundary(self, term):
        """Registers a BoundaryTerm: apply / apply_elems / diag_rhs include it from now on."""
        check(capi.load().l3k_mf_attach_boundary(self._h, term._h))
        self._boundary_terms.append(term)  # keeps the term alive as long as the system

    # post::FieldAccess: SoA (n_fields, n_local_nodes) device tensor, kept alive here
    def set_fields(self, fields):
        if fields is not None and (fields.dim() != 2 or not fields.is_contiguous()):
            raise L3KError("fields must be a contiguous (n_fields, n_local_nodes) tensor")
        self._fields = fields
        check(capi.load().l3k_mf_set_fields(self._h, _ptr(fields), 0 if fields is None else fields.shape[1]))

    def set_time(self, t):
        check(capi.load().l3k_mf_set_time(self._h, float(t)))

    def route(self, which=2, ncols=1, with_energy=False):
        """One line naming the kernel (template, variant, lanes, LDS, grid) the element launch of apply_elems(which, ...,
        ncols columns) takes right now -- decided by the launcher's own code (l3k_mf_route)."""
        buf = C.create_string_buffer(512)
        check(capi.load().l3k_mf_route(self._h, which, ncols, int(with_energy), buf, len(buf)))
        return buf.value.decode()

    @staticmethod
    def _cols(t):
        if t.dim() != 2 or t.stride(1) != 1:
            raise L3KError("multivectors are (ncols, ld) tensors with unit stride along rows")
        return t.shape[0], (t.stride(0) if t.shape[0] > 1 else t.shape[1])

    def apply(self, X, Y, alpha=1.0, beta=0.0):
        nc, ldx = self._cols(X)
        nc2, ldy = self._cols(Y)
        if nc != nc2:
            raise L3KError("X and Y must have the same number of columns")  # MatrixFreeSystem.hpp:1035
        check(capi.load().l3k_mf_apply(self._h, _ptr(X), ldx, _ptr(Y), ldy, nc, alpha, beta))
        return Y

    # split-phase pieces (used by DistributedOperator)
    def apply_energy(self, X, Y, S):
        """Y <- A X (one column) and S[1] <- <X, A X> in one pass (l3k_mf_apply_energy): S is the PCG's device scalar
        block (8 doubles)."""
        check(capi.load().l3k_mf_apply_energy(self._h, _ptr(X), _ptr(Y), _ptr(S)))
        return Y

    def energy_begin(self, S):
        check(capi.load().l3k_mf_energy_begin(self._h, _ptr(S)))

    def energy_end(self, X):
        """True if S[1] now holds this rank's share of <X, A X> (else take the dot product)."""
        fused = C.c_int(0)
        check(capi.load().l3k_mf_energy_end(self._h, _ptr(X), C.byref(fused)))
        return bool(fused.value)

    def scale(self, Y, beta):
        nc, ldy = self._cols(Y)
        check(capi.load().l3k_mf_scale(self._h, _ptr(Y), ldy, nc, beta))

    def apply_elems(self, which, X, XG, Y, YG, alpha, beta):
        """beta must be the value scale() was called with (see l3k_mf_scale in include/l3k.h)."""
        nc, ldx = self._cols(X)
        _, ldy = self._cols(Y)
        ldxg = XG.shape[1] if XG is not None else 0
        ldyg = YG.shape[1] if YG is not None else 0
        check(capi.load().l3k_mf_apply_elems(self._h, which, _ptr(X), ldx, _ptr(XG), ldxg, _ptr(Y), ldy, _ptr(YG), ldyg,
                                             nc, alpha, beta))

    def dirichlet_rows(self, X, Y, alpha):
        nc, ldx = self._cols(X)
        _, ldy = self._cols(Y)
        check(capi.load().l3k_mf_dirichlet_rows(self._h, _ptr(X), ldx, _ptr(Y), ldy, nc, alpha))

    def pack_rows(self, src, idx, dst):
        nc, ld = self._cols(src)
        check(capi.load().l3k_pack_rows(self.ctx._h, _ptr(src), ld, nc, _ptr(idx), idx.numel(), _ptr(dst)))

    def unpack_add_rows(self, src, idx, dst):
        nc, ld = self._cols(dst)
        check(capi.load().l3k_unpack_add_rows(self.ctx._h, _ptr(src), idx.numel(), _ptr(idx), _ptr(dst), ld, nc))

    def diag_rhs(self, dirichlet_vals=None, which=2, diag=None, rhs=None, diag_ghost=None, rhs_ghost=None,
                 finalize=True):
        """computeDiagAndRhs (algsys/MatrixFreeSystem.hpp:888-941): diag(A) and the rhs with Dirichlet lifting.
        dirichlet_vals: (n_rhs, n_local_dofs) tensor or None (= 0).  Returns (diag [n_owned], rhs (n_rhs, n_owned));
        accumulates into the given tensors (the caller zeroes them), allocates zeroed ones otherwise."""
        import torch
        n_owned, n_ghost = self.mesh.n_owned_dofs, self.mesh.n_ghost_dofs
        dev = "cuda"
        if diag is None:
            diag = torch.zeros(n_owned, dtype=torch.float64, device=dev)
        if rhs is None:
            rhs = torch.zeros((self.n_rhs, n_owned), dtype=torch.float64, device=dev)
        if n_ghost and diag_ghost is None:
            diag_ghost = torch.zeros(n_ghost, dtype=torch.float64, device=dev)
            rhs_ghost = torch.zeros((self.n_rhs, n_ghost), dtype=torch.float64, device=dev)
        g = dirichlet_vals
        ldg = 0 if g is None else g.shape[1]
        check(capi.load().l3k_mf_diag_rhs(self._h, which, _ptr(g), ldg, _ptr(diag), _ptr(rhs), rhs.shape[1],
                                          _ptr(diag_ghost), _ptr(rhs_ghost), max(n_ghost, 1), int(finalize)))
        return diag, rhs

    def dirichlet_finalize(self, dirichlet_vals, diag, rhs):
        """diag = 1, rhs = g on the owned Dirichlet rows (the last step of computeDiagAndRhs, after the export)"""
        g = dirichlet_vals
        check(capi.load().l3k_mf_dirichlet_finalize(self._h, _ptr(g), 0 if g is None else g.shape[1], _ptr(diag), _ptr(rhs),
                                                    rhs.shape[1]))

    def local_assemble(self, first=0, count=None, want_K=True, want_F=True, want_checksum=False):
        """assembleLocalSystem for elements [first, first+count) (algsys/AssembleLocalSystem.hpp:234-256).
        Returns (K [count, Nd, Nd] row-major, F [count, n_rhs, Nd] i.e. column-major Nd x n_rhs per element, checksum)."""
        import torch
        count = self.mesh.part.n_elems - first if count is None else count
        Nd = (self.mesh.part.order + 1) ** 3 * self.info["n_unknowns"]
        K = torch.empty((count, Nd, Nd), dtype=torch.float64, device="cuda") if want_K else None
        F = torch.zeros((count, self.n_rhs, Nd), dtype=torch.float64, device="cuda") if want_F else None
        cs = torch.empty(count, dtype=torch.float64, device="cuda") if want_checksum else None
        check(capi.load().l3k_local_assemble(self._h, first, count, _ptr(K), _ptr(F), _ptr(cs)))
        return K, F, cs

    def local_assemble_into(self, K, first=0, count=None):
        """The element matrices of [first, first + count) into the caller's tensor K [count, Nd, Nd] (row-major, bitwise symmetric)."""
        count = K.shape[0] if count is None else count
        Nd = (self.mesh.part.order + 1) ** 3 * self.info["n_unknowns"]
        if tuple(K.shape[1:]) != (Nd, Nd) or K.shape[0] < count or not K.is_contiguous() or str(K.dtype) != "torch.float64":
            raise L3KError(f"local_assemble_into: K must be a contiguous float64 tensor [>= {count}, {Nd}, {Nd}]")
        check(capi.load().l3k_local_assemble(self._h, first, count, _ptr(K), None, None))
        return K

    def assembled_scatter(self, K, F, row_ptr, col_ind, values, rhs, first=0, skip_dirichlet=False):
        """scatterLocalSystem for the batch [first, first + len(K)) (algsys/ScatterLocalSystem.hpp:24-54): K [count, Nd, Nd]
        and F [count, n_rhs, Nd] as local_assemble returns them, summed into `values` (over the caller's CSR graph
        row_ptr int64 / col_ind int32, device tensors) and `rhs` [n_rhs, n_local_dofs].  Returns the number of entries
        that were not in the graph."""
        import ctypes as C
        count = (K if K is not None else F).shape[0]
        missing = C.c_int64(0)
        check(capi.load().l3k_assembled_scatter(self._h, first, count, _ptr(K), _ptr(F), _ptr(row_ptr), _ptr(col_ind), _ptr(values),
                                                _ptr(rhs), 0 if rhs is None else rhs.stride(0) if rhs.dim() == 2 else rhs.numel(),
                                                int(skip_dirichlet), C.byref(missing)))
        return missing.value

    def local_assemble_tiled(self, first=0, count=None):
        """K_e of the elements [first, first + count) in the tiled layout of l3k_local_assemble_tiled: a tensor
        [count, U, U, n, n, n, n, n, n] indexed [e, u, u', bx', bz, bx, by, by', bz']."""
        import torch
        count = self.mesh.part.n_elems - first if count is None else count
        n, U = self.mesh.part.order + 1, self.info["n_unknowns"]
        Kt = torch.empty((count, U, U, n, n, n, n, n, n), dtype=torch.float64, device="cuda")
        check(capi.load().l3k_local_assemble_tiled(self._h, first, count, _ptr(Kt)))
        return Kt

    def assemble_global(self, row_ptr, col_ind, values, rhs=None, first=0, count=None, skip_dirichlet=False, workspace_bytes=0):
        """assembleGlobalSystem for the elements [first, first + count) (algsys/AssembleGlobalSystem.hpp:20-53): element systems
        formed and summed into `values` (over the CSR graph row_ptr int64 / col_ind int32) and `rhs` [n_rhs, n_local_dofs] inside
        the library, assembly and scatter of consecutive sub-batches overlapped on two streams.  Returns the number of entries
        outside the graph."""
        import ctypes as C
        count = self.mesh.part.n_elems - first if count is None else count
        missing = C.c_int64(0)
        check(capi.load().l3k_assemble_global(self._h, first, count, _ptr(row_ptr), _ptr(col_ind), _ptr(values), _ptr(rhs),
                                              0 if rhs is None else rhs.stride(0) if rhs.dim() == 2 else rhs.numel(),
                                              int(skip_dirichlet), workspace_bytes, C.byref(missing)))
        return missing.value

    def new_ghost_buffer(self, ncols, like):
        import torch
        return torch.zeros((ncols, max(self.mesh.n_ghost_dofs, 1)), dtype=torch.float64, device=like.device)
