"""ctypes binding of oracle/liboracle.so -- the CPU oracle (TEST INFRASTRUCTURE, see oracle/oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")

KERNEL_DIFFUSION3D = 0
KERNEL_DIFFUSION3D_VAR = 1
KERNEL_DIFFUSION2D = 2
KERNEL_DIFFUSION2D_VAR = 3
KERNEL_ADVDIFF3D = 4
KERNEL_MASS3D = 8
KERNEL_DIFFUSION3D_POINT = 10  # operators and rhs read point.space.{x,y,z} and point.time
KERNEL_ADVECTION3D = 11  # U = E = 1, F = 3
KERNEL_DIVCURL3D = 12  # U = 3, E = 4
KERNEL_NS3D = 13  # benchmarks/Kernels.hpp:3-65: U = 7, E = 8, F = 7

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class OrcMesh(C.Structure):
    _fields_ = [
        ("dim", C.c_int), ("p", C.c_int), ("nq", C.c_int),
        ("n_elems", C.c_int64),
        ("elem_nodes", C.POINTER(C.c_uint32)),
        ("elem_verts", _dp),
        ("n_local_nodes", C.c_int64),
        ("dofs_per_node", C.c_int),
        ("field_inds", _ip),
        ("dirichlet", C.POINTER(C.c_uint8)),
        ("fields", _dp),
    ]


def build(path=None, march=None):
    """(Re)build the oracle shared library; returns its path."""
    out = path or os.path.join(_ORACLE_DIR, "liboracle.so")
    cmd = ["make", "-s", "-C", _ORACLE_DIR, f"OUT={out}"]
    if march:
        cmd.append(f"MARCH={march}")
    subprocess.run(cmd, check=True)
    return out


_lib = None


def lib(path=None):
    global _lib
    if _lib is not None and path is None:
        return _lib
    so = path or os.path.join(_ORACLE_DIR, "liboracle.so")
    if not os.path.exists(so):
        build(so)
    L = C.CDLL(so)
    L.orc_last_error.restype = C.c_char_p
    if path is None:
        _lib = L
    return L


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _chk(rc, L=None):
    if rc != 0:
        raise RuntimeError(f"oracle error {rc}: {(L or lib()).orc_last_error().decode()}")


def kernel_params(kid):
    v = [C.c_int() for _ in range(4)]
    _chk(lib().orc_kernel_params(kid, *[C.byref(x) for x in v]))
    return dict(dim=v[0].value, E=v[1].value, U=v[2].value, F=v[3].value)


def gll_nodes(n):
    x = np.zeros(n)
    _chk(lib().orc_gll_nodes(n, _d(x)))
    return x


def gl_rule(nq):
    x, w = np.zeros(nq), np.zeros(nq)
    _chk(lib().orc_gl_rule(nq, _d(x), _d(w)))
    return x, w


def legendre_coefs(n):
    """math/Legendre.hpp: coefficients of P_n, highest power first"""
    c = np.zeros(n + 1)
    _chk(lib().orc_legendre_coefs(n, _d(c)))
    return c


def lagrange_interp(x, y):
    """math/LagrangeInterpolation.hpp: monomial coefficients (highest power first) of the interpolant"""
    x, y = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64)
    c = np.zeros(len(x))
    _chk(lib().orc_lagrange_interp(len(x), _d(x), _d(y), _d(c)))
    return c


def poly_eval(coefs, x):
    L = lib()
    L.orc_poly_eval.restype = C.c_double
    coefs = np.ascontiguousarray(coefs, dtype=np.float64)
    return L.orc_poly_eval(len(coefs), _d(coefs), C.c_double(x))


def n_qps1d(p, value_order=1, derivative_order=0):
    return lib().orc_n_qps1d(p, value_order, derivative_order)


def basis_1d(p, nq):
    I, D = np.zeros((p + 1, nq)), np.zeros((p + 1, nq))
    _chk(lib().orc_basis_1d(p, nq, _d(I), _d(D)))
    return I, D


def lagrange_1d(p, x):
    v, d = np.zeros(p + 1), np.zeros(p + 1)
    _chk(lib().orc_lagrange_1d(p, C.c_double(x), _d(v), _d(d)))
    return v, d


def ref_basis_at_qps(dim, p, nq):
    N, nqp = (p + 1) ** dim, nq ** dim
    vals, ders = np.zeros((nqp, N)), np.zeros((nqp, dim, N))
    w, pts = np.zeros(nqp), np.zeros((nqp, dim))
    _chk(lib().orc_ref_basis_at_qps(dim, p, nq, _d(vals), _d(ders), _d(w), _d(pts)))
    return vals, ders, w, pts


def oddeven_check(p, nq, cols, in_back, in_fwd):
    err = np.zeros(4)
    _chk(lib().orc_oddeven_check(p, nq, cols, _d(np.ascontiguousarray(in_back)), _d(np.ascontiguousarray(in_fwd)),
                                 _d(err)))
    return err


def jacobi_mat(dim, verts, point):
    J = np.zeros((dim, dim))
    verts = np.ascontiguousarray(verts, dtype=np.float64)
    point = np.ascontiguousarray(point, dtype=np.float64)
    _chk(lib().orc_jacobi_mat(dim, _d(verts), _d(point), _d(J)))
    return J


def map_to_physical(dim, verts, point):
    out = np.zeros(3)
    verts = np.ascontiguousarray(verts, dtype=np.float64)
    point = np.ascontiguousarray(point, dtype=np.float64)
    _chk(lib().orc_map_to_physical(dim, _d(verts), _d(point), _d(out)))
    return out


def node_location(dim, p, verts, node):
    out = np.zeros(3)
    verts = np.ascontiguousarray(verts, dtype=np.float64)
    _chk(lib().orc_node_location(dim, p, _d(verts), node, _d(out)))
    return out


def _prep(kid, p, verts, node_fields, kparams):
    kp = kernel_params(kid)
    N = (p + 1) ** kp["dim"]
    verts = np.ascontiguousarray(verts, dtype=np.float64)
    nf = None if node_fields is None else np.ascontiguousarray(node_fields, dtype=np.float64)
    if kp["F"] > 0:
        assert nf is not None and nf.shape == (N, kp["F"])
    kpar = None if kparams is None else np.ascontiguousarray(kparams, dtype=np.float64)
    return kp, N, verts, nf, kpar


def assemble_local(kid, p, nq, R, verts, node_fields=None, kparams=None, time=0.0):
    """Returns (K [Nd,Nd], F [Nd,R])."""
    kp, N, verts, nf, kpar = _prep(kid, p, verts, node_fields, kparams)
    Nd = N * kp["U"]
    K = np.zeros((Nd, Nd))
    F = np.zeros((Nd, R), order="F")
    _chk(lib().orc_assemble_local(kid, p, nq, R, _d(verts), _d(nf), _d(kpar), C.c_double(time), _d(K), _d(F)))
    return K, F


def apply_local(kid, p, nq, verts, x, node_fields=None, kparams=None, time=0.0):
    kp, N, verts, nf, kpar = _prep(kid, p, verts, node_fields, kparams)
    x = np.asfortranarray(x, dtype=np.float64)
    y = np.zeros_like(x, order="F")
    _chk(lib().orc_apply_local(kid, p, nq, x.shape[1], _d(verts), _d(nf), _d(kpar), C.c_double(time), _d(x), _d(y)))
    return y


def diag_rhs_local(kid, p, nq, R, verts, dir_inds=None, dir_vals=None, node_fields=None, kparams=None, time=0.0):
    kp, N, verts, nf, kpar = _prep(kid, p, verts, node_fields, kparams)
    Nd = N * kp["U"]
    diag = np.zeros(Nd)
    rhs = np.zeros((Nd, R), order="F")
    nd = 0 if dir_inds is None else len(dir_inds)
    di = None if nd == 0 else np.ascontiguousarray(dir_inds, dtype=np.int32)
    dv = None if nd == 0 else np.asfortranarray(dir_vals, dtype=np.float64)
    _chk(lib().orc_diag_rhs_local(kid, p, nq, R, _d(verts), _d(nf), _d(kpar), C.c_double(time), nd,
                                  None if di is None else di.ctypes.data_as(_ip), _d(dv), _d(diag), _d(rhs)))
    return diag, rhs


def apply_sumfact(kid, p, nq, verts, x, node_fields=None, kparams=None, time=0.0, odd_even=False, true_z=False):
    kp, N, verts, nf, kpar = _prep(kid, p, verts, node_fields, kparams)
    x = np.asfortranarray(x, dtype=np.float64)
    y = np.zeros_like(x, order="F")
    _chk(lib().orc_apply_sumfact(kid, p, nq, x.shape[1], int(odd_even), int(true_z), _d(verts), _d(nf), _d(kpar),
                                 C.c_double(time), _d(x), _d(y)))
    return y


def update_solution(mesh, x, sol_inds, fields, sol_man_inds):
    """MatrixFreeSystem::updateSolution: x [n_local_dofs, n_rhs] Fortran-ordered, fields [n_fields, n_local_nodes] (in place)."""
    x = np.asfortranarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x.reshape(-1, 1, order="F")
    assert fields.flags.c_contiguous and fields.shape[1] == mesh.n_local_nodes
    si, di = np.ascontiguousarray(sol_inds, dtype=np.int32), np.ascontiguousarray(sol_man_inds, dtype=np.int32)
    _chk(lib().orc_update_solution(C.byref(mesh.struct), _d(x), C.c_size_t(x.shape[0]), x.shape[1], len(si), si.ctypes.data_as(_ip),
                                   di.ctypes.data_as(_ip), _d(fields), fields.shape[0]))
    return fields


def set_reference_z0(on):
    """mf_apply passes z = 0 to domain kernels as the reference's evalAtHexQPs does (SumFactorization.hpp:732); default off."""
    _chk(lib().orc_set_reference_z0(int(bool(on))))


class MeshView:
    """Keeps numpy arrays alive behind an orc_mesh struct."""

    def __init__(self, dim, p, nq, elem_nodes, elem_verts, n_local_nodes, dofs_per_node, field_inds, dirichlet=None,
                 fields=None):
        self.elem_nodes = np.ascontiguousarray(elem_nodes, dtype=np.uint32)
        self.elem_verts = np.ascontiguousarray(elem_verts, dtype=np.float64)
        self.field_inds = np.ascontiguousarray(field_inds, dtype=np.int32)
        self.dirichlet = None if dirichlet is None else np.ascontiguousarray(dirichlet, dtype=np.uint8)
        self.fields = None if fields is None else np.ascontiguousarray(fields, dtype=np.float64)
        self.n_local_nodes = int(n_local_nodes)
        self.dofs_per_node = int(dofs_per_node)
        self.n_elems = self.elem_nodes.shape[0]
        s = OrcMesh()
        s.dim, s.p, s.nq = dim, p, nq
        s.n_elems = self.n_elems
        s.elem_nodes = self.elem_nodes.ctypes.data_as(C.POINTER(C.c_uint32))
        s.elem_verts = _d(self.elem_verts)
        s.n_local_nodes = self.n_local_nodes
        s.dofs_per_node = self.dofs_per_node
        s.field_inds = self.field_inds.ctypes.data_as(_ip)
        s.dirichlet = None if self.dirichlet is None else self.dirichlet.ctypes.data_as(C.POINTER(C.c_uint8))
        s.fields = _d(self.fields)
        self.struct = s

    @property
    def n_local_dofs(self):
        return self.n_local_nodes * self.dofs_per_node


def mf_apply(mesh, kid, x, y=None, alpha=1.0, beta=0.0, kparams=None, time=0.0, odd_even=False, e_begin=0,
             e_end=None, do_scale=True, do_dirichlet_rows=True, n_owned_dofs=None, nthreads=1, L=None):
    """y <- alpha A x + beta y on local dofs; x, y: [n_local_dofs, ncols] Fortran-ordered."""
    L = L or lib()
    x = np.asfortranarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x.reshape(-1, 1, order="F")
    if y is None:
        y = np.zeros_like(x, order="F")
    assert y.flags.f_contiguous and x.shape == y.shape
    kpar = None if kparams is None else np.ascontiguousarray(kparams, dtype=np.float64)
    e_end = mesh.n_elems if e_end is None else e_end
    n_owned = mesh.n_local_dofs if n_owned_dofs is None else n_owned_dofs
    rc = L.orc_mf_apply(C.byref(mesh.struct), kid, _d(kpar), C.c_double(time), int(odd_even), x.shape[1], _d(x),
                        C.c_size_t(x.shape[0]), _d(y), C.c_size_t(y.shape[0]), C.c_double(alpha), C.c_double(beta),
                        C.c_int64(e_begin), C.c_int64(e_end), int(do_scale), int(do_dirichlet_rows),
                        C.c_int64(n_owned), int(nthreads))
    _chk(rc, L)
    return y


def mf_diag_rhs(mesh, kid, R=1, dirichlet_vals=None, kparams=None, time=0.0, diag=None, rhs=None, e_begin=0,
                e_end=None, finalize=True, n_owned_dofs=None, nthreads=1):
    nl = mesh.n_local_dofs
    diag = np.zeros(nl) if diag is None else diag
    rhs = np.zeros((nl, R), order="F") if rhs is None else rhs
    g = None if dirichlet_vals is None else np.asfortranarray(dirichlet_vals, dtype=np.float64).reshape(nl, R, order="F")
    kpar = None if kparams is None else np.ascontiguousarray(kparams, dtype=np.float64)
    e_end = mesh.n_elems if e_end is None else e_end
    n_owned = nl if n_owned_dofs is None else n_owned_dofs
    _chk(lib().orc_mf_diag_rhs(C.byref(mesh.struct), kid, _d(kpar), C.c_double(time), R, _d(g), C.c_size_t(nl),
                               _d(diag), _d(rhs), C.c_size_t(nl), C.c_int64(e_begin), C.c_int64(e_end),
                               int(finalize), C.c_int64(n_owned), int(nthreads)))
    return diag, rhs


# ---- boundary terms and post-processing integrals ------------------------------------------
KERNEL_ADIABATIC2D = 5
KERNEL_ADIABATIC3D = 6
KERNEL_ROBIN3D = 7
KERNEL_NORMALFLUX3D = 9
KERNEL_ROBINPOINT3D = 14  # boundary kernel whose coefficients read point.space and point.time
RESIDUAL_DIFFUSION3D_ERROR = 0
RESIDUAL_LINEAR2D_ERROR = 1
RESIDUAL_LINEAR3D_ERROR = 2
RESIDUAL_UNIT2D = 3
RESIDUAL_UNIT3D = 4
RESIDUAL_COORDX2D = 5
RESIDUAL_COORDX3D = 6

_i64p = C.POINTER(C.c_int64)
_u8p = C.POINTER(C.c_uint8)


def side_basis_at_qps(dim, p, nq, side):
    N, nqp = (p + 1) ** dim, nq ** (dim - 1)
    vals, ders = np.zeros((nqp, N)), np.zeros((nqp, dim, N))
    w, pts = np.zeros(nqp), np.zeros((nqp, dim))
    _chk(lib().orc_side_basis_at_qps(dim, p, nq, side, _d(vals), _d(ders), _d(w), _d(pts)))
    return vals, ders, w, pts


def boundary_geometry(dim, verts, point, side):
    verts = np.ascontiguousarray(verts, dtype=np.float64)
    point = np.ascontiguousarray(point, dtype=np.float64)
    nrm, jac = np.zeros(3), C.c_double()
    _chk(lib().orc_boundary_geometry(dim, _d(verts), _d(point), side, _d(nrm), C.byref(jac)))
    return nrm[:dim], jac.value


def assemble_local_side(side, kid, p, nq, R, verts, node_fields=None, kparams=None, time=0.0):
    kp, N, verts, nf, kpar = _prep(kid, p, verts, node_fields, kparams)
    Nd = N * kp["U"]
    K = np.zeros((Nd, Nd))
    F = np.zeros((Nd, R), order="F")
    _chk(lib().orc_assemble_local_side(side, kid, p, nq, R, _d(verts), _d(nf), _d(kpar), C.c_double(time), _d(K), _d(F)))
    return K, F


def apply_local_side(side, kid, p, nq, verts, x, node_fields=None, kparams=None, time=0.0):
    kp, N, verts, nf, kpar = _prep(kid, p, verts, node_fields, kparams)
    x = np.asfortranarray(x, dtype=np.float64)
    y = np.zeros_like(x, order="F")
    _chk(lib().orc_apply_local_side(side, kid, p, nq, x.shape[1], _d(verts), _d(nf), _d(kpar), C.c_double(time), _d(x),
                                    _d(y)))
    return y


def diag_rhs_local_side(side, kid, p, nq, R, verts, dir_inds=None, dir_vals=None, node_fields=None, kparams=None,
                        time=0.0):
    kp, N, verts, nf, kpar = _prep(kid, p, verts, node_fields, kparams)
    Nd = N * kp["U"]
    diag = np.zeros(Nd)
    rhs = np.zeros((Nd, R), order="F")
    nd = 0 if dir_inds is None else len(dir_inds)
    di = None if nd == 0 else np.ascontiguousarray(dir_inds, dtype=np.int32)
    dv = None if nd == 0 else np.asfortranarray(dir_vals, dtype=np.float64)
    _chk(lib().orc_diag_rhs_local_side(side, kid, p, nq, R, _d(verts), _d(nf), _d(kpar), C.c_double(time), nd,
                                       None if di is None else di.ctypes.data_as(_ip), _d(dv), _d(diag), _d(rhs)))
    return diag, rhs


def residual_params(rid):
    v = [C.c_int() for _ in range(3)]
    _chk(lib().orc_residual_params(rid, *[C.byref(x) for x in v]))
    return dict(dim=v[0].value, E=v[1].value, F=v[2].value)


def integrate_local(side, rid, p, nq, verts, node_fields, square=False, kparams=None, time=0.0):
    rp = residual_params(rid)
    verts = np.ascontiguousarray(verts, dtype=np.float64)
    nf = None if rp["F"] == 0 else np.ascontiguousarray(node_fields, dtype=np.float64)
    assert nf is None or nf.shape == ((p + 1) ** rp["dim"], rp["F"])
    kpar = None if kparams is None else np.ascontiguousarray(kparams, dtype=np.float64)
    out = np.zeros(rp["E"])
    _chk(lib().orc_integrate_local(side, rid, p, nq, int(square), _d(verts), _d(nf), _d(kpar), C.c_double(time), _d(out)))
    return out


def _faces(face_elem, face_side):
    fe = np.ascontiguousarray(face_elem, dtype=np.int64)
    fs = np.ascontiguousarray(face_side, dtype=np.uint8)
    assert fe.shape == fs.shape
    return fe, fs


def mf_integrate(mesh, rid, nq, square=False, kparams=None, time=0.0, face_elem=None, face_side=None):
    """Integral (or integral of the square) of a residual kernel over the mesh / the listed element sides;
    mesh.fields must hold the kernel's F fields (SoA)."""
    rp = residual_params(rid)
    kpar = None if kparams is None else np.ascontiguousarray(kparams, dtype=np.float64)
    out = np.zeros(rp["E"])
    if face_elem is None:
        _chk(lib().orc_mf_integrate(C.byref(mesh.struct), rid, nq, int(square), _d(kpar), C.c_double(time),
                                    C.c_int64(-1), None, None, _d(out)))
    else:
        fe, fs = _faces(face_elem, face_side)
        _chk(lib().orc_mf_integrate(C.byref(mesh.struct), rid, nq, int(square), _d(kpar), C.c_double(time),
                                    C.c_int64(len(fe)), fe.ctypes.data_as(_i64p), fs.ctypes.data_as(_u8p), _d(out)))
    return out


def bnd_apply(mesh, kid, face_elem, face_side, x, y, alpha=1.0, kparams=None, time=0.0):
    """y += alpha * A_b x (in place on the Fortran-ordered y)."""
    fe, fs = _faces(face_elem, face_side)
    x = np.asfortranarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x.reshape(-1, 1, order="F")
    assert y.flags.f_contiguous and y.shape == x.shape
    kpar = None if kparams is None else np.ascontiguousarray(kparams, dtype=np.float64)
    _chk(lib().orc_bnd_apply(C.byref(mesh.struct), kid, _d(kpar), C.c_double(time), x.shape[1], C.c_int64(len(fe)),
                             fe.ctypes.data_as(_i64p), fs.ctypes.data_as(_u8p), _d(x), C.c_size_t(x.shape[0]), _d(y),
                             C.c_size_t(y.shape[0]), C.c_double(alpha)))
    return y


def bnd_diag_rhs(mesh, kid, face_elem, face_side, diag, rhs, dirichlet_vals=None, kparams=None, time=0.0):
    """diag, rhs accumulated in place."""
    fe, fs = _faces(face_elem, face_side)
    nl, R = rhs.shape
    assert rhs.flags.f_contiguous
    g = None if dirichlet_vals is None else np.asfortranarray(dirichlet_vals, dtype=np.float64).reshape(nl, R, order="F")
    kpar = None if kparams is None else np.ascontiguousarray(kparams, dtype=np.float64)
    _chk(lib().orc_bnd_diag_rhs(C.byref(mesh.struct), kid, _d(kpar), C.c_double(time), R, C.c_int64(len(fe)),
                                fe.ctypes.data_as(_i64p), fs.ctypes.data_as(_u8p), _d(g), C.c_size_t(nl), _d(diag),
                                _d(rhs), C.c_size_t(nl)))
    return diag, rhs


def values_at_nodes(mesh, rid, dof_inds, face_elem=None, face_side=None, kparams=None, time=0.0):
    """computeValuesAtNodes pieces: returns (sum, count) over local dofs; the nodal value is sum / count where count > 0."""
    kpar = None if kparams is None else np.ascontiguousarray(kparams, dtype=np.float64)
    di = np.ascontiguousarray(dof_inds, dtype=np.int32)
    assert di.size == residual_params(rid)["E"]
    s, c = np.zeros(mesh.n_local_dofs), np.zeros(mesh.n_local_dofs)
    if face_elem is None:
        nf, fe_p, fs_p = -1, None, None
    else:
        fe, fs = _faces(face_elem, face_side)
        nf, fe_p, fs_p = len(fe), fe.ctypes.data_as(_i64p), fs.ctypes.data_as(_u8p)
    _chk(lib().orc_values_at_nodes(C.byref(mesh.struct), rid, _d(kpar), C.c_double(time), C.c_int64(nf), fe_p, fs_p,
                                   di.ctypes.data_as(_ip), _d(s), _d(c)))
    return s, c
