"""ctypes binding of the C ABI declared in include/l3k.h (l3ster_amd/lib/libl3k.so).

This is the reference-side binding a maintainer would write; it contains no compute.  The library
must be present: there is no Python / CPU fallback for the device entry points.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("L3K_LIB") or os.path.join(_HERE, "lib", "libl3k.so")

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)
c_uint32_p = C.POINTER(C.c_uint32)
c_uint8_p = C.POINTER(C.c_uint8)


class KParams(C.Structure):
    _fields_ = [("dimension", C.c_int), ("n_equations", C.c_int), ("n_unknowns", C.c_int), ("n_fields", C.c_int),
                ("n_rhs", C.c_int)]


class AsmOpts(C.Structure):
    _fields_ = [("value_order", C.c_int), ("derivative_order", C.c_int), ("eval_strategy", C.c_int)]


class CgOpts(C.Structure):
    _fields_ = [("tol", C.c_double), ("max_iters", C.c_int), ("residual_scaling", C.c_int), ("check_every", C.c_int)]


class CgResult(C.Structure):
    _fields_ = [("achieved_tol", C.c_double), ("iterations", C.c_int), ("converged", C.c_int)]


class MeshDesc(C.Structure):
    _fields_ = [("dim", C.c_int), ("order", C.c_int), ("n_elems", C.c_int64), ("n_interior_elems", C.c_int64),
                ("elem_nodes", c_uint32_p), ("elem_verts", c_double_p), ("n_owned_nodes", C.c_int64),
                ("n_ghost_nodes", C.c_int64), ("dofs_per_node", C.c_int), ("dirichlet", c_uint8_p)]


class HostMeshView(C.Structure):
    _fields_ = [("dim", C.c_int), ("order", C.c_int), ("n_elems", C.c_int64), ("n_interior_elems", C.c_int64),
                ("n_owned_nodes", C.c_int64), ("n_ghost_nodes", C.c_int64), ("global_node_base", C.c_int64),
                ("n_global_nodes", C.c_int64), ("elem_nodes", c_uint32_p), ("elem_verts", c_double_p),
                ("node_grid_id", c_int64_p), ("node_boundary", c_uint8_p), ("n_nbrs", C.c_int), ("nbr_rank", c_int_p),
                ("send_offsets", c_int64_p), ("send_nodes", c_int32_p), ("ghost_offsets", c_int64_p),
                ("elem_boundary", c_uint8_p), ("ghost_global_id", c_int64_p)]


c_uint64_p = C.POINTER(C.c_uint64)
c_uint16_p = C.POINTER(C.c_uint16)


class HaloTransport(C.Structure):
    """l3k_halo_transport: group begin / send / recv / group end (+ destroy) over device pointers and a HIP stream."""
    GROUP_BEGIN = C.CFUNCTYPE(C.c_int, C.c_void_p)
    SEND = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
    RECV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
    GROUP_END = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
    DESTROY = C.CFUNCTYPE(None, C.c_void_p)
    _fields_ = [("user", C.c_void_p), ("group_begin", GROUP_BEGIN), ("send", SEND), ("recv", RECV), ("group_end", GROUP_END),
                ("destroy", DESTROY)]


class MeshFileElems(C.Structure):
    _fields_ = [("n", C.c_size_t), ("nodes", c_uint64_p), ("verts", c_double_p), ("ids", c_uint64_p)]


class MeshFileDomain(C.Structure):
    _fields_ = [("id", C.c_uint16), ("hex", MeshFileElems), ("quad", MeshFileElems), ("line", MeshFileElems)]


class MeshFilePartDesc(C.Structure):
    _fields_ = [("order", C.c_int), ("n_domains", C.c_size_t), ("domains", C.POINTER(MeshFileDomain)),
                ("nodes_begin", C.c_uint64), ("n_owned_nodes", C.c_size_t), ("n_boundary_ids", C.c_size_t),
                ("boundary_ids", c_uint16_p)]


class Tuning(C.Structure):
    """l3k_tuning: the launch-route settings of a context (include/l3k.h)"""
    _fields_ = [("generic_below", C.c_int64), ("static_deal", C.c_int), ("waves_per_cu", C.c_int), ("no_affine", C.c_int),
                ("column_by_column", C.c_int), ("assemble_dense", C.c_int), ("assemble_two_launches", C.c_int),
                ("scatter_per_entry", C.c_int), ("assemble_direct_store", C.c_int), ("assemble_sub_batch", C.c_int), ("assemble_no_symmetrise", C.c_int)]


# every symbol include/l3k.h declares: (name, restype, argtypes)
_vp = C.c_void_p
SIGNATURES = {
    "l3k_version": (C.c_int, []),
    "l3k_last_error": (C.c_char_p, []),
    "l3k_gll_nodes": (C.c_int, [C.c_int, c_double_p]),
    "l3k_gl_rule": (C.c_int, [C.c_int, c_double_p, c_double_p]),
    "l3k_n_qps1d": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "l3k_basis_1d": (C.c_int, [C.c_int, C.c_int, c_double_p, c_double_p]),
    "l3k_colloc_deriv": (C.c_int, [C.c_int, c_double_p]),
    "l3k_kernel_info": (C.c_int, [C.c_int, C.POINTER(KParams), C.POINTER(C.c_char_p), C.POINTER(C.c_size_t)]),
    "l3k_plugin_load": (C.c_int, [C.c_char_p]),
    "l3k_instance_count": (C.c_int, []),
    "l3k_instance_info": (C.c_int, [C.c_int, c_int_p, c_int_p, c_int_p, c_int_p]),
    "l3k_ctx_create": (C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    "l3k_ctx_set_stream": (C.c_int, [_vp, _vp]),
    "l3k_ctx_set_deterministic": (C.c_int, [_vp, C.c_int]),
    "l3k_ctx_set_reference_z0": (C.c_int, [_vp, C.c_int]),
    "l3k_ctx_get_tuning": (C.c_int, [_vp, C.POINTER(Tuning)]),
    "l3k_ctx_set_tuning": (C.c_int, [_vp, C.POINTER(Tuning)]),
    "l3k_mf_route": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "l3k_ctx_synchronize": (C.c_int, [_vp]),
    "l3k_ctx_destroy": (C.c_int, [_vp]),
    "l3k_mesh_create": (C.c_int, [_vp, C.POINTER(MeshDesc), C.POINTER(_vp)]),
    "l3k_mesh_destroy": (C.c_int, [_vp]),
    "l3k_mf_create": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_size_t, C.POINTER(AsmOpts), c_int_p, C.c_int,
                                C.POINTER(_vp)]),
    "l3k_mf_destroy": (C.c_int, [_vp]),
    "l3k_mf_set_fields": (C.c_int, [_vp, _vp, C.c_size_t]),
    "l3k_mf_set_time": (C.c_int, [_vp, C.c_double]),
    "l3k_mf_apply": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_int, C.c_double, C.c_double]),
    "l3k_mf_scale": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, C.c_double]),
    "l3k_mf_apply_elems": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_size_t,
                                     C.c_int, C.c_double, C.c_double]),
    "l3k_mf_dirichlet_rows": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_int, C.c_double]),
    "l3k_pack_rows": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp, C.c_int64, _vp]),
    "l3k_unpack_add_rows": (C.c_int, [_vp, _vp, C.c_int64, _vp, _vp, C.c_size_t, C.c_int]),
    "l3k_mf_diag_rhs": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t, _vp, _vp, C.c_size_t, _vp, _vp, C.c_size_t, C.c_int]),
    "l3k_mf_dirichlet_finalize": (C.c_int, [_vp, _vp, C.c_size_t, _vp, _vp, C.c_size_t]),
    "l3k_local_assemble": (C.c_int, [_vp, C.c_int64, C.c_int64, _vp, _vp, _vp]),
    "l3k_halo_unique_id": (C.c_int, [C.c_char_p]),
    "l3k_halo_create": (C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, c_int_p, c_int64_p, c_int32_p, c_int64_p,
                                  C.POINTER(_vp)]),
    "l3k_halo_create_transport": (C.c_int, [_vp, C.POINTER(HaloTransport), C.c_int, C.c_int, C.c_int, C.c_int, c_int_p, c_int64_p,
                                            c_int32_p, c_int64_p, C.POINTER(_vp)]),
    "l3k_inproc_group_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "l3k_inproc_group_destroy": (C.c_int, [_vp]),
    "l3k_inproc_transport": (C.c_int, [_vp, C.c_int, C.POINTER(HaloTransport)]),
    "l3k_halo_destroy": (C.c_int, [_vp]),
    "l3k_halo_n_ghost_dofs": (C.c_int64, [_vp]),
    "l3k_halo_import": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp, C.c_size_t]),
    "l3k_halo_export_add": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp, C.c_size_t]),
    "l3k_halo_timing_begin": (C.c_int, [_vp, C.c_int]),
    "l3k_halo_timing_get": (C.c_int, [_vp, C.c_int, c_double_p]),
    "l3k_mf_apply_dist": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_int, C.c_double, C.c_double]),
    "l3k_assembled_scatter": (C.c_int, [_vp, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, C.c_int,
                                        c_int64_p]),
    "l3k_local_assemble_tiled": (C.c_int, [_vp, C.c_int64, C.c_int64, _vp]),
    "l3k_assemble_global": (C.c_int, [_vp, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, C.c_size_t, C.c_int, C.c_size_t, c_int64_p]),
    "l3k_bnd_create": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_size_t, C.POINTER(AsmOpts), c_int_p, C.c_int, C.c_int64,
                                 c_int64_p, c_uint8_p, C.POINTER(_vp)]),
    "l3k_bnd_destroy": (C.c_int, [_vp]),
    "l3k_bnd_set_fields": (C.c_int, [_vp, _vp, C.c_size_t]),
    "l3k_bnd_set_time": (C.c_int, [_vp, C.c_double]),
    "l3k_bnd_apply": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_size_t, C.c_int,
                                C.c_double]),
    "l3k_bnd_diag_rhs": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t, _vp, _vp, C.c_size_t, _vp, _vp, C.c_size_t]),
    "l3k_mf_attach_boundary": (C.c_int, [_vp, _vp]),
    "l3k_residual_info": (C.c_int, [C.c_int, C.POINTER(KParams), C.POINTER(C.c_char_p), C.POINTER(C.c_size_t)]),
    "l3k_integrate": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_size_t, C.POINTER(AsmOpts), _vp, C.c_size_t, C.c_double,
                                C.c_int, C.c_int64, c_int64_p, c_uint8_p, c_double_p]),
    "l3k_values_at_nodes": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, C.c_double, C.c_int64, c_int64_p,
                                      c_uint8_p, c_int_p, _vp, _vp]),
    "l3k_average_values": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp]),
    "l3k_update_solution": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_int, C.c_int, c_int_p, c_int_p, _vp, C.c_size_t,
                                      C.c_int]),
    "l3k_jacobi_inverse": (C.c_int, [_vp, _vp, C.c_int64, C.c_double, C.c_double, _vp]),
    "l3k_pcg_solve": (C.c_int, [_vp, _vp, _vp, _vp, C.POINTER(CgOpts), C.POINTER(CgResult)]),
    "l3k_pcg_solve_cols": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_int, _vp, C.POINTER(CgOpts), C.POINTER(CgResult)]),
    "l3k_cg_init": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int64, _vp]),
    "l3k_cg_dot_pap": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp]),
    "l3k_mf_apply_energy": (C.c_int, [_vp, _vp, _vp, _vp]),
    "l3k_mf_energy_begin": (C.c_int, [_vp, _vp]),
    "l3k_mf_energy_end": (C.c_int, [_vp, _vp, c_int_p]),
    "l3k_cg_update_z": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, _vp]),
    "l3k_cg_update_px": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, _vp]),
    "l3k_cube_partition_create": (C.c_int, [c_int_p, C.c_int, c_int_p, C.c_int, C.c_double, C.POINTER(_vp)]),
    "l3k_hostmesh_destroy": (C.c_int, [_vp]),
    "l3k_hostmesh_view_get": (C.c_int, [_vp, C.POINTER(HostMeshView)]),
    "l3k_elevate_order": (C.c_int, [_vp, C.c_int64, c_uint32_p, C.c_int64, C.c_int, c_uint32_p, c_int64_p, c_int64_p]),
    "l3k_results_save": (C.c_int, [C.c_char_p, C.c_char_p, C.c_size_t, C.c_int64, C.c_int64, C.c_int64, c_double_p, C.c_size_t,
                                   C.c_int]),
    "l3k_results_info": (C.c_int, [C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "l3k_results_load": (C.c_int, [C.c_char_p, C.c_size_t, C.c_int64, c_int64_p, C.c_int64, c_double_p]),
    "l3k_meshfile_part_bytes": (C.c_int, [C.POINTER(MeshFilePartDesc), C.POINTER(C.c_size_t)]),
    "l3k_meshfile_save": (C.c_int, [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_size_t,
                                    C.POINTER(MeshFilePartDesc), C.c_int]),
    "l3k_meshfile_info": (C.c_int, [C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_size_t]),
    "l3k_meshfile_load": (C.c_int, [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(_vp)]),
    "l3k_meshfile_part_get": (C.c_int, [_vp, C.POINTER(MeshFilePartDesc)]),
    "l3k_meshfile_part_destroy": (C.c_int, [_vp]),
}

_lib = None


class L3KError(RuntimeError):
    pass


def load():
    """Loads libl3k.so; raises if it has not been built (python -m l3ster_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise L3KError(f"{LIB_PATH} is missing: build the HIP extension first (python -m l3ster_amd.build). "
                       "There is no CPU fallback for the device path.")
    # torch ships its own HIP runtime (same soname as /opt/rocm's, which libl3k.so names in its RUNPATH): whichever is
    # mapped first serves the whole process, and device memory / streams come from torch here -- so torch goes first
    try:
        import torch  # noqa: F401
    except ImportError:  # pure C-ABI use without torch: the system runtime
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise L3KError(f"libl3k error {rc}: {load().l3k_last_error().decode()}")
