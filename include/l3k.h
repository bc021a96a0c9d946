/* l3k.h -- C ABI of the MI355X-native element-local hot path (libl3k.so).
 *
 * Drop-in boundary for the L3STER hot path.  The reference is header-only C++ with no FFI; each entry
 * point below names the reference interface (file:line under /root/reference/include/l3ster) it stands in for.  Plain
 * pointers and sizes only; never throws across the boundary: every function returns 0 on success, < 0 on error, and
 * l3k_last_error() gives the text (the reference throws std::runtime_error from util::throwingAssert,
 * util/Assertion.hpp:88-95).
 *
 * Pointer conventions: arguments named d_* are DEVICE pointers (HBM of the ctx's GPU); everything else is host memory.
 * Vectors are column-major [row][col] with an explicit leading dimension, owned rows only, exactly like the host view
 * of the Tpetra multivectors the reference's Operator::apply works on (algsys/ComputeValuesAtNodes.hpp:27-31,62-65).
 * Calls on one l3k_ctx are serialised by the caller (the reference's apply is not re-entrant either: shared
 * import/export buffers, algsys/MatrixFreeSystem.hpp:1008-1009).  All device work is enqueued on the ctx's HIP stream
 * and is asynchronous with respect to the host unless stated otherwise.
 */
#ifndef L3K_H
#define L3K_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 101 (round 4): l3k_cg_update_xr / _update_p replaced by l3k_cg_update_z / _update_px (the iteration keeps z = M^-1 r; d_r holds z),
 * new: l3k_ctx_set_reference_z0, l3k_ctx_get/set_tuning, l3k_mf_route, l3k_update_solution, l3k_pcg_solve_cols */
#define L3K_VERSION 101

typedef struct l3k_ctx      l3k_ctx;
typedef struct l3k_mesh     l3k_mesh;
typedef struct l3k_mf       l3k_mf;
typedef struct l3k_hostmesh l3k_hostmesh;

int         l3k_version(void);
const char* l3k_last_error(void);

/* ---- compile-time structs of the reference, as runtime descriptors ---------------------------------------------- */
/* KernelParams, common/KernelInterface.hpp:13-20 */
typedef struct
{
    int dimension, n_equations, n_unknowns, n_fields, n_rhs;
} l3k_kparams;
/* AssemblyOptions, algsys/AssembleLocalSystem.hpp:24-49.  eval_strategy: 0 Auto, 1 LocalElement, 2 SumFactorization,
 * 3 SumFactorizationOddEvenDecomposition (LocalEvalStrategy, :16-22).  On the device every strategy runs the same
 * sum-factorised kernel (collocation-derivative form, mathematically identical). */
typedef struct
{
    int value_order, derivative_order, eval_strategy;
} l3k_asmopts;

/* ---- tables (host; no GPU needed) ------------------------------------------------------------------------------- */
/* math::getLobattoRuleAbsc, math/LobattoRuleAbsc.hpp:30-35 */
int l3k_gll_nodes(int n, double* x);
/* quad::getReferenceQuadrature<GaussLegendre>, quad/ReferenceQuadrature.hpp:24-51 (nq = QO/2+1 points) */
int l3k_gl_rule(int nq, double* x, double* w);
/* AssemblyOptions::order + getRefQuadSize: nq1d = value_order*p + derivative_order*(p-1) + 1 */
int l3k_n_qps1d(int p, int value_order, int derivative_order);
/* interpolation_matrix / derivative_matrix, algsys/SumFactorization.hpp:25-65: row-major [p+1][nq] */
int l3k_basis_1d(int p, int nq, double* I, double* D);
/* collocation derivative matrix on the nq Gauss points: C[q'][q] = l_q'^GL '(x_q), row-major [nq][nq] (device algorithm
 * table; D = I * C for nq >= p+1) */
int l3k_colloc_deriv(int nq, double* C);

/* ---- kernel registry ------------------------------------------------------------------------------------------------
 * The reference takes the operator definition as a C++ callable wrapped by wrapDomainEquationKernel<params>
 * (common/KernelInterface.hpp:178-183) and instantiates the element loops on its type.  Here kernels are
 * __host__ __device__ functors with the same (in, out) signature (include/l3k/kernel_interface.hpp), registered in
 * l3ster_amd/csrc/user_kernels.hpp and instantiated into the HIP templates at build time; they are selected by id and
 * carry an optional POD parameter block. */
enum
{
    L3K_KERNEL_DIFFUSION3D     = 0, /* benchmarks/Diffusion3D.hpp:51-79; params {double k, s}                        */
    L3K_KERNEL_DIFFUSION3D_VAR = 1, /* tests/Kernels.hpp:84-118, n_fields = 1                                          */
    L3K_KERNEL_ADVDIFF3D       = 4, /* config-5 synthetic; params {double k, sigma, s}, n_fields = 3 */
    /* boundary equation kernels (wrapBoundaryEquationKernel: the input carries the outward normal); l3k_bnd_create */
    L3K_KERNEL_ADIABATIC3D     = 6, /* 3-D twin of tests/Kernels.hpp:120-128: q.n = 0                                 */
    L3K_KERNEL_ROBIN3D         = 7  /* synthetic: q.n + h T = h T_inf; params {double h, t_inf}                      */
};
/* residual kernels (wrapDomainResidualKernel / wrapBoundaryResidualKernel) for l3k_integrate */
enum
{
    L3K_RESIDUAL_DIFFUSION3D_ERROR = 0, /* benchmarks/Diffusion3D.hpp:81-103; fields (T,qx,qy,qz); params {double k, s} */
    L3K_RESIDUAL_LINEAR3D_ERROR    = 2, /* 3-D twin of tests/Diffusion2D.hpp:84-92: error against T = x, q = (1,0,0)   */
    L3K_RESIDUAL_UNIT3D            = 4, /* tests/MappingTests.cpp:567-569: integrand 1                                */
    L3K_RESIDUAL_COORDX3D          = 6  /* 3-D twin of tests/Diffusion2D.hpp:49-50: out[0] = x (Dirichlet value kernel) */
};
int l3k_kernel_info(int kernel_id, l3k_kparams* params, const char** name, size_t* param_bytes);
/* Kernels that are not compiled into libl3k.so: a kernel PLUGIN is a shared library built from the user's functor (the
 * reference compiles the user's lambda with the application; here `l3ster_amd.plugin.compile_kernel` / the Makefile
 * fragment in INTEGRATION.md run hipcc on a generated translation unit that instantiates the element kernels for the
 * functor and the requested (order, nq, ncols) shapes).  Loading it registers the kernel id and its instantiations;
 * ids >= 1000 are free for plugins. */
int l3k_plugin_load(const char* path);
/* number of (kernel, order, nq, ncols) device instantiations, and the i-th one: for "is this shape built?" queries */
int l3k_instance_count(void);
int l3k_instance_info(int i, int* kernel_id, int* order, int* nq, int* ncols);

/* ---- context ----------------------------------------------------------------------------------------------------- */
/* One ctx per GPU / per process.  hip_stream is a hipStream_t (NULL = default stream); torch users pass
 * torch.cuda.current_stream().cuda_stream. */
int l3k_ctx_create(int hip_device, void* hip_stream, l3k_ctx** out);
int l3k_ctx_set_stream(l3k_ctx* ctx, void* hip_stream);
/* Bitwise-reproducible mode (also: environment L3K_DETERMINISTIC=1 when the context is created).  The reference scatters
 * with relaxed atomics (algsys/MatrixFreeSystem.hpp:513), so its global results change in the last bits from run to run
 * and PCG iteration counts move with them; for parity runs this build can fix the order instead: meshes created while the
 * mode is on carry a colouring of their elements (no two elements of a colour share a node) and every domain-kernel
 * launch (apply, diag / rhs) goes colour by colour, so each row receives its contributions in a fixed order; <p, A p>
 * comes from the fixed-order dot product.  Boundary terms created while the mode is on colour their element sides the
 * same way and launch per colour behind the domain kernel.  Slower (one launch per colour). */
int l3k_ctx_set_deterministic(l3k_ctx* ctx, int on);
/* Which point a DOMAIN kernel sees in the matrix-free apply.  The reference is not consistent with itself here: its hex
 * sum-factorisation path builds SpaceTimePoint{Point<3>{x, y, 0.}, time} (algsys/SumFactorization.hpp:732, a copy of the 2-D
 * line :656), its local-element path -- assembleLocalSystem, evaluateLocalOperator, precomputeOperatorDiagonalAndRhs -- the
 * true point (algsys/AssembleLocalSystem.hpp:229-230).  Default (on = 0): the true point everywhere, i.e. apply, diag / rhs
 * and LocalAssembly describe ONE operator.  on = 1: applies launched from this context afterwards pass z = 0 like the
 * reference's evalAtHexQPs (bit-for-bit reference behaviour for a kernel that reads point.space.z() under
 * sum-factorisation); diag / rhs and LocalAssembly keep the true point, as in the reference.  Kernels that do not read z
 * are unaffected. */
int l3k_ctx_set_reference_z0(l3k_ctx* ctx, int on);
/* Launch-route settings of a context.  The defaults are what the measurements recorded in DESIGN.md chose; the fields exist so
 * that tests and tools can take the other route ON PURPOSE.  The environment is consulted once, in l3k_ctx_create
 * (L3K_GENERIC_BELOW, L3K_FAST_STATIC, L3K_FAST_WAVES_PER_CU, L3K_NO_AFFINE, L3K_COLUMN_BY_COLUMN, L3K_ASSEMBLE_DENSE,
 * L3K_ASM_TWO_LAUNCHES, L3K_SCATTER_PER_ENTRY, L3K_ASM_DIRECT_STORE, L3K_ASM_NO_SYMMETRISE initialise the fields below, L3K_DETERMINISTIC the deterministic mode): nothing on the
 * launch path reads the environment, and l3k_mf_route names the kernel a launch takes. */
typedef struct
{
    int64_t generic_below;         /* element launches of fewer elements take the generic LDS kernel (latency-bound sizes); 1500 */
    int     static_deal;           /* single-wave kernel: static deal of the element batches instead of the dynamic one; 0      */
    int     waves_per_cu;          /* single-wave kernel: persistent waves per CU, 0 = as many as LDS and registers admit; 0    */
    int     no_affine;             /* never take the affine variant (one Jacobian per element) on all-affine meshes; 0          */
    int     column_by_column;      /* multi-column applies as one launch per column (cross-check of the one-pass variant); 0    */
    int     assemble_dense;        /* LocalAssembly as the dense FP64-MFMA product instead of the sum-factorised kernels; 0     */
    int     assemble_two_launches; /* stored row-major LocalAssembly as two launches (diagonal / off-diagonal blocks); 0        */
    int     scatter_per_entry;     /* l3k_assembled_scatter: one wave per row with a search per entry (round-2 kernel); 0       */
    int     assemble_direct_store; /* stored row-major LocalAssembly written by the assembly kernel itself (8-byte stores at a
                                      32-byte stride, 3.9 x write traffic) instead of x-major tiled layout + mirroring
                                      transposition kernel; 0                                                                   */
    int     assemble_sub_batch;    /* stored row-major LocalAssembly: elements per pipelined sub-batch, 0 = chosen by size; 0      */
    int     assemble_no_symmetrise; /* stored row-major LocalAssembly through the plain tiled layout and the plain transposition
                                      (both triangles formed, each in its own summation order): K_e symmetric to rounding
                                      (1e-13) instead of bit for bit like the reference's -- the cross-check of the default; 0  */
} l3k_tuning;
int l3k_ctx_get_tuning(const l3k_ctx* ctx, l3k_tuning* out);
int l3k_ctx_set_tuning(l3k_ctx* ctx, const l3k_tuning* in);
int l3k_ctx_synchronize(l3k_ctx* ctx);
int l3k_ctx_destroy(l3k_ctx* ctx);

/* ---- device mesh --------------------------------------------------------------------------------------------------
 * What the reference keeps in LocalMeshView / LocalElementView (mesh/LocalMeshView.hpp:13-145), LocalDofMap
 * (dofs/NodeToDofMap.hpp:84-109) and LocalDirichletBC (bcs/LocalDirichletBC.hpp:13-32), flattened.  Uploaded once. */
typedef struct
{
    int             dim;            /* 3 (hex)                                                                      */
    int             order;          /* p; nodes per element N = (p+1)^dim, lexicographic, xi fastest               */
    int64_t         n_elems;        /* elements [0, n_interior_elems) touch owned dofs only, the rest are "border" */
    int64_t         n_interior_elems; /* splitBorderAndInterior, algsys/MatrixFreeSystem.hpp:969-981              */
    const uint32_t* elem_nodes;     /* [n_elems][N] local node ids (n_loc_id_t, common/Typedefs.h:14)              */
    const double*   elem_verts;     /* [n_elems][2^dim][3], vertex v = i + 2j + 4k (mesh/primitives/CubeMesh.hpp)  */
    int64_t         n_owned_nodes;  /* local node numbering: owned first, then ghosts (LocalMeshView.hpp:425-458)  */
    int64_t         n_ghost_nodes;
    int             dofs_per_node;  /* dof(node,k) = node*dofs_per_node + k (dofs/NodeToDofMap.hpp:250-264)        */
    const uint8_t*  dirichlet;      /* [(n_owned+n_ghost)*dofs_per_node] byte mask or NULL (isDirichletDof)       */
} l3k_mesh_desc;
int l3k_mesh_create(l3k_ctx* ctx, const l3k_mesh_desc* desc, l3k_mesh** out);
int l3k_mesh_destroy(l3k_mesh* mesh);

/* ---- matrix-free operator ------------------------------------------------------------------------------------------
 * Stands in for algsys::MatrixFreeSystem (assembleProblem + Operator::apply), algsys/MatrixFreeSystem.hpp:24-89.
 * field_inds[n_unknowns]: which per-node dof each unknown maps to (detail::getDofs, :298-311). */
int l3k_mf_create(l3k_ctx* ctx, l3k_mesh* mesh, int kernel_id, const void* kparam_blob, size_t kparam_bytes,
                  const l3k_asmopts* opts, const int* field_inds, int n_rhs, l3k_mf** out);
int l3k_mf_destroy(l3k_mf* mf);
/* external fields read by the kernel: SoA [n_fields][ld] over local nodes, value(node, f) = d_soa[node + f*ld]
 * (post::FieldAccess, post/FieldAccess.hpp:21-30).  The pointer is kept, not copied. */
int l3k_mf_set_fields(l3k_mf* mf, const double* d_soa, size_t ld);
int l3k_mf_set_time(l3k_mf* mf, double time);
/* Which kernel would l3k_mf_apply_elems(mf, which, ..., ncols, ...) launch for its elements right now?  Writes a one-line
 * description into buf (NUL-terminated, truncated to n bytes): kernel template and variant, lanes, LDS, waves per CU, grid --
 * decided by the same code as the launch itself.  For run-time reports (bench.py prints it) and for tests that assert a route.
 * with_energy != 0: as inside l3k_mf_apply_energy / between l3k_mf_energy_begin and _end. */
int l3k_mf_route(l3k_mf* mf, int which, int ncols, int with_energy, char* buf, size_t n);

/* Y <- alpha*A*X + beta*Y.  Operator::apply / applyImpl, algsys/MatrixFreeSystem.hpp:34-41,1020-1140, for a rank
 * without ghosts (n_ghost_nodes == 0): scale (:1038), gather with Dirichlet -> 0 (:421-467), sum-factorised element
 * kernel (algsys/SumFactorization.hpp:882-917), scatter-add skipping Dirichlet dofs (:494-537), y[d] += alpha*x[d] on
 * owned Dirichlet rows (:1087-1098).  ncols <= n_rhs, else error (:1035-1037). */
/* Alignment: when the kernel's unknowns are all dofs of a node (the dense layout), x, y and the ghost buffers must be
 * 16-byte aligned and leading dimensions even when ncols > 1 (a node's dofs move with 16-byte accesses); align node rows
 * to 32 bytes for full speed. */
int l3k_mf_apply(l3k_mf* mf, const double* d_x, size_t ldx, double* d_y, size_t ldy, int ncols, double alpha,
                 double beta);

/* Split-phase form for ranks with ghosts (the pieces applyImpl interleaves with comm::Import/Export,
 * algsys/MatrixFreeSystem.hpp:1046-1111).  Ghost values live in separate [n_ghost_dofs][ncols] buffers exactly like
 * m_import_shared_buf / m_export_shared_buf (:1008-1009), accessed through BorderAccessor semantics
 * (algsys/ComputeValuesAtNodes.hpp:21-50): local dof < n_owned -> owned vector, else ghost buffer.
 *   which: 0 = interior elements, 1 = border elements, 2 = all; 3 / 4 = first / second half of the interior elements
 *   (the schedule import || first half, border, export || second half hides both exchanges; the reference overlaps
 *   only the import, :1046-1086). */
/* Y <- beta*Y (:1038) on the rows the element kernels ACCUMULATE into.  Rows of nodes that belong to exactly one element
 * (the element-internal nodes of the reference's numbering, mesh/LocalMeshView.hpp:425-458) are not touched here when
 * the operator covers all dofs of a node: l3k_mf_apply_elems WRITES alpha*A*x + beta*y there (no atomics, and for
 * beta = 0 no read), so it must be given the same beta.  Together the two calls equal the reference's scale + scatter. */
int l3k_mf_scale(l3k_mf* mf, double* d_y, size_t ldy, int ncols, double beta);
int l3k_mf_apply_elems(l3k_mf* mf, int which, const double* d_x, size_t ldx, const double* d_xghost, size_t ldxg,
                       double* d_y, size_t ldy, double* d_yghost, size_t ldyg, int ncols, double alpha, double beta);
int l3k_mf_dirichlet_rows(l3k_mf* mf, const double* d_x, size_t ldx, double* d_y, size_t ldy, int ncols,
                          double alpha);                                                        /* :1087-1098      */
/* comm::Import pack (owner side gathers owned rows through m_owned_inds, comm/ImportExport.hpp:356-372) and
 * comm::Export unpack-add (util::AtomicSumInto through m_owned_inds, :448-470), on DOF rows:
 *   pack:       d_dst[i + n*c]  = d_src[idx[i] + ld*c]
 *   unpack_add: d_dst[idx[i] + ld*c] += d_src[i + n*c]      (idx must not repeat within one call)                  */
int l3k_pack_rows(l3k_ctx* ctx, const double* d_src, size_t ld, int ncols, const int32_t* d_idx, int64_t n,
                  double* d_dst);
int l3k_unpack_add_rows(l3k_ctx* ctx, const double* d_src, int64_t n, const int32_t* d_idx, double* d_dst, size_t ld,
                        int ncols);

/* diag(A) and rhs with Dirichlet lifting: computeDiagAndRhs, algsys/MatrixFreeSystem.hpp:888-941 around
 * precomputeOperatorDiagonalAndRhs (algsys/EvaluateLocalOperator.hpp:172-208,276-301).  d_dirichlet_vals
 * [n_local_dofs][n_rhs] (ld) or NULL (= 0); outputs accumulate into owned rows d_diag[n_owned_dofs],
 * d_rhs[n_owned_dofs][n_rhs] and ghost rows d_diag_ghost / d_rhs_ghost (may be NULL when n_ghost_nodes == 0);
 * the caller zeroes them first (:921-923).  finalize != 0 sets diag = 1, rhs = g on owned Dirichlet rows (:911-915). */
int l3k_mf_diag_rhs(l3k_mf* mf, int which, const double* d_dirichlet_vals, size_t ldg, double* d_diag, double* d_rhs,
                    size_t ldr, double* d_diag_ghost, double* d_rhs_ghost, size_t ldrg, int finalize);
/* the finalisation alone (diag = 1, rhs = g on owned Dirichlet rows, :911-915): partitioned systems call l3k_mf_diag_rhs
 * with finalize = 0, export-add the ghost rows to their owners (:925-938), then this */
int l3k_mf_dirichlet_finalize(l3k_mf* mf, const double* d_dirichlet_vals, size_t ldg, double* d_diag, double* d_rhs, size_t ldr);

/* ---- boundary equation kernels on element sides ---------------------------------------------------------------------
 * assembleProblem(kernel, boundary_ids) with a BoundaryEquationKernel (algsys/MatrixFreeSystem.hpp:58-68): the term
 * sum over the listed element sides of  int_side B^T B  with the side quadrature, surface jacobian and outward normal
 * of map::mapBoundary (algsys/EvaluateLocalOperator.hpp:238-274,303-330; mapping/BoundaryNormal.hpp:8-64;
 * mapping/BoundaryIntegralJacobian.hpp:9-29; basisfun/ReferenceElementBasisAtQuadrature.hpp:57-97).
 * A side is (element index in the mesh, side): hex sides 0..5 = z-, z+, y-, y+, x-, x+ (mesh/ElementTraits.hpp:84-95).
 * l3k_mf_attach_boundary registers the term with a system: l3k_mf_apply / l3k_mf_apply_elems / l3k_mf_diag_rhs then
 * include it, like the reference evaluates every kernel passed to assembleProblem.  `which` as in l3k_mf_apply_elems:
 * sides of interior elements (0), of border elements (1), all (2).  The term does not own the system or the mesh and
 * must outlive the systems it is attached to. */
typedef struct l3k_bnd l3k_bnd;
int l3k_bnd_create(l3k_ctx* ctx, l3k_mesh* mesh, int kernel_id, const void* kparam_blob, size_t kparam_bytes,
                   const l3k_asmopts* opts, const int* field_inds, int n_rhs, int64_t n_faces, const int64_t* face_elem,
                   const uint8_t* face_side, l3k_bnd** out);
int l3k_bnd_destroy(l3k_bnd* bnd);
int l3k_bnd_set_fields(l3k_bnd* bnd, const double* d_soa, size_t ld);
int l3k_bnd_set_time(l3k_bnd* bnd, double time);
/* y += alpha * A_b x  (Dirichlet columns read as 0, Dirichlet rows skipped: same semantics as the element kernels) */
int l3k_bnd_apply(l3k_bnd* bnd, int which, const double* d_x, size_t ldx, const double* d_xghost, size_t ldxg, double* d_y,
                  size_t ldy, double* d_yghost, size_t ldyg, int ncols, double alpha);
/* diag += diag(A_b) (d_diag may be NULL), rhs += B_b^T W (f_b - B_b g) */
int l3k_bnd_diag_rhs(l3k_bnd* bnd, int which, const double* d_dirichlet_vals, size_t ldg, double* d_diag, double* d_rhs,
                     size_t ldr, double* d_diag_ghost, double* d_rhs_ghost, size_t ldrg);
int l3k_mf_attach_boundary(l3k_mf* mf, l3k_bnd* bnd);

/* ---- integrals of residual kernels (post-processing) -----------------------------------------------------------------
 * evalLocalIntegral, post/Integral.hpp:54-111: h_out[n_equations] (HOST) = sum over all elements (n_faces < 0) or over
 * the listed element sides of the integral of the residual kernel evaluated on the fields d_fields (SoA [F][ld], local
 * node index); square != 0 integrates the squared components.  computeNormL2 (post/NormL2.hpp:31-62) is
 * sqrt(all-reduce(l3k_integrate(opts with value_order and derivative_order doubled, square = 1))).  The quadrature
 * size follows from opts as for the operator kernels.  Results are bitwise reproducible (fixed summation order). */
int l3k_residual_info(int residual_id, l3k_kparams* params, const char** name, size_t* param_bytes);
int l3k_integrate(l3k_ctx* ctx, l3k_mesh* mesh, int residual_id, const void* kparam_blob, size_t kparam_bytes,
                  const l3k_asmopts* opts, const double* d_fields, size_t ldf, double time, int square, int64_t n_faces,
                  const int64_t* face_elem, const uint8_t* face_side, double* h_out);

/* ---- values of residual kernels at the nodes (Dirichlet values, initial conditions) ---------------------------------------
 * computeValuesAtNodes (algsys/ComputeValuesAtNodes.hpp:371-448 domain, :508-594 boundary), the engine behind
 * setDirichletBCValues / setValues: the kernel is evaluated at the nodes of the listed element sides (n_faces >= 0) or of
 * every element (n_faces < 0) with the nodal field values, their physical derivatives, the point and (sides) the outward
 * normal; equation e is ADDED to d_sum[node * dofs_per_node + dof_inds[e]] and 1 to d_count[...] (both over all local
 * dofs, owned then ghost; the caller zeroes them, and in a partitioned run exports the ghost rows of both to their owners
 * before averaging).  l3k_average_values: values[i] = sum[i] / count[i] where count[i] > 0, other entries untouched
 * (averageElementContributions :112-154). */
int l3k_values_at_nodes(l3k_ctx* ctx, l3k_mesh* mesh, int residual_id, const void* kparam_blob, size_t kparam_bytes,
                        const double* d_fields, size_t ldf, double time, int64_t n_faces, const int64_t* face_elem,
                        const uint8_t* face_side, const int* dof_inds, double* d_sum, double* d_count);
int l3k_average_values(l3k_ctx* ctx, const double* d_sum, const double* d_count, int64_t n, double* d_values);
/* MatrixFreeSystem::updateSolution(sol_inds, sol_man, sol_man_inds) (algsys/MatrixFreeSystem.hpp:1231-1273; AlgebraicSystem's
 * twin): the solution's per-node dofs sol_inds[i] of column r copied into field sol_man_inds[i * ncols + r] of the SoA field
 * storage a kernel's FieldAccess reads (value(node, f) = d_fields[node + f * ldf], the layout of l3k_mf_set_fields), for EVERY
 * local node: owned rows from d_x, ghost rows from d_xghost -- the values the reference imports inside updateSolution; a
 * partitioned host calls l3k_halo_import first (NULL on a rank without ghosts).  Index lists are host arrays.  Errors as the
 * reference's asserts: "Source index out of bounds" (>= dofs_per_node), "Destination index out of bounds" (>= n_fields). */
int l3k_update_solution(l3k_ctx* ctx, l3k_mesh* mesh, const double* d_x, size_t ldx, const double* d_xghost, size_t ldxg, int ncols,
                        int n_inds, const int* sol_inds, const int* sol_man_inds, double* d_fields, size_t ldf, int n_fields);

/* ---- Jacobi-preconditioned conjugate gradients ------------------------------------------------------------------------
 * The reference hands the iteration to Trilinos Belos ("Block CG", solve/BelosSolvers.hpp:116-122) with its native Jacobi
 * preconditioner (solve/NativePreconditioners.hpp:36-96); Belos is not part of the reference tree, the arithmetic here
 * is the textbook Hestenes-Stiefel PCG for one column, pinned end to end.  Everything stays on the
 * context's stream; dot products are two-stage reductions in a fixed order.
 *   l3k_jacobi_inverse : minv = sign(d) * damping / max(|d|, threshold)          (NativeJacobiImpl::init :75-96)
 *   l3k_pcg_solve      : single rank (no ghost nodes); x holds the initial guess and the result; d_minv may be NULL;
 *                        residual_scaling 0 none / 1 initial residual / 2 norm of b (IterSolverOpts,
 *                        solve/SolverInterface.hpp:26-37); check_every = iterations between convergence checks (each is
 *                        one 32-byte device-to-host copy)
 *                        Rows with minv == 0 (a preconditioner zeroed on constrained dofs; damping 0) are frozen: x keeps
 *                        its initial value there and the row is left out of the residual norm (the iteration keeps
 *                        z = M^-1 r, from which r cannot be recovered where minv = 0)
 *   l3k_cg_*           : the fused vector kernels of one iteration for partitioned vectors; the caller all-reduces the
 *                        device scalar block s[8] (0 <r,z>, 1 <p,Ap>, 2 <r,z> new, 3 <r,r>) between them.  Protocol below
 *                        (l3k_cg_init, _dot_pap, _update_z, _update_px): init and update_z write LOCAL sums into s[2], s[3];
 *                        all-reduce them before the next call; after init also copy s[2] to s[0] once the reduced value is
 *                        in place. */
typedef struct
{
    double tol;
    int    max_iters;
    int    residual_scaling;
    int    check_every;
} l3k_cg_opts;
typedef struct
{
    double achieved_tol;
    int    iterations;
    int    converged;
} l3k_cg_result;
int l3k_jacobi_inverse(l3k_ctx* ctx, const double* d_diag, int64_t n, double damping, double threshold, double* d_minv);
int l3k_pcg_solve(l3k_mf* mf, const double* d_b, double* d_x, const double* d_minv, const l3k_cg_opts* opts,
                  l3k_cg_result* result);
/* ... for the ncols columns of a multivector (column c at + c * ld), one after the other as Belos "Block CG" with block size 1
 * does for the reference's n_rhs right-hand sides; results[ncols] */
int l3k_pcg_solve_cols(l3k_mf* mf, const double* d_b, size_t ldb, double* d_x, size_t ldx, int ncols, const double* d_minv,
                       const l3k_cg_opts* opts, l3k_cg_result* results);
/* The pieces of the iteration for hosts that reduce the scalars across ranks themselves (d_s: device block, 0 <r,z> old, 1 <p,Ap>,
 * 2 <r,z> new, 3 <r,r>).  The iteration keeps the preconditioned residual z = M^-1 r instead of r (9 instead of 11 vector passes):
 *   l3k_cg_init:      d_z holds A x0 on entry; z = minv (b - A x0), p = z, s[2] = <r,z>, s[3] = <r,r> (this rank's share)
 *   l3k_cg_update_z:  alpha = s[0]/s[1]; z -= alpha minv Ap; s[2], s[3] as above (r = z / minv)
 *   l3k_cg_update_px: x += alpha p; beta = s[2]/s[0]; p = z + beta p; then s[0] <- s[2]
 * d_minv may be NULL (no preconditioner: z = r). */
int l3k_cg_init(l3k_ctx* ctx, double* d_z, const double* d_b, double* d_p, const double* d_minv, int64_t n, double* d_s);
int l3k_cg_dot_pap(l3k_ctx* ctx, const double* d_p, const double* d_ap, int64_t n, double* d_s);
/* One rank, one column: y <- A x and s[1] <- <x, A x> in one pass (what the PCG needs of an apply followed by
 * l3k_cg_dot_pap).  On the single-wave route of domain kernels the element kernel accumulates x^T A x = sum_q w detJ |B_q x|^2
 * at the quadrature points (the Dirichlet rows, identity rows of the operator, add x_d^2): no pass over x and y afterwards.
 * Otherwise (small meshes, attached boundary terms, other dof layouts) it is the apply followed by the dot product.     */
int l3k_mf_apply_energy(l3k_mf* mf, const double* d_x, double* d_y, double* d_s);
/* The same for the split-phase (partitioned) apply: l3k_mf_energy_begin zeroes s[1] and arms the accumulation for the
 * l3k_mf_apply_elems calls that follow; l3k_mf_energy_end adds the owned Dirichlet rows' share, disarms, and reports in
 * *fused whether every element launch in between accumulated (if not, s[1] is incomplete: take l3k_cg_dot_pap instead).
 * s[1] then holds this rank's share of <x, A x>; the all-reduce over the ranks is the caller's, as for the other scalars. */
int l3k_mf_energy_begin(l3k_mf* mf, double* d_s);
int l3k_mf_energy_end(l3k_mf* mf, const double* d_x, int* fused);
int l3k_cg_update_z(l3k_ctx* ctx, double* d_z, const double* d_ap, const double* d_minv, int64_t n, double* d_s);
int l3k_cg_update_px(l3k_ctx* ctx, double* d_p, double* d_x, const double* d_z, int64_t n, double* d_s);

/* ---- LocalAssembly --------------------------------------------------------------------------------------------------
 * assembleLocalSystem for a batch of elements, algsys/AssembleLocalSystem.hpp:234-256: K_e row-major [Nd][Nd],
 * F_e column-major [Nd][n_rhs] per element, elements [first, first+count).  d_K may be NULL (then only the checksum
 * below is produced); d_checksum[count] receives sum_ij K_e[i][j]*(1 + ((i*31 + j*17) % 7)) (streaming mode,
 * SURVEY.md §0 D6).  K_e is symmetric bit for bit, like the reference's selfadjointView copy (:176-182).  From order 4 the stored
 * matrices are formed in the tiled layout of l3k_local_assemble_tiled and turned by a transposition kernel on a second stream
 * (the function returns when the matrices are complete); l3k_tuning selects the other routes (direct store, dense product). */
int l3k_local_assemble(l3k_mf* mf, int64_t first, int64_t count, double* d_K, double* d_F, double* d_checksum);

/* scatterLocalSystem for a batch, algsys/ScatterLocalSystem.hpp:24-54 (called per element by assembleGlobalSystem,
 * algsys/AssembleGlobalSystem.hpp:20-53): the element matrices / right-hand sides of elements [first, first+count) -- the
 * d_K / d_F of l3k_local_assemble, in its layouts -- are summed into the caller's global system on the device.
 *   matrix: d_values[nnz] over the caller's CSR graph of the rank-local matrix (d_row_ptr[n_local_dofs + 1], d_col_ind[nnz]
 *           ascending within a row; local dof numbering, rows and columns = node * dofs_per_node + field_inds[u], the
 *           reference's row_dofs / col_dofs): values[pos(row, col)] += K_e[i][j], atomically.  What Tpetra's
 *           sumIntoLocalValues does per element row happens here for the whole batch; the host takes the finished
 *           values array (one setAllValues, or one sumIntoLocalValues per batch of rows).
 *   rhs:    d_rhs[r * ldr + row] += F_e[i][r], atomically (the reference's std::atomic_ref fetch_add).
 * Entries whose (row, col) is not in the graph are skipped and counted in *n_missing (may be NULL), as
 * sumIntoLocalValues does.  skip_dirichlet != 0: rows and columns of dofs flagged in the mesh's Dirichlet mask are left
 * out, which makes the assembled operator the matrix-free one on the free dofs (gather reads Dirichlet dofs as 0, scatter
 * skips them, algsys/MatrixFreeSystem.hpp:441-466,517-536); 0 = the plain sum of the reference's assembled path.
 * Either of (d_K, d_values) / (d_F, d_rhs) may be NULL together. */
int l3k_assembled_scatter(l3k_mf* mf, int64_t first, int64_t count, const double* d_K, const double* d_F,
                          const int64_t* d_row_ptr, const int32_t* d_col_ind, double* d_values, double* d_rhs, size_t ldr,
                          int skip_dirichlet, int64_t* n_missing);
/* assembleGlobalSystem (algsys/AssembleGlobalSystem.hpp:20-53: for every element assembleLocalSystem, then scatterLocalSystem)
 * for the elements [first, first+count) in ONE call: the library forms sub-batches of element systems in a workspace of its
 * own (workspace_bytes in total, 0 = 1 GiB; two halves) on the context's stream while the previous sub-batch is summed into
 * d_values / d_rhs on a second stream -- the element matrices never leave the device and are never seen by the host.  Same
 * graph, rhs, skip_dirichlet and n_missing semantics as l3k_assembled_scatter; d_rhs may be NULL.  Returns when the work is
 * complete (it reads back the degenerate-element flag: error -2, "Encountered degenerate element", AssembleLocalSystem.hpp:249). */
/* K_e of the elements [first, first+count) in the TILED layout that l3k_assemble_global keeps between its two kernels, for
 * consumers that do not need the reference's row-major matrix: per element the U x U blocks K[(b,u),(b',u')] with the nodes
 * b = bx + n(by + n bz), b' = bx' + n(by' + n bz') (n = order + 1) stored as [u][u'][bx'][bz][bx][by][by'][bz'], bz' fastest --
 * Nd^2 doubles as in the row-major layout, every entry present (no mirroring left to the reader).  The stores of a wave of the
 * assembly kernel fill contiguous memory (the row-major stores are 8-byte pieces at a stride of 8U bytes, measured x3.9
 * write traffic), a reader finds a matrix row in 4n runs of n^2 doubles.  Shapes without the sum-factorised assembly
 * kernel (order 8) return an error. */
int l3k_local_assemble_tiled(l3k_mf* mf, int64_t first, int64_t count, double* d_Kt);
int l3k_assemble_global(l3k_mf* mf, int64_t first, int64_t count, const int64_t* d_row_ptr, const int32_t* d_col_ind,
                        double* d_values, double* d_rhs, size_t ldr, int skip_dirichlet, size_t workspace_bytes,
                        int64_t* n_missing);

/* ---- ghost exchange of a partitioned system: RCCL neighbour send / receive behind the C ABI -----------------------------
 * Stands in for comm::Import / comm::Export and their ImportExportContext (comm/ImportExport.hpp:29-72,130-215,295-372,
 * 402-470) and for the communication part of MatrixFreeSystem::applyImpl (algsys/MatrixFreeSystem.hpp:1046-1111).  The
 * reference exchanges host memory through MPI; here the host passes device pointers and the library issues one
 * ncclGroupStart / ncclSend + ncclRecv per neighbour / ncclGroupEnd per direction on a stream of its own (RCCL over xGMI),
 * ordered against the context's stream by events.  RCCL (librccl.so.1) is loaded when the first halo is created.
 *
 * l3k_halo_unique_id: ncclGetUniqueId (128 bytes) -- one rank calls it and hands the bytes to the others out of band (the
 *   reference's host has MPI_Bcast for that).
 * l3k_halo_create: collective over the `world` ranks (ncclCommInitRank).  Exchange lists as l3k_host_mesh_view holds them:
 *   neighbour i = nbr_rank[i]; owned nodes send_nodes[send_offsets[i] .. send_offsets[i+1]) are read by it (import send,
 *   export receive); ghost nodes [ghost_offsets[i], ghost_offsets[i+1]) (numbered from 0 behind the owned nodes) are owned
 *   by it (import receive, export send).  A rank may list itself (periodic identification of its own nodes).
 * l3k_halo_import:     ghost rows <- owners' rows           (owner -> sharer copy)
 * l3k_halo_export_add: owners' rows += ghost rows           (sharer -> owner add; per neighbour in list order)
 * l3k_mf_apply_dist:   y <- alpha A x + beta y on the owned rows, x and y [ncols][ld] over the owned rows only: scale, pack,
 *   post import || first half of the interior elements, border elements, post export || second half of the interior
 *   elements, unpack-add, Dirichlet rows.  Every call returns with the work queued on the context's stream.
 *
 * The transport is a small function table (RCCL unless the caller brings one): a group of point-to-point messages between
 * `begin` and `end`, every message on the given HIP stream (the halo's communication stream), complete in stream order behind
 * `end` -- the contract of ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd.  l3k_halo_create_transport takes any table
 * (an MPI-with-device-pointers host, a test double); `destroy` (may be NULL) is called with `user` when the halo is destroyed
 * (not if its creation fails).  The library's own second implementation is the IN-PROCESS transport: the ranks are threads
 * of one process, each with its own context and stream on one GPU or on several, and a message is a device-to-device copy
 * ordered by events -- for single-process multi-GPU hosts, and the seam through which the tests run l3k_mf_apply_dist with
 * several ranks on a one-GPU box (tests/MpiImportExportTest.cpp:17-136 runs the reference's exchange on np in {1, 2, 4}).
 * l3k_inproc_group_create(world): the shared mailboxes; l3k_inproc_transport(group, rank, &table): rank's table (the group
 * owns it and must outlive the halos). */
typedef struct l3k_halo_transport
{
    void* user;
    int (*group_begin)(void* user);
    int (*send)(void* user, const double* d_buf, size_t count, int peer, void* hip_stream);
    int (*recv)(void* user, double* d_buf, size_t count, int peer, void* hip_stream);
    int (*group_end)(void* user, void* hip_stream);
    void (*destroy)(void* user);
} l3k_halo_transport;
typedef struct l3k_inproc_group l3k_inproc_group;
typedef struct l3k_halo l3k_halo;
int     l3k_halo_unique_id(char* id128);
int     l3k_halo_create(l3k_ctx* ctx, const char* id128, int rank, int world, int dofs_per_node, int n_nbrs, const int* nbr_rank,
                        const int64_t* send_offsets, const int32_t* send_nodes, const int64_t* ghost_offsets, l3k_halo** out);
int     l3k_halo_create_transport(l3k_ctx* ctx, const l3k_halo_transport* transport, int rank, int world, int dofs_per_node,
                                  int n_nbrs, const int* nbr_rank, const int64_t* send_offsets, const int32_t* send_nodes,
                                  const int64_t* ghost_offsets, l3k_halo** out);
int     l3k_inproc_group_create(int world, l3k_inproc_group** out);
int     l3k_inproc_group_destroy(l3k_inproc_group* group);
int     l3k_inproc_transport(l3k_inproc_group* group, int rank, l3k_halo_transport* out);
int     l3k_halo_destroy(l3k_halo* halo);
int64_t l3k_halo_n_ghost_dofs(const l3k_halo* halo);
int     l3k_halo_import(l3k_halo* halo, const double* d_owned, size_t ld, int ncols, double* d_ghost, size_t ldg);
int     l3k_halo_export_add(l3k_halo* halo, const double* d_ghost, size_t ldg, int ncols, double* d_owned, size_t ld);
/* HIP events around the three element launches (first interior half, border elements, second interior half) of the next
 * n_applies calls of l3k_mf_apply_dist, on the stream they run on; _get waits for that apply and returns the durations. */
int     l3k_halo_timing_begin(l3k_halo* halo, int n_applies);
int     l3k_halo_timing_get(l3k_halo* halo, int apply, double ms[3]);
int     l3k_mf_apply_dist(l3k_mf* mf, l3k_halo* halo, const double* d_x, size_t ldx, double* d_y, size_t ldy, int ncols,
                          double alpha, double beta);

/* ---- host-side synthetic mesh + block partition --------------------------------------------------------------------
 * Stand-in for makeCubeMesh + convertMeshToOrder + partitionMesh + the ownership / import-export context
 * (mesh/primitives/CubeMesh.hpp:16-138, mesh/ConvertMeshToOrder.hpp:51-104, mesh/PartitionMesh.hpp:142-183,
 * util/SegmentedOwnership.hpp:11-45, comm/ImportExport.hpp:29-72) for structured hex cubes: ne[3] elements per edge
 * on [0,1]^3, order p, parts[3] blocks, this rank's block `rank` = bx + parts[0]*(by + parts[1]*bz).
 * perturb: vertices moved by perturb*h*sin(2 pi x)sin(2 pi y)sin(2 pi z).  Host only, no GPU. */
int l3k_cube_partition_create(const int ne[3], int order, const int parts[3], int rank, double perturb,
                              l3k_hostmesh** out);
int l3k_hostmesh_destroy(l3k_hostmesh* hm);
typedef struct
{
    int             dim, order;
    int64_t         n_elems, n_interior_elems;
    int64_t         n_owned_nodes, n_ghost_nodes;
    int64_t         global_node_base;      /* first global node id owned by this rank (contiguous ownership)        */
    int64_t         n_global_nodes;
    const uint32_t* elem_nodes;            /* [n_elems][N], interior elements first                                  */
    const double*   elem_verts;            /* [n_elems][8][3]                                                        */
    const int64_t*  node_grid_id;          /* [n_owned+n_ghost] partition-independent id gx + NX*(gy + NY*gz)        */
    const uint8_t*  node_boundary;         /* [n_owned+n_ghost] bit s set if the node lies on cube side s            */
                                           /* sides: 0 z=0, 1 z=1, 2 y=0, 3 y=1, 4 x=0, 5 x=1 (ElementTraits.hpp:84-95) */
    int             n_nbrs;                /* neighbours, ascending rank                                             */
    const int*      nbr_rank;              /* [n_nbrs]                                                               */
    const int64_t*  send_offsets;          /* [n_nbrs+1] into send_nodes: owned nodes each neighbour shares (import
                                              send / export receive), ascending local id                             */
    const int32_t*  send_nodes;
    const int64_t*  ghost_offsets;         /* [n_nbrs+1] ghost-node ranges (relative to n_owned_nodes) owned by each
                                              neighbour (import receive / export send); ghosts are sorted by global id */
    const uint8_t*  elem_boundary;         /* [n_elems] bit s set if side s of the element lies on cube side s: the
                                              boundary views of makeCubeMesh (mesh/primitives/CubeMesh.hpp:66-138)   */
    const int64_t*  ghost_global_id;       /* [n_ghost] global node id of every ghost (ascending)                    */
} l3k_hostmesh_view;
int l3k_hostmesh_view_get(const l3k_hostmesh* hm, l3k_hostmesh_view* out);

/* ---- order elevation on the device (mesh::convertMeshToOrder, mesh/ConvertMeshToOrder.hpp:51-104, followed
 * by the [non-internal | internal] renumbering of mesh/LocalMeshView.hpp:425-458) --------------------------------------
 * conn: host, [n_elems][8] vertex ids of an order-1 hex mesh, local vertex v = i + 2j + 4k.  Writes elem_nodes (host,
 * [n_elems][(order+1)^3], local node i + n(j + n k)) numbered [vertices | edge nodes | face nodes | element-internal
 * nodes, contiguous per element]: ready for l3k_mesh_desc.elem_nodes of a single-rank mesh (n_owned_nodes = *n_nodes).
 * Shared edges / faces are found by sorting their vertex keys on the device; every element is processed in parallel.    */
int l3k_elevate_order(l3k_ctx* ctx, int64_t n_elems, const uint32_t* conn, int64_t n_vertices, int order, uint32_t* elem_nodes,
                      int64_t* n_nodes, int64_t* n_noninternal);

/* ---- native results file (host only; post/NativeIO.hpp:15-60 save, :115-146 LoadedResults, :277-295 loadResultsImpl) --
 * "L3STER results file\nv1.0\n// <comment>\n", size_t n_fields, size_t n_nodes_global, then n_fields arrays of
 * n_nodes_global doubles indexed by global node id.  Every rank saves its owned slice [node_begin, node_begin + n_local)
 * of each field (fields: host, [n_fields][ld]); exactly one rank passes write_header != 0 (the reference: rank 0).  No
 * ordering between the ranks' calls is required.  Loading gathers field values by global node id (node_ids == NULL:
 * the contiguous range starting at node_begin), as Loader::loadResults does with the saved partition's node ids.       */
int l3k_results_save(const char* path, const char* comment, size_t n_fields, int64_t n_global_nodes, int64_t node_begin,
                     int64_t n_local_nodes, const double* fields, size_t ld, int write_header);
int l3k_results_info(const char* path, size_t* n_fields, size_t* n_nodes);
int l3k_results_load(const char* path, size_t field, int64_t n, const int64_t* node_ids, int64_t node_begin, double* out);

/* ---- native mesh file (host only; post/NativeIO.hpp:75-108 save(comm, mesh, path, comment), :161-232
 * extractSavedPartitionInfo / loadUnifiedMesh / loadPartitionedMesh; mesh/MeshUtils.hpp:318-360 serializeMesh /
 * deserializeMesh; util/Serialization.hpp:20-66) ------------------------------------------------------------------------
 * "L3STER mesh file\nv1.0\n// <comment>\n", size_t n_parts, n_parts size_t part sizes, then every rank's serialised
 * MeshPartition<order>: its domains in ascending id, each the elements of type Hex, Quad, Line (mesh/ElementType.hpp:11-16)
 * as { uint64 nodes[(p+1)^d] (GLOBAL ids); double vertices[2^d][3]; uint64 id }, then nodes_begin, num_owned_nodes and
 * the boundary domain ids.  The structs below are the structure-of-arrays view of one part.  Writing: every rank
 * computes its size (l3k_meshfile_part_bytes), the sizes are gathered by the caller (the reference: comm.gather, :83),
 * every rank then calls l3k_meshfile_save with the full size table and the SAME comment (its length fixes the offsets of
 * the parts); exactly one passes write_header != 0; no ordering between the ranks' calls is required.  Reading: l3k_meshfile_load parses part `part` as a mesh of the given order
 * (the order is not stored in the file: the reference's loader takes it as a template argument); all parts in turn give
 * loadUnifiedMesh.                                                                                                      */
typedef struct
{
    size_t          n;
    const uint64_t* nodes; /* [n][(order+1)^d] global node ids, element-local lexicographic order                       */
    const double*   verts; /* [n][2^d][3]                                                                                */
    const uint64_t* ids;   /* [n] element ids (unique over the whole mesh, boundary elements included)                   */
} l3k_meshfile_elems;
typedef struct
{
    uint16_t           id; /* d_id_t (common/Typedefs.hpp)                                                               */
    l3k_meshfile_elems hex, quad, line;
} l3k_meshfile_domain;
typedef struct
{
    int                        order;
    size_t                     n_domains;
    const l3k_meshfile_domain* domains;
    uint64_t                   nodes_begin;   /* first owned global node id (0 if none owned, MeshUtils.hpp:327)        */
    size_t                     n_owned_nodes;
    size_t                     n_boundary_ids;
    const uint16_t*            boundary_ids;
} l3k_meshfile_part_desc;
typedef struct l3k_meshfile_part l3k_meshfile_part;
int l3k_meshfile_part_bytes(const l3k_meshfile_part_desc* desc, size_t* bytes);
int l3k_meshfile_save(const char* path, const char* comment, size_t n_parts, const size_t* part_bytes, size_t part,
                      const l3k_meshfile_part_desc* desc, int write_header);
int l3k_meshfile_info(const char* path, size_t* n_parts, size_t* part_bytes, size_t capacity);
int l3k_meshfile_load(const char* path, size_t part, int order, l3k_meshfile_part** out);
int l3k_meshfile_part_get(const l3k_meshfile_part* part, l3k_meshfile_part_desc* out); /* pointers live as long as part */
int l3k_meshfile_part_destroy(l3k_meshfile_part* part);

#ifdef __cplusplus
}
#endif
#endif
