/* oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  This library is a plain CPU restatement of the element-local hot path of
 * kubagalecki/L3STER (reference snapshot under /root/reference, cited per function as file:line relative to that
 * root).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the checker /
 * reported baseline.  The product path (l3ster_amd/, include/l3k.h) never links or calls it.
 *
 * Parity pinning: the reference holds NO stored golden vectors; it cannot be compiled here
 * (needs Eigen/TBB/Trilinos/GCC>=14, all absent).  This oracle is therefore pinned by the reference's own analytic
 * known-answer tests re-expressed in tests/test_oracle_kats.py (K1..K5 of SURVEY.md §8c) and cross-checked against an
 * independent numpy/mpmath restatement (oracle/oracle_np.py) through committed fixtures (tests/golden/).
 *
 * Layout conventions:
 *   hex node index I = ix + n*(iy + n*iz), n = p+1          basisfun/ReferenceBasisFunction.hpp:107-117
 *   vertices v = i + 2j + 4k                                 mesh/primitives/CubeMesh.hpp:46-61
 *   element dof index = node*U + u                           algsys/AssembleLocalSystem.hpp:159
 *   Jacobi matrix J[d][s] = d x_s / d xi_d                   mapping/JacobiMat.hpp:36
 */
#ifndef L3K_ORACLE_H
#define L3K_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* kernel ids (same numbering as include/l3k.h; the definitions are restated independently in oracle.cpp) */
enum
{
    ORC_KERNEL_DIFFUSION3D     = 0, /* benchmarks/Diffusion3D.hpp:51-79 == tests/Kernels.hpp:55-81 (+rhs[0]=s)     */
    ORC_KERNEL_DIFFUSION3D_VAR = 1, /* tests/Kernels.hpp:84-118, F=1                                                 */
    ORC_KERNEL_DIFFUSION2D     = 2, /* tests/Kernels.hpp:5-24                                                        */
    ORC_KERNEL_DIFFUSION2D_VAR = 3, /* tests/Kernels.hpp:27-52, F=1                                                  */
    ORC_KERNEL_ADVDIFF3D       = 4, /* synthetic config 5: Diffusion3D rows + sigma, u.grad, F=3     */
    /* boundary equation kernels (the input carries the outward normal) */
    ORC_KERNEL_ADIABATIC2D     = 5, /* tests/Kernels.hpp:120-128: q.n = 0                                            */
    ORC_KERNEL_ADIABATIC3D     = 6, /* 3-D twin                                                                      */
    ORC_KERNEL_ROBIN3D         = 7, /* synthetic: q.n + h T = h T_inf, kp = {h, T_inf}                               */
    ORC_KERNEL_MASS3D          = 8, /* A0 = I, rhs = (1, 2): pins w * detJ                                           */
    ORC_KERNEL_NORMALFLUX3D    = 9, /* boundary kernel with derivative operators A1..A3                              */
    /* domain kernels that read the space-time point (examples/03-advection-2D/source.cpp:52-66,
     * examples/04-periodic-bc/source.cpp:60-95 do) and kernels with an odd number of unknowns */
    ORC_KERNEL_DIFFUSION3D_POINT = 10, /* Diffusion3D with k, A0 and s functions of (x, y, z, t); kp = {k0, s0}    */
    ORC_KERNEL_ADVECTION3D       = 11, /* scalar BDF3 advection, U = E = 1, F = 3, velocity from the point; kp = {dt} */
    ORC_KERNEL_DIVCURL3D         = 12, /* div-curl system, U = 3, E = 4; kp = {f}                                   */
    ORC_KERNEL_NS3D              = 13, /* benchmarks/Kernels.hpp:3-65: U = 7, E = 8, F = 7                          */
    ORC_KERNEL_ROBINPOINT3D      = 14  /* boundary kernel reading point and time: q.n + h(x,t) T = h(x,t) T_inf(x); kp = {h0, t0} */
};
/* residual kernels for integrals / L2 norms (fields are the kernel's n_fields inputs) */
enum
{
    ORC_RESIDUAL_DIFFUSION3D_ERROR = 0, /* benchmarks/Diffusion3D.hpp:81-103, F = E = 4, kp = {k, s}                 */
    ORC_RESIDUAL_LINEAR2D_ERROR    = 1, /* tests/Diffusion2D.hpp:84-92 (exact T = x), F = E = 3                     */
    ORC_RESIDUAL_LINEAR3D_ERROR    = 2, /* 3-D twin, F = E = 4                                                      */
    ORC_RESIDUAL_UNIT2D            = 3, /* tests/MappingTests.cpp:567-569: integrand 1, F = 0, E = 1                */
    ORC_RESIDUAL_UNIT3D            = 4,
    ORC_RESIDUAL_COORDX2D          = 5, /* tests/Diffusion2D.hpp:49-50: out[0] = x (Dirichlet value kernel), F = 0  */
    ORC_RESIDUAL_COORDX3D          = 6
};

/* dims of a kernel: returns 0 on success */
int orc_kernel_params(int kernel_id, int* dim, int* n_eq, int* n_unk, int* n_fields);

/* ---- tables ------------------------------------------------------------------------------------------------ */
/* Gauss-Lobatto-Legendre abscissae, ascending.   math/LobattoRuleAbsc.hpp:11-35 */
int orc_gll_nodes(int n, double* x);
/* Gauss-Legendre rule with nq points, ascending. math/ComputeGaussRule.hpp:26-60, quad/ReferenceQuadrature.hpp:24-51 */
int orc_gl_rule(int nq, double* x, double* w);
/* quadrature size rule: nq1d = value_order*p + derivative_order*(p-1) + 1.  algsys/AssembleLocalSystem.hpp:32-35,
 * quad/ReferenceQuadrature.hpp:18 */
int orc_n_qps1d(int p, int value_order, int derivative_order);
/* 1-D tables I[b][q] = phi_b(x_q), D[b][q] = phi_b'(x_q), row-major n x nq.  algsys/SumFactorization.hpp:25-65 */
int orc_basis_1d(int p, int nq, double* I, double* D);
/* 1-D Lagrange basis on GLL nodes of order p evaluated at an arbitrary point: vals[n], ders[n].
 * basisfun/ReferenceBasisFunction.hpp:28-72 */
int orc_lagrange_1d(int p, double x, double* vals, double* ders);
/* full reference basis at the tensor quadrature (local-element path), QP index with xi SLOWEST
 * (quad/GenerateQuadrature.hpp:64-71): vals[nqp][N], ders[nqp][dim][N], weights[nqp], points[nqp][dim].
 * basisfun/ReferenceElementBasisAtQuadrature.hpp:10-19 */
int orc_ref_basis_at_qps(int dim, int p, int nq, double* vals, double* ders, double* weights, double* points);
/* odd-even decomposition check (K4): runs the 4 sweep primitives standard and odd-even on `in` (rows x cols,
 * col-major), writes max abs difference per primitive into err[4].  algsys/SumFactorization.hpp:67-86,159-343 */
int orc_oddeven_check(int p, int nq, int cols, const double* in_back, const double* in_fwd, double* err);

/* ---- mapping ----------------------------------------------------------------------------------------------- */
/* J[d][s] at reference point, from the 2^dim vertices only.  mapping/JacobiMat.hpp:15-45 */
int orc_jacobi_mat(int dim, const double* verts /*[2^dim][3]*/, const double* point /*[dim]*/, double* J /*[dim][dim]*/);
/* reference -> physical.  mapping/MapReferenceToPhysical.hpp:14-25 */
int orc_map_to_physical(int dim, const double* verts, const double* point, double* xyz /*[3]*/);
/* physical location of local node i of an order-p element.  mesh/NodePhysicalLocation.hpp */
int orc_node_location(int dim, int p, const double* verts, int node, double* xyz);

/* ---- element-local operators ------------------------------------------------------------------------------- */
/* common arguments:
 *   verts       [2^dim][3]
 *   node_fields [N][F] row-major (F = n_fields of the kernel; may be NULL when F == 0)
 *   kparams     kernel parameter block (doubles; may be NULL -> defaults)
 *   x, y, F_e   column-major [Nd][R], Nd = N*U, dof = node*U + u
 *   K           row-major [Nd][Nd]
 */
/* assembleLocalSystem.  algsys/AssembleLocalSystem.hpp:77-216,234-256.  returns -2 on detJ <= 0 (line 249) */
int orc_assemble_local(int kernel_id, int p, int nq, int R, const double* verts, const double* node_fields,
                       const double* kparams, double time, double* K, double* F_e);
/* evaluateLocalOperator.  algsys/EvaluateLocalOperator.hpp:37-146,211-236 */
int orc_apply_local(int kernel_id, int p, int nq, int R, const double* verts, const double* node_fields,
                    const double* kparams, double time, const double* x, double* y);
/* precomputeOperatorDiagonalAndRhs.  algsys/EvaluateLocalOperator.hpp:172-208,276-301
 * dir_inds: element-local dof indices, dir_vals column-major [n_dir][R] */
int orc_diag_rhs_local(int kernel_id, int p, int nq, int R, const double* verts, const double* node_fields,
                       const double* kparams, double time, int n_dir, const int* dir_inds, const double* dir_vals,
                       double* diag, double* rhs);
/* evalLocalOperatorSumFact (Quad / Hex), sweeps in the reference's order, optional odd-even decomposition.
 * algsys/SumFactorization.hpp:438-504,614-756,758-917.  pass_true_z != 0 passes the true z to the kernel instead
 * of the reference's z = 0 (algsys/SumFactorization.hpp:732). */
int orc_apply_sumfact(int kernel_id, int p, int nq, int R, int odd_even, int pass_true_z, const double* verts,
                      const double* node_fields, const double* kparams, double time, const double* x, double* y);
/* process-wide switch for orc_mf_apply: on != 0 -> domain kernels see z = 0 as in the reference's evalAtHexQPs
 * (algsys/SumFactorization.hpp:732); default 0 -> the true z.  orc_mf_diag_rhs and the local-element functions always
 * pass the true z (algsys/AssembleLocalSystem.hpp:229-230). */
int orc_set_reference_z0(int on);

/* ---- mesh-level matrix-free operator (one rank) ------------------------------------------------------------ */
typedef struct
{
    int             dim, p, nq;         /* element shape                                                          */
    int64_t         n_elems;
    const uint32_t* elem_nodes;         /* [n_elems][N] local node ids, lexicographic                             */
    const double*   elem_verts;         /* [n_elems][2^dim][3]                                                    */
    int64_t         n_local_nodes;      /* owned + ghost nodes                                                    */
    int             dofs_per_node;      /* dof(node, k) = node*dofs_per_node + k                                  */
    const int*      field_inds;         /* [U] which per-node dof each unknown of the kernel maps to             */
    const uint8_t*  dirichlet;          /* [n_local_nodes*dofs_per_node] byte mask, may be NULL                  */
    const double*   fields;             /* SoA [F][n_local_nodes]  post/FieldAccess.hpp:21-30, may be NULL      */
} orc_mesh;

/* y <- alpha*A*x + beta*y on local dofs (x,y column-major [n_local_dofs][ncols] with leading dims ldx,ldy):
 * y scaling (MatrixFreeSystem.hpp:1038), gather with Dirichlet->0 (:421-467), sum-factorised element apply,
 * scatter-add skipping Dirichlet (:494-537), then y[d] += alpha*x[d] on the first n_owned_dofs Dirichlet rows
 * (:1087-1098).  elements [e_begin, e_end) only; do_scale / do_dirichlet_rows let the caller split interior/border.
 * nthreads > 1: element loop split over std::threads with atomic adds (the reference's TBB + atomic_ref scheme). */
int orc_mf_apply(const orc_mesh* m, int kernel_id, const double* kparams, double time, int odd_even, int ncols,
                 const double* x, size_t ldx, double* y, size_t ldy, double alpha, double beta, int64_t e_begin,
                 int64_t e_end, int do_scale, int do_dirichlet_rows, int64_t n_owned_dofs, int nthreads);
/* diag + rhs (MatrixFreeSystem.hpp:888-941 minus the export): diag[n_local_dofs], rhs [n_local_dofs][R] (ld),
 * dirichlet_vals [n_local_dofs][R] (ld) or NULL.  Accumulates (caller zeroes); finalize != 0 sets diag=1, rhs=g on
 * the first n_owned_dofs Dirichlet rows (:911-915). */
int orc_mf_diag_rhs(const orc_mesh* m, int kernel_id, const double* kparams, double time, int R,
                    const double* dirichlet_vals, size_t ldg, double* diag, double* rhs, size_t ldr,
                    int64_t e_begin, int64_t e_end, int finalize, int64_t n_owned_dofs, int nthreads);

/* MatrixFreeSystem::updateSolution (algsys/MatrixFreeSystem.hpp:1231-1273): x column-major [n_local_dofs][n_rhs] (owned then ghost
 * rows), fields SoA [n_fields][n_local_nodes]; nodes no element holds keep their field values */
int orc_update_solution(const orc_mesh* m, const double* x, size_t ldx, int n_rhs, int n_inds, const int* sol_inds,
                        const int* sol_man_inds, double* fields, int n_fields);

/* ---- boundary terms and post-processing ---------------------------------------------- */
/* side < 0: domain.  Sides as mesh/ElementTraits.hpp:84-95: hex 0..5 = z-,z+,y-,y+,x-,x+; quad 0..3 = y-,y+,x-,x+ */
/* full element basis at the quadrature of one side: vals[nq^(dim-1)][N], ders[.][dim][N], weights, points[.][dim].
 * basisfun/ReferenceElementBasisAtQuadrature.hpp:21-97, mapping/ReferenceBoundaryToSideMapping.hpp */
int orc_side_basis_at_qps(int dim, int p, int nq, int side, double* vals, double* ders, double* weights, double* points);
/* outward unit normal and surface jacobian.  mapping/BoundaryNormal.hpp:8-64, mapping/BoundaryIntegralJacobian.hpp */
int orc_boundary_geometry(int dim, const double* verts, const double* point, int side, double* normal, double* jacobian);
/* the local-element functions above on one side of the element with a boundary equation kernel
 * (algsys/AssembleLocalSystem.hpp:258-281, algsys/EvaluateLocalOperator.hpp:238-274,303-330) */
int orc_assemble_local_side(int side, int kernel_id, int p, int nq, int R, const double* verts, const double* node_fields,
                            const double* kparams, double time, double* K, double* F_e);
int orc_apply_local_side(int side, int kernel_id, int p, int nq, int R, const double* verts, const double* node_fields,
                         const double* kparams, double time, const double* x, double* y);
int orc_diag_rhs_local_side(int side, int kernel_id, int p, int nq, int R, const double* verts, const double* node_fields,
                            const double* kparams, double time, int n_dir, const int* dir_inds, const double* dir_vals,
                            double* diag, double* rhs);
int orc_residual_params(int residual_id, int* dim, int* n_eq, int* n_fields);
/* out[E] += integral over the element (side < 0) or one of its sides of the residual kernel (squared if square != 0).
 * post/Integral.hpp:11-52, post/NormL2.hpp:21-29.  node_fields [N][F] */
int orc_integrate_local(int side, int residual_id, int p, int nq, int square, const double* verts,
                        const double* node_fields, const double* kparams, double time, double* out);
/* out[E] = sum over all elements (n_faces < 0) or the listed element sides; m->fields = the F fields (SoA).
 * post/Integral.hpp:54-111 */
int orc_mf_integrate(const orc_mesh* m, int residual_id, int nq, int square, const double* kparams, double time,
                     int64_t n_faces, const int64_t* face_elem, const uint8_t* face_side, double* out);
/* y += alpha * A_b x for a boundary equation kernel on the listed element sides (Dirichlet semantics of orc_mf_apply) */
int orc_bnd_apply(const orc_mesh* m, int kernel_id, const double* kparams, double time, int ncols, int64_t n_faces,
                  const int64_t* face_elem, const uint8_t* face_side, const double* x, size_t ldx, double* y, size_t ldy,
                  double alpha);
/* diag += diag(A_b), rhs += B_b^T W (f_b - B_b g) on the listed element sides */
int orc_bnd_diag_rhs(const orc_mesh* m, int kernel_id, const double* kparams, double time, int R, int64_t n_faces,
                     const int64_t* face_elem, const uint8_t* face_side, const double* dirichlet_vals, size_t ldg,
                     double* diag, double* rhs, size_t ldr);

/* computeValuesAtNodes (algsys/ComputeValuesAtNodes.hpp:371-448,508-594): residual kernel evaluated at the nodes of the
 * listed element sides (n_faces >= 0) or of all elements (n_faces < 0); sum[dof(node, dof_inds[e])] += value_e,
 * count[...] += 1; the average sum/count (where count > 0) is the nodal value (:112-154) */
int orc_values_at_nodes(const orc_mesh* m, int residual_id, const double* kparams, double time, int64_t n_faces,
                        const int64_t* face_elem, const uint8_t* face_side, const int* dof_inds, double* sum, double* count);

const char* orc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
