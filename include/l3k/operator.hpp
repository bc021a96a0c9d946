// operator.hpp -- C++ host-side mirror of the reference's interface for the hot path, on top of the C ABI (l3k.h).
//
// The reference is C++ (header-only); this header gives its maintainers the same vocabulary over libl3k.so:
//   l3k::CubeMesh                ~ generateAndDistributeMesh<order>(makeCubeMesh(...))   comm/DistributeMesh.hpp:284-299
//   l3k::MatrixFreeSystem        ~ algsys::MatrixFreeSystem                               algsys/MatrixFreeSystem.hpp:19-249
//       .apply(X, Y, alpha, beta)   ~ Operator::apply                                      :34-41
//       .diagAndRhs(...)            ~ endAssembly() -> computeDiagAndRhs                   :877-941
//       .assembleLocal(...)         ~ assembleLocalSystem                                  algsys/AssembleLocalSystem.hpp:234-256
//       .assembleProblem(bnd)       ~ assembleProblem(boundary kernel, boundary_ids)      algsys/MatrixFreeSystem.hpp:58-68
//       .apply(halo, X, Y, ...)     ~ applyImpl of a partitioned system, exchange included    :1020-1140
//       .scatterLocalSystems(...)   ~ scatterLocalSystem / assembleGlobalSystem             algsys/ScatterLocalSystem.hpp:24-54
//   l3k::Halo                    ~ comm::ImportExportContext + comm::Import / comm::Export  comm/ImportExport.hpp:29-72,130-215
//   l3k::BoundaryTerm            ~ a BoundaryEquationKernel on a set of boundary views   algsys/EvaluateLocalOperator.hpp:238-330
//   l3k::computeIntegral / computeNormL2 ~ post/Integral.hpp:113-128, post/NormL2.hpp:31-62 (one rank)
// Errors: the reference throws std::runtime_error from util::throwingAssert (util/Assertion.hpp:88-95); so does this
// wrapper, with the text of l3k_last_error().  RAII handles, no torch, no other dependency than l3k.h + the HIP runtime
// of the caller for device memory.
#ifndef L3K_OPERATOR_HPP
#define L3K_OPERATOR_HPP

#include "l3k.h"

#include <array>
#include <cmath>
#include <cstdint>
#include <span>
#include <stdexcept>
#include <string>
#include <vector>

namespace l3k
{
inline void check(int rc)
{
    if (rc != 0)
        throw std::runtime_error{std::string{"libl3k: "} + l3k_last_error()};
}

// AssemblyOptions, algsys/AssembleLocalSystem.hpp:24-49
struct AssemblyOptions
{
    int value_order = 1, derivative_order = 0, eval_strategy = 0;
};

class Context
{
public:
    explicit Context(int hip_device = 0, void* hip_stream = nullptr) { check(l3k_ctx_create(hip_device, hip_stream, &m_ctx)); }
    Context(const Context&)            = delete;
    Context& operator=(const Context&) = delete;
    ~Context() { l3k_ctx_destroy(m_ctx); }
    void     setStream(void* hip_stream) { check(l3k_ctx_set_stream(m_ctx, hip_stream)); }
    // bitwise-reproducible element launches for the meshes created from now on (the reference's relaxed atomics,
    // algsys/MatrixFreeSystem.hpp:513, are not reproducible run to run)
    void     setDeterministic(bool on = true) { check(l3k_ctx_set_deterministic(m_ctx, on ? 1 : 0)); }
    // the point a domain kernel sees under sum factorisation: the true z (default) or the reference's z = 0
    // (algsys/SumFactorization.hpp:732)
    void setReferenceZ0(bool on = true) { check(l3k_ctx_set_reference_z0(m_ctx, on ? 1 : 0)); }
    // launch-route settings (the defaults are the measured choices; tests and tools take the other routes on purpose)
    l3k_tuning tuning() const
    {
        l3k_tuning t{};
        check(l3k_ctx_get_tuning(m_ctx, &t));
        return t;
    }
    void setTuning(const l3k_tuning& t) { check(l3k_ctx_set_tuning(m_ctx, &t)); }
    void     synchronize() { check(l3k_ctx_synchronize(m_ctx)); }
    l3k_ctx* get() const { return m_ctx; }

private:
    l3k_ctx* m_ctx{};
};

// One rank's block of the synthetic structured hex mesh (host arrays, reference numbering conventions)
class CubeMesh
{
public:
    CubeMesh(std::array< int, 3 > ne, int order, std::array< int, 3 > parts = {1, 1, 1}, int rank = 0, double perturb = 0.)
    {
        check(l3k_cube_partition_create(ne.data(), order, parts.data(), rank, perturb, &m_mesh));
        check(l3k_hostmesh_view_get(m_mesh, &m_view));
    }
    CubeMesh(const CubeMesh&)            = delete;
    CubeMesh& operator=(const CubeMesh&) = delete;
    ~CubeMesh() { l3k_hostmesh_destroy(m_mesh); }
    const l3k_hostmesh_view& view() const { return m_view; }
    int64_t                  nLocalNodes() const { return m_view.n_owned_nodes + m_view.n_ghost_nodes; }
    // the boundary views of makeCubeMesh (mesh/primitives/CubeMesh.hpp:66-138): (element, side) pairs of this rank's
    // element sides on the cube sides selected by bit s of `sides`
    struct Sides
    {
        std::vector< int64_t > elems;
        std::vector< uint8_t > sides;
    };
    Sides boundarySides(unsigned sides = 0x3f) const
    {
        Sides out;
        for (int s = 0; s < 6; ++s)
            if (sides & (1u << s))
                for (int64_t e = 0; e < m_view.n_elems; ++e)
                    if (m_view.elem_boundary[e] & (1u << s))
                    {
                        out.elems.push_back(e);
                        out.sides.push_back(static_cast< uint8_t >(s));
                    }
        return out;
    }
    // BCDefinition::defineDirichlet(boundary_ids, {unknowns}) as a byte mask over local dofs; sides as in
    // mesh/ElementTraits.hpp:84-95 (bit s of `sides`)
    std::vector< uint8_t > dirichletMask(int dofs_per_node, std::span< const int > unknowns, unsigned sides = 0x3f) const
    {
        std::vector< uint8_t > mask(static_cast< size_t >(nLocalNodes()) * dofs_per_node, 0);
        for (int64_t n = 0; n < nLocalNodes(); ++n)
            if (m_view.node_boundary[n] & sides)
                for (int u : unknowns)
                    mask[n * dofs_per_node + u] = 1;
        return mask;
    }

private:
    l3k_hostmesh*     m_mesh{};
    l3k_hostmesh_view m_view{};
};

class DeviceMesh
{
public:
    DeviceMesh(Context& ctx, const CubeMesh& mesh, int dofs_per_node, const uint8_t* dirichlet = nullptr) : m_dpn{dofs_per_node}
    {
        const auto&         v = mesh.view();
        const l3k_mesh_desc d{v.dim, v.order, v.n_elems, v.n_interior_elems, v.elem_nodes, v.elem_verts, v.n_owned_nodes,
                              v.n_ghost_nodes, dofs_per_node, dirichlet};
        check(l3k_mesh_create(ctx.get(), &d, &m_mesh));
        m_owned = v.n_owned_nodes * dofs_per_node;
        m_ctx   = ctx.get();
    }
    DeviceMesh(const DeviceMesh&)            = delete;
    DeviceMesh& operator=(const DeviceMesh&) = delete;
    ~DeviceMesh() { l3k_mesh_destroy(m_mesh); }
    l3k_mesh* get() const { return m_mesh; }
    l3k_ctx*  ctx() const { return m_ctx; }
    int64_t   nOwnedDofs() const { return m_owned; }

private:
    l3k_mesh* m_mesh{};
    l3k_ctx*  m_ctx{};
    int64_t   m_owned{};
    int       m_dpn{};
};

// The mailboxes of the in-process transport: the ranks are threads of this process, each with its own Context (one GPU or
// several); must outlive the halos built on its tables
class InprocGroup
{
public:
    explicit InprocGroup(int world) { check(l3k_inproc_group_create(world, &m_group)); }
    InprocGroup(const InprocGroup&)            = delete;
    InprocGroup& operator=(const InprocGroup&) = delete;
    ~InprocGroup() { l3k_inproc_group_destroy(m_group); }
    l3k_halo_transport transport(int rank) const
    {
        l3k_halo_transport t{};
        check(l3k_inproc_transport(m_group, rank, &t));
        return t;
    }

private:
    l3k_inproc_group* m_group{};
};

// One rank's ghost exchange (ImportExportContext + Import / Export, comm/ImportExport.hpp:29-72,130-215) carried by RCCL
// inside the library.  uniqueId(): 128 bytes drawn by one rank and handed to the others by the host's own means (the
// reference has MPI_Bcast); the constructor is collective over the `world` ranks.
class Halo
{
public:
    static std::array< char, 128 > uniqueId()
    {
        std::array< char, 128 > id{};
        check(l3k_halo_unique_id(id.data()));
        return id;
    }
    Halo(Context& ctx, const CubeMesh& mesh, int dofs_per_node, const std::array< char, 128 >& unique_id, int rank, int world)
    {
        const auto& v = mesh.view();
        check(l3k_halo_create(ctx.get(), unique_id.data(), rank, world, dofs_per_node, v.n_nbrs, v.nbr_rank, v.send_offsets,
                              v.send_nodes, v.ghost_offsets, &m_halo));
    }
    // With a transport table instead of RCCL: the host's own (e.g. MPI on device pointers), or the library's in-process
    // transport for the threads of one process (InprocGroup below).  Not collective.
    Halo(Context& ctx, const CubeMesh& mesh, int dofs_per_node, const l3k_halo_transport& transport, int rank, int world)
    {
        const auto& v = mesh.view();
        check(l3k_halo_create_transport(ctx.get(), &transport, rank, world, dofs_per_node, v.n_nbrs, v.nbr_rank, v.send_offsets,
                                        v.send_nodes, v.ghost_offsets, &m_halo));
    }
    Halo(const Halo&)            = delete;
    Halo& operator=(const Halo&) = delete;
    ~Halo() { l3k_halo_destroy(m_halo); }
    int64_t nGhostDofs() const { return l3k_halo_n_ghost_dofs(m_halo); }
    // comm::Import: ghost rows <- owners' rows; comm::Export: owners' rows += ghost rows
    void importGhosts(const double* d_owned, size_t ld, int ncols, double* d_ghost, size_t ldg) const
    {
        check(l3k_halo_import(m_halo, d_owned, ld, ncols, d_ghost, ldg));
    }
    void exportAdd(const double* d_ghost, size_t ldg, int ncols, double* d_owned, size_t ld) const
    {
        check(l3k_halo_export_add(m_halo, d_ghost, ldg, ncols, d_owned, ld));
    }
    l3k_halo* get() const { return m_halo; }

private:
    l3k_halo* m_halo{};
};

// A boundary equation kernel on a list of element sides (the reference's assembleProblem(kernel, boundary_ids))
class BoundaryTerm
{
public:
    template < typename KernelParamBlock >
    BoundaryTerm(const DeviceMesh& mesh, int kernel_id, const KernelParamBlock& params, const CubeMesh::Sides& sides,
                 AssemblyOptions opts = {}, std::span< const int > field_inds = {}, int n_rhs = 1)
    {
        const l3k_asmopts o{opts.value_order, opts.derivative_order, opts.eval_strategy};
        check(l3k_bnd_create(mesh.ctx(), mesh.get(), kernel_id, &params, sizeof params, &o,
                             field_inds.empty() ? nullptr : field_inds.data(), n_rhs, int64_t(sides.elems.size()),
                             sides.elems.data(), sides.sides.data(), &m_bnd));
    }
    BoundaryTerm(const DeviceMesh& mesh, int kernel_id, const CubeMesh::Sides& sides, AssemblyOptions opts = {}, int n_rhs = 1)
    {
        const l3k_asmopts o{opts.value_order, opts.derivative_order, opts.eval_strategy};
        check(l3k_bnd_create(mesh.ctx(), mesh.get(), kernel_id, nullptr, 0, &o, nullptr, n_rhs,
                             int64_t(sides.elems.size()), sides.elems.data(), sides.sides.data(), &m_bnd));
    }
    BoundaryTerm(const BoundaryTerm&)            = delete;
    BoundaryTerm& operator=(const BoundaryTerm&) = delete;
    ~BoundaryTerm() { l3k_bnd_destroy(m_bnd); }
    void     setFields(const double* d_soa, size_t ld) { check(l3k_bnd_set_fields(m_bnd, d_soa, ld)); }
    void     setTime(double t) { check(l3k_bnd_set_time(m_bnd, t)); } // in.point.time of the boundary kernel
    l3k_bnd* get() const { return m_bnd; }

private:
    l3k_bnd* m_bnd{};
};

// computeIntegral (post/Integral.hpp:113-128) for one rank: integral of a residual kernel over the mesh (sides == nullptr)
// or over element sides; fields = SoA device array [n_fields][ld]
template < typename KernelParamBlock >
std::vector< double > computeIntegral(const DeviceMesh& mesh, int residual_id, const KernelParamBlock* params,
                                      const double* d_fields, size_t ldf, AssemblyOptions opts = {},
                                      const CubeMesh::Sides* sides = nullptr, bool square = false, double time = 0.)
{
    l3k_kparams kp{};
    check(l3k_residual_info(residual_id, &kp, nullptr, nullptr));
    std::vector< double > out(static_cast< size_t >(kp.n_equations));
    const l3k_asmopts     o{opts.value_order, opts.derivative_order, opts.eval_strategy};
    check(l3k_integrate(mesh.ctx(), mesh.get(), residual_id, params, params ? sizeof(KernelParamBlock) : 0, &o, d_fields,
                        ldf, time, square ? 1 : 0, sides ? int64_t(sides->elems.size()) : -1,
                        sides ? sides->elems.data() : nullptr, sides ? sides->sides.data() : nullptr, out.data()));
    return out;
}
// computeNormL2 (post/NormL2.hpp:31-62): squared residual, doubled quadrature orders, square root
template < typename KernelParamBlock >
std::vector< double > computeNormL2(const DeviceMesh& mesh, int residual_id, const KernelParamBlock* params,
                                    const double* d_fields, size_t ldf, AssemblyOptions opts = {},
                                    const CubeMesh::Sides* sides = nullptr, double time = 0.)
{
    opts.value_order *= 2;
    opts.derivative_order *= 2;
    auto out = computeIntegral(mesh, residual_id, params, d_fields, ldf, opts, sides, true, time);
    for (auto& v : out)
        v = std::sqrt(v);
    return out;
}

// algsys::MatrixFreeSystem for one rank without ghosts.  Kernel = registered functor id + POD parameter block
// (l3ster_amd/csrc/user_kernels.hpp); vectors are DEVICE pointers, column-major with leading dimension.
// MatrixFreeSystem::updateSolution(sol_inds, sol_man, sol_man_inds) (algsys/MatrixFreeSystem.hpp:1231-1273): the solution's per-node
// dofs sol_inds of every column into the fields sol_man_inds (index-major, one per (index, column)) of the SoA field storage that a
// kernel's FieldAccess reads; ghost rows from d_xghost (the imported values; nullptr on a rank without ghosts)
// computeValuesAtNodes (algsys/ComputeValuesAtNodes.hpp:217-594; the engine of setDirichletBCValues / setValues) on one rank: the
// residual kernel evaluated at the nodes of the listed element sides (sides == nullptr: of every element), equation e written to
// per-node dof dof_inds[e] of d_values (all local dofs), element contributions averaged; entries no listed node touches keep
// their values.  d_work: 2 * n_local_dofs doubles of device scratch, ZEROED by the caller (sums and counts accumulate).  (A
// partitioned host exports the ghost rows of sum and count between l3k_values_at_nodes and l3k_average_values itself.)
template < typename KernelParamBlock >
void computeValuesAtNodes(const DeviceMesh& mesh, int residual_id, const KernelParamBlock* params, std::span< const int > dof_inds,
                          const CubeMesh::Sides* sides, const double* d_fields, size_t ldf, double time, int64_t n_local_dofs,
                          double* d_values, double* d_work)
{
    check(l3k_values_at_nodes(mesh.ctx(), mesh.get(), residual_id, params, params ? sizeof(KernelParamBlock) : 0, d_fields, ldf, time,
                              sides ? int64_t(sides->elems.size()) : -1, sides ? sides->elems.data() : nullptr,
                              sides ? sides->sides.data() : nullptr, dof_inds.data(), d_work, d_work + n_local_dofs));
    check(l3k_average_values(mesh.ctx(), d_work, d_work + n_local_dofs, n_local_dofs, d_values));
}

// NativeJacobiImpl::init (solve/NativePreconditioners.hpp:75-96): minv = sign(d) * damping / max(|d|, threshold)
inline void jacobiInverse(const DeviceMesh& mesh, const double* d_diag, int64_t n, double* d_minv, double damping = 1., double threshold = 0.)
{
    check(l3k_jacobi_inverse(mesh.ctx(), d_diag, n, damping, threshold, d_minv));
}

inline void updateSolution(const DeviceMesh& mesh, const double* d_x, size_t ldx, const double* d_xghost, size_t ldxg, int ncols,
                           std::span< const int > sol_inds, std::span< const int > sol_man_inds, double* d_fields, size_t ldf, int n_fields)
{
    if (sol_man_inds.size() != sol_inds.size() * size_t(ncols))
        throw std::runtime_error{"Source and destination indices lengths must match"};
    check(l3k_update_solution(mesh.ctx(), mesh.get(), d_x, ldx, d_xghost, ldxg, ncols, int(sol_inds.size()), sol_inds.data(),
                              sol_man_inds.data(), d_fields, ldf, n_fields));
}

class MatrixFreeSystem
{
public:
    template < typename KernelParamBlock >
    MatrixFreeSystem(const DeviceMesh& mesh, int kernel_id, const KernelParamBlock& params, AssemblyOptions opts = {},
                     std::span< const int > field_inds = {}, int n_rhs = 1)
    {
        const l3k_asmopts o{opts.value_order, opts.derivative_order, opts.eval_strategy};
        check(l3k_mf_create(mesh.ctx(), mesh.get(), kernel_id, &params, sizeof params, &o,
                            field_inds.empty() ? nullptr : field_inds.data(), n_rhs, &m_mf));
    }
    MatrixFreeSystem(const DeviceMesh& mesh, int kernel_id, AssemblyOptions opts = {}, int n_rhs = 1)
    {
        const l3k_asmopts o{opts.value_order, opts.derivative_order, opts.eval_strategy};
        check(l3k_mf_create(mesh.ctx(), mesh.get(), kernel_id, nullptr, 0, &o, nullptr, n_rhs, &m_mf));
    }
    MatrixFreeSystem(const MatrixFreeSystem&)            = delete;
    MatrixFreeSystem& operator=(const MatrixFreeSystem&) = delete;
    ~MatrixFreeSystem() { l3k_mf_destroy(m_mf); }

    void setFields(const double* d_soa, size_t ld) { check(l3k_mf_set_fields(m_mf, d_soa, ld)); } // post::FieldAccess
    void setTime(double t) { check(l3k_mf_set_time(m_mf, t)); }
    // assembleProblem(boundary kernel, boundary ids): the term takes part in apply / diagAndRhs from now on and must
    // outlive this system
    void assembleProblem(const BoundaryTerm& term) { check(l3k_mf_attach_boundary(m_mf, term.get())); }
    // Y <- alpha*A*X + beta*Y (Operator::apply; the operator is symmetric, `mode` is ignored by the reference too)
    void apply(const double* d_x, size_t ldx, double* d_y, size_t ldy, int ncols = 1, double alpha = 1., double beta = 0.) const
    {
        check(l3k_mf_apply(m_mf, d_x, ldx, d_y, ldy, ncols, alpha, beta));
    }
    // the same on the owned rows of a partitioned system: import || interior elements, border elements, export || interior
    // elements, unpack-add, Dirichlet rows (MatrixFreeSystem::applyImpl :1020-1140), the exchange through `halo`
    void apply(const Halo& halo, const double* d_x, size_t ldx, double* d_y, size_t ldy, int ncols = 1, double alpha = 1.,
               double beta = 0.) const
    {
        check(l3k_mf_apply_dist(m_mf, halo.get(), d_x, ldx, d_y, ldy, ncols, alpha, beta));
    }
    // scatterLocalSystem for the batch [first, first + count) of assembleLocal's output into the CSR values of the caller's
    // graph and the global right-hand sides (algsys/ScatterLocalSystem.hpp:24-54); returns the number of entries outside
    // the graph (skipped, as sumIntoLocalValues does)
    int64_t scatterLocalSystems(int64_t first, int64_t count, const double* d_K, const double* d_F, const int64_t* d_row_ptr,
                                const int32_t* d_col_ind, double* d_values, double* d_rhs, size_t ldr, bool skip_dirichlet = false) const
    {
        int64_t missing = 0;
        check(l3k_assembled_scatter(m_mf, first, count, d_K, d_F, d_row_ptr, d_col_ind, d_values, d_rhs, ldr, skip_dirichlet ? 1 : 0,
                                    &missing));
        return missing;
    }
    // assembleGlobalSystem (algsys/AssembleGlobalSystem.hpp:20-53) in one call: element systems formed and summed into the CSR
    // values / right-hand sides inside the library, sub-batch by sub-batch on two streams; returns the entries outside the graph
    int64_t assembleGlobal(int64_t first, int64_t count, const int64_t* d_row_ptr, const int32_t* d_col_ind, double* d_values,
                           double* d_rhs, size_t ldr, bool skip_dirichlet = false, size_t workspace_bytes = 0) const
    {
        int64_t missing = 0;
        check(l3k_assemble_global(m_mf, first, count, d_row_ptr, d_col_ind, d_values, d_rhs, ldr, skip_dirichlet ? 1 : 0, workspace_bytes,
                                  &missing));
        return missing;
    }
    // diag(A) and rhs with Dirichlet lifting; the caller zeroes d_diag / d_rhs first (computeDiagAndRhs :921-923)
    void diagAndRhs(const double* d_dirichlet_vals, size_t ldg, double* d_diag, double* d_rhs, size_t ldr) const
    {
        check(l3k_mf_diag_rhs(m_mf, 2, d_dirichlet_vals, ldg, d_diag, d_rhs, ldr, nullptr, nullptr, 0, 1));
    }
    // K_e (row-major Nd x Nd), F_e (column-major Nd x n_rhs) of elements [first, first+count)
    void assembleLocal(int64_t first, int64_t count, double* d_K, double* d_F, double* d_checksum = nullptr) const
    {
        check(l3k_local_assemble(m_mf, first, count, d_K, d_F, d_checksum));
    }
    // alg_sys.solve(CG{opts, NativeJacobiOpts{}}) for one rank (solve/BelosSolvers.hpp:116-122 + NativePreconditioners.hpp):
    // d_x holds the initial guess and the result; d_minv from l3k_jacobi_inverse (or nullptr); throws if not converged
    // like the reference (solve/BelosSolvers.hpp:103)
    l3k_cg_result solve(const double* d_b, double* d_x, const double* d_minv, l3k_cg_opts opts = {1e-6, 10000, 0, 1}) const
    {
        l3k_cg_result res{};
        check(l3k_pcg_solve(m_mf, d_b, d_x, d_minv, &opts, &res));
        if (!res.converged)
            throw std::runtime_error{"Solver failed to converge"};
        return res;
    }
    // the n_rhs columns of a multivector one after the other (the reference hands them to Belos "Block CG")
    std::vector< l3k_cg_result > solve(const double* d_b, size_t ldb, double* d_x, size_t ldx, int ncols, const double* d_minv,
                                       l3k_cg_opts opts = {1e-6, 10000, 0, 1}) const
    {
        std::vector< l3k_cg_result > res(static_cast< size_t >(ncols));
        check(l3k_pcg_solve_cols(m_mf, d_b, ldb, d_x, ldx, ncols, d_minv, &opts, res.data()));
        for (const auto& r : res)
            if (!r.converged)
                throw std::runtime_error{"Solver failed to converge"};
        return res;
    }
    // the kernel a launch of this system takes (which: 0 all / 1 interior / 2 border elements), as text
    std::string route(int which = 0, int ncols = 1, bool with_energy = false) const
    {
        char buf[512] = {};
        check(l3k_mf_route(m_mf, which, ncols, with_energy ? 1 : 0, buf, sizeof buf));
        return buf;
    }
    // Y <- A X and d_s[1] <- <X, A X> in one pass (what a CG iteration needs of the operator; d_s: 8 device doubles)
    void applyEnergy(const double* d_x, double* d_y, double* d_s) const { check(l3k_mf_apply_energy(m_mf, d_x, d_y, d_s)); }
    l3k_mf* get() const { return m_mf; }

private:
    l3k_mf* m_mf{};
};

// convertMeshToOrder< order >(mesh_o1) for one rank's hexahedra (mesh/ConvertMeshToOrder.hpp:51-104), on the device:
// conn = [n_elems][8] vertex ids, local vertex i + 2j + 4k.  Returns the element-node table [n_elems][(order+1)^3] in the
// numbering [vertices | edge nodes | face nodes | element-internal nodes]; n_nodes receives the node count.
inline std::vector< uint32_t > elevateOrder(Context& ctx, std::span< const uint32_t > conn, int64_t n_vertices, int order,
                                            int64_t& n_nodes)
{
    const int64_t           n_elems = int64_t(conn.size() / 8), N = int64_t(order + 1) * (order + 1) * (order + 1);
    std::vector< uint32_t > elem_nodes(size_t(n_elems * N));
    int64_t                 n_noninternal = 0;
    check(l3k_elevate_order(ctx.get(), n_elems, conn.data(), n_vertices, order, elem_nodes.data(), &n_nodes, &n_noninternal));
    return elem_nodes;
}

// save(comm, mesh, solution_manager, path, inds, comment) / Loader::loadResults of post/NativeIO.hpp for this rank's owned
// nodes [node_begin, node_begin + n_local): fields = [n_fields][ld] host values
inline void saveResults(const char* path, const char* comment, size_t n_fields, int64_t n_global_nodes, int64_t node_begin,
                        int64_t n_local, const double* fields, size_t ld, bool write_header = true)
{
    check(l3k_results_save(path, comment, n_fields, n_global_nodes, node_begin, n_local, fields, ld, write_header ? 1 : 0));
}
inline std::vector< double > loadResults(const char* path, size_t field, std::span< const int64_t > node_ids)
{
    std::vector< double > out(node_ids.size());
    check(l3k_results_load(path, field, int64_t(node_ids.size()), node_ids.data(), 0, out.data()));
    return out;
}

// save(comm, mesh, path, comment) / loadPartitionedMesh of post/NativeIO.hpp:75-108, :219-232 for this rank's part; the
// sizes of all parts are gathered by the caller (the reference: comm.gather at :83)
inline size_t meshFilePartBytes(const l3k_meshfile_part_desc& desc)
{
    size_t bytes = 0;
    check(l3k_meshfile_part_bytes(&desc, &bytes));
    return bytes;
}
inline void saveMesh(const char* path, const char* comment, std::span< const size_t > part_bytes, size_t part,
                     const l3k_meshfile_part_desc& desc)
{
    check(l3k_meshfile_save(path, comment, part_bytes.size(), part_bytes.data(), part, &desc, part == 0 ? 1 : 0));
}
class LoadedMeshPart
{
public:
    LoadedMeshPart(const char* path, size_t part, int order) { check(l3k_meshfile_load(path, part, order, &m_part)); }
    ~LoadedMeshPart() { l3k_meshfile_part_destroy(m_part); }
    LoadedMeshPart(const LoadedMeshPart&)            = delete;
    LoadedMeshPart& operator=(const LoadedMeshPart&) = delete;
    l3k_meshfile_part_desc get() const // the arrays live as long as this object
    {
        l3k_meshfile_part_desc d{};
        check(l3k_meshfile_part_get(m_part, &d));
        return d;
    }

private:
    l3k_meshfile_part* m_part = nullptr;
};
} // namespace l3k
#endif
