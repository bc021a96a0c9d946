// api_post.hip -- post-processing entry points of libl3k.so: integrals / L2 norms of residual kernels and values at nodes.
#include "objects.hpp"

using l3k::api::findResidual;

namespace
{
constexpr int reduce_threads = 256;
// out[v] = sum_b partial[b][v], fixed summation order (one workgroup per component)
__global__ __launch_bounds__(reduce_threads) void reducePartialsKernel(const double* __restrict__ partial, int64_t n_blocks,
                                                                         int nv, double* __restrict__ out)
{
    __shared__ double scratch[reduce_threads];
    const int         tid = threadIdx.x, v = blockIdx.x;
    double            s   = 0.;
    for (int64_t b = tid; b < n_blocks; b += reduce_threads)
        s += partial[b * nv + v];
    scratch[tid] = s;
    __syncthreads();
    for (int w = reduce_threads / 2; w > 0; w >>= 1)
    {
        if (tid < w)
            scratch[tid] += scratch[tid + w];
        __syncthreads();
    }
    if (tid == 0)
        out[v] = scratch[0];
}


// averageElementContributions (algsys/ComputeValuesAtNodes.hpp:112-154): entries nobody wrote keep their value
__global__ void averageValuesKernel(const double* __restrict__ sum, const double* __restrict__ count, int64_t n, double* __restrict__ values)
{
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
        if (count[i] > 0.)
            values[i] = sum[i] / count[i];
}
} // namespace

extern "C" {

// ------------------------------------------------------------------------------------------------ integrals
int l3k_residual_info(int residual_id, l3k_kparams* params, const char** name, size_t* param_bytes)
{
    const auto* k = findResidual(residual_id);
    if (!k)
    {
        setError("unknown residual kernel id %d", residual_id);
        return -1;
    }
    if (params)
        *params = k->kp;
    if (name)
        *name = k->name;
    if (param_bytes)
        *param_bytes = k->bytes;
    return 0;
}
int l3k_integrate(l3k_ctx* ctx, l3k_mesh* mesh, int residual_id, const void* kparam_blob, size_t kparam_bytes,
                  const l3k_asmopts* opts, const double* d_fields, size_t ldf, double time, int square, int64_t n_faces,
                  const int64_t* face_elem, const uint8_t* face_side, double* h_out)
{
    if (!ctx || !mesh || !h_out || (n_faces > 0 && (!face_elem || !face_side)))
    {
        setError("l3k_integrate: bad argument");
        return -1;
    }
    const auto* k = findResidual(residual_id);
    if (!k)
    {
        setError("unknown residual kernel id %d", residual_id);
        return -1;
    }
    if (k->kp.dimension != mesh->dim)
    {
        setError("kernel dimension %d != mesh dimension %d", k->kp.dimension, mesh->dim);
        return -1;
    }
    if (kparam_blob && kparam_bytes != k->bytes)
    {
        setError("kernel %s expects a %zu-byte parameter block, got %zu", k->name, k->bytes, kparam_bytes);
        return -1;
    }
    if (k->kp.n_fields > 0 && (!d_fields || ldf < size_t(mesh->n_owned_nodes + mesh->n_ghost_nodes)))
    {
        setError("kernel %s reads %d fields: pass them as SoA with ld >= number of local nodes", k->name, k->kp.n_fields);
        return -1;
    }
    const bool side = n_faces >= 0;
    for (int64_t i = 0; i < n_faces; ++i)
        if (face_elem[i] < 0 || face_elem[i] >= mesh->n_elems || face_side[i] >= 6)
        {
            setError("side %lld = (element %lld, side %d) is outside the mesh", (long long)i, (long long)face_elem[i],
                     int(face_side[i]));
            return -1;
        }
    const int E = k->kp.n_equations;
    for (int i = 0; i < E; ++i)
        h_out[i] = 0.;
    const int64_t count = side ? n_faces : mesh->n_elems;
    if (count == 0)
        return 0;
    const l3k_asmopts o  = opts ? *opts : l3k_asmopts{1, 0, 0};
    const int         nq = l3k_n_qps1d(mesh->order, o.value_order, o.derivative_order);
    if (nq < mesh->order + 1)
    {
        setError("nq = %d < p+1 = %d: the collocation-derivative device algorithm needs nq >= p+1", nq, mesh->order + 1);
        return -1;
    }
    const auto* inst = l3k::dev::findIntegralInstance(residual_id, mesh->order, nq);
    if (!inst)
    {
        setError("no device instantiation for residual kernel %d, order %d, nq %d: add it to "
                 "L3K_FOR_EACH_RESIDUAL_INSTANCE (l3ster_amd/csrc/user_kernels.hpp) and rebuild",
                 residual_id, mesh->order, nq);
        return -4;
    }
    L3K_HIP(hipSetDevice(ctx->device));
    hipStream_t        s = ctx->stream;
    DevBuf< double >   tables, partial;
    DevBuf< int64_t >  fe;
    DevBuf< uint8_t >  fs;
    const auto         block = l3k::host::deviceTableBlock(mesh->order, nq);
    if (int rc = tables.upload(block.data(), block.size(), s))
        return rc;
    if (side)
    {
        if (int rc = fe.upload(face_elem, size_t(n_faces), s))
            return rc;
        if (int rc = fs.upload(face_side, size_t(n_faces), s))
            return rc;
    }
    partial.n = size_t(count + 1) * E; // [count][E] partial sums + [E] result
    L3K_HIP(hipMalloc(reinterpret_cast< void** >(&partial.ptr), partial.n * sizeof(double)));
    l3k::dev::ElemArgs a{};
    a.elem_nodes = mesh->elem_nodes.ptr;
    a.elem_verts = mesh->elem_verts.ptr;
    a.tables     = tables.ptr;
    a.fields     = d_fields;
    a.ldf        = ldf;
    a.time       = time;
    a.elem_begin = 0, a.elem_count = mesh->n_elems;
    a.face_elem = fe.ptr, a.face_side = fs.ptr, a.face_begin = 0, a.face_count = side ? n_faces : 0;
    a.partial = partial.ptr;
    a.square  = square;
    if (int rc = (side ? inst->boundary : inst->domain)(a, kparam_blob, s))
        return rc;
    double* d_out = partial.ptr + size_t(count) * E;
    hipLaunchKernelGGL(reducePartialsKernel, dim3(E), dim3(reduce_threads), 0, s, partial.ptr, count, E,
                       d_out);
    L3K_HIP(hipGetLastError());
    L3K_HIP(hipMemcpyAsync(h_out, d_out, sizeof(double) * E, hipMemcpyDeviceToHost, s));
    L3K_HIP(hipStreamSynchronize(s));
    return 0;
}
// ------------------------------------------------------------------------------------------------ values at nodes
int l3k_values_at_nodes(l3k_ctx* ctx, l3k_mesh* mesh, int residual_id, const void* kparam_blob, size_t kparam_bytes,
                        const double* d_fields, size_t ldf, double time, int64_t n_faces, const int64_t* face_elem,
                        const uint8_t* face_side, const int* dof_inds, double* d_sum, double* d_count)
{
    if (!ctx || !mesh || !dof_inds || !d_sum || !d_count || (n_faces > 0 && (!face_elem || !face_side)))
    {
        setError("l3k_values_at_nodes: bad argument");
        return -1;
    }
    const auto* k = findResidual(residual_id);
    if (!k)
    {
        setError("unknown residual kernel id %d", residual_id);
        return -1;
    }
    if (k->kp.dimension != mesh->dim || k->kp.n_equations > l3k::dev::max_unknowns)
    {
        setError("kernel %s does not fit this mesh (dimension %d, %d equations)", k->name, k->kp.dimension, k->kp.n_equations);
        return -1;
    }
    if (kparam_blob && kparam_bytes != k->bytes)
    {
        setError("kernel %s expects a %zu-byte parameter block, got %zu", k->name, k->bytes, kparam_bytes);
        return -1;
    }
    if (k->kp.n_fields > 0 && (!d_fields || ldf < size_t(mesh->n_owned_nodes + mesh->n_ghost_nodes)))
    {
        setError("kernel %s reads %d fields: pass them as SoA with ld >= number of local nodes", k->name, k->kp.n_fields);
        return -1;
    }
    for (int e = 0; e < k->kp.n_equations; ++e)
        if (dof_inds[e] < 0 || dof_inds[e] >= mesh->dofs_per_node)
        {
            setError("dof_inds[%d] = %d outside [0, dofs_per_node = %d)", e, dof_inds[e], mesh->dofs_per_node);
            return -1;
        }
    for (int64_t i = 0; i < n_faces; ++i)
        if (face_elem[i] < 0 || face_elem[i] >= mesh->n_elems || face_side[i] >= 6)
        {
            setError("side %lld = (element %lld, side %d) is outside the mesh", (long long)i, (long long)face_elem[i],
                     int(face_side[i]));
            return -1;
        }
    const bool    side  = n_faces >= 0;
    const int64_t count = side ? n_faces : mesh->n_elems;
    if (count == 0)
        return 0;
    const auto* inst = l3k::dev::findIntegralInstance(residual_id, mesh->order, -1);
    if (!inst)
    {
        setError("no device instantiation for residual kernel %d, order %d: add it to L3K_FOR_EACH_RESIDUAL_INSTANCE "
                 "(l3ster_amd/csrc/user_kernels.hpp) and rebuild", residual_id, mesh->order);
        return -4;
    }
    L3K_HIP(hipSetDevice(ctx->device));
    hipStream_t       s = ctx->stream;
    DevBuf< double >  tables;
    DevBuf< int64_t > fe;
    DevBuf< uint8_t > fs;
    const auto        block = l3k::host::deviceTableBlock(mesh->order, inst->nq);
    if (int rc = tables.upload(block.data(), block.size(), s))
        return rc;
    if (side)
    {
        if (int rc = fe.upload(face_elem, size_t(n_faces), s))
            return rc;
        if (int rc = fs.upload(face_side, size_t(n_faces), s))
            return rc;
    }
    l3k::dev::ElemArgs a{};
    a.elem_nodes = mesh->elem_nodes.ptr;
    a.elem_verts = mesh->elem_verts.ptr;
    a.tables     = tables.ptr;
    a.fields     = d_fields;
    a.ldf        = ldf;
    a.time       = time;
    a.dofs_per_node = mesh->dofs_per_node;
    a.elem_begin = 0, a.elem_count = mesh->n_elems;
    a.face_elem = side ? fe.ptr : nullptr, a.face_side = fs.ptr, a.face_begin = 0, a.face_count = side ? n_faces : 0;
    for (int e = 0; e < k->kp.n_equations; ++e)
        a.field_inds[e] = dof_inds[e];
    a.node_sum   = d_sum;
    a.node_count = d_count;
    if (int rc = inst->at_nodes(a, kparam_blob, s))
        return rc;
    L3K_HIP(hipStreamSynchronize(s)); // the staging buffers are released on return
    return 0;
}
// MatrixFreeSystem::updateSolution (algsys/MatrixFreeSystem.hpp:1231-1273): solution dofs -> SolutionManager fields
__global__ void updateSolutionKernel(const double* __restrict__ x, size_t ldx, const double* __restrict__ xg, size_t ldxg, int64_t n_owned_nodes,
                                     int64_t n_local_nodes, int dpn, int ncols, int n_inds, const int* __restrict__ sol_inds,
                                     const int* __restrict__ dest, double* __restrict__ fields, size_t ldf)
{
    const int64_t total = n_local_nodes * n_inds * ncols;
    for (int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += int64_t(gridDim.x) * blockDim.x)
    {
        // node fastest: neighbouring threads write neighbouring entries of one field (the SoA storage), reads at a stride of dpn
        const int64_t node = t % n_local_nodes, ir = t / n_local_nodes;
        const int     i = int(ir / ncols), r = int(ir - int64_t(i) * ncols);
        const int64_t dof = node * dpn + sol_inds[i];
        const double  v   = node < n_owned_nodes ? x[dof + ldx * r] : xg[(dof - n_owned_nodes * dpn) + ldxg * r];
        fields[node + size_t(dest[i * ncols + r]) * ldf] = v;
    }
}

int l3k_update_solution(l3k_ctx* ctx, l3k_mesh* mesh, const double* d_x, size_t ldx, const double* d_xghost, size_t ldxg, int ncols,
                        int n_inds, const int* sol_inds, const int* sol_man_inds, double* d_fields, size_t ldf, int n_fields)
{
    if (!ctx || !mesh || !d_x || !d_fields || !sol_inds || !sol_man_inds || ncols < 1 || n_inds < 0 || n_fields < 0)
    {
        setError("l3k_update_solution: bad argument");
        return -1;
    }
    for (int i = 0; i < n_inds; ++i)
        if (sol_inds[i] < 0 || sol_inds[i] >= mesh->dofs_per_node)
        {
            setError("Source index out of bounds"); // MatrixFreeSystem.hpp:1239-1240
            return -1;
        }
    for (int i = 0; i < n_inds * ncols; ++i)
        if (sol_man_inds[i] < 0 || sol_man_inds[i] >= n_fields)
        {
            setError("Destination index out of bounds"); // :1241-1242
            return -1;
        }
    const int64_t n_local = mesh->n_owned_nodes + mesh->n_ghost_nodes;
    if (mesh->n_ghost_nodes > 0 && !d_xghost)
    {
        setError("l3k_update_solution: the mesh has ghost nodes: pass the imported ghost rows (l3k_halo_import)");
        return -1;
    }
    if (ldx < size_t(mesh->nOwnedDofs()) || ldf < size_t(n_local) || (d_xghost && ldxg < size_t(mesh->n_ghost_nodes * mesh->dofs_per_node)))
    {
        setError("l3k_update_solution: leading dimension too small");
        return -1;
    }
    if (n_inds == 0 || n_local == 0)
        return 0;
    L3K_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    int*        d_idx = nullptr; // [n_inds] source indices, then [n_inds * ncols] destination fields
    const size_t n_idx = size_t(n_inds) * (1 + ncols);
    std::vector< int > h(n_idx);
    std::copy_n(sol_inds, n_inds, h.begin());
    std::copy_n(sol_man_inds, size_t(n_inds) * ncols, h.begin() + n_inds);
    L3K_HIP(hipMalloc(reinterpret_cast< void** >(&d_idx), n_idx * sizeof(int)));
    if (hipMemcpyAsync(d_idx, h.data(), n_idx * sizeof(int), hipMemcpyHostToDevice, s) != hipSuccess)
    {
        (void)hipFree(d_idx);
        setError("l3k_update_solution: copy of the index lists failed");
        return -3;
    }
    hipLaunchKernelGGL(updateSolutionKernel, dim3(gridFor(n_local * n_inds * ncols)), dim3(256), 0, s, d_x, ldx, d_xghost, ldxg,
                       mesh->n_owned_nodes, n_local, mesh->dofs_per_node, ncols, n_inds, d_idx, d_idx + n_inds, d_fields, ldf);
    const hipError_t err = hipGetLastError();
    const hipError_t sy  = hipStreamSynchronize(s); // (the staged index lists are released on return)
    (void)hipFree(d_idx);
    if (err != hipSuccess || sy != hipSuccess)
    {
        setError("l3k_update_solution: kernel failed: %s", hipGetErrorString(err != hipSuccess ? err : sy));
        return -3;
    }
    return 0;
}

int l3k_average_values(l3k_ctx* ctx, const double* d_sum, const double* d_count, int64_t n, double* d_values)
{
    if (!ctx || (n > 0 && (!d_sum || !d_count || !d_values)))
    {
        setError("l3k_average_values: null argument");
        return -1;
    }
    if (n > 0)
        hipLaunchKernelGGL(averageValuesKernel, dim3(gridFor(n)), dim3(256), 0, ctx->stream, d_sum, d_count, n, d_values);
    L3K_HIP(hipGetLastError());
    return 0;
}

} // extern "C"
