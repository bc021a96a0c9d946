// api_assembled.hip -- scatter of local systems into the global system on the device: the assembled path's hand-off.
//
// Reference: algsys/ScatterLocalSystem.hpp:24-54 (per local row one CrsMatrix::sumIntoLocalValues(row, cols, vals) and an
// atomic add per right-hand side) called per element from algsys/AssembleGlobalSystem.hpp:20-53.  Here a whole batch of
// element matrices (the output of l3k_local_assemble, resident in HBM) is summed into the VALUES ARRAY of the caller's
// CSR graph (local row pointers / sorted local column indices, i.e. what Tpetra::CrsGraph::getLocalRowPtrsDevice /
// getLocalIndicesDevice hand out): the Tpetra side takes the finished values with one setAllValues, or one
// sumIntoLocalValues per row batch -- not one call per element row.
#include "objects.hpp"

namespace
{
struct ScatterArgs
{
    const uint32_t* elem_nodes;
    const uint8_t*  dirichlet; // per local dof, or null
    const double *  K, *F;
    const int64_t*  row_ptr;
    const int32_t*  col_ind;
    double *        values, *rhs;
    size_t          ldr;
    unsigned long long* n_missing;
    int64_t         first, count;
    int             NN, U, dpn, n_rhs, skip_dirichlet;
    int             field_inds[l3k::dev::max_unknowns];
};

// one wave per (element, local row): the row of K_e is read contiguously; every entry finds its position in the CSR row by
// binary search over the row's sorted column indices and is added atomically (elements sharing the row run concurrently)
__global__ __launch_bounds__(64) void assembledScatterKernel(const ScatterArgs a)
{
    const int     Nd   = a.NN * a.U;
    const int64_t e    = blockIdx.x / Nd; // element of the batch
    const int     i    = int(blockIdx.x - e * Nd);
    const uint32_t* en = a.elem_nodes + (a.first + e) * a.NN;
    const int64_t row  = int64_t(en[i / a.U]) * a.dpn + a.field_inds[i % a.U];
    if (a.skip_dirichlet && a.dirichlet && a.dirichlet[row])
        return;
    const int lane = threadIdx.x;
    if (a.F && a.rhs && lane < a.n_rhs)
        unsafeAtomicAdd(a.rhs + size_t(lane) * a.ldr + row, a.F[(e * a.n_rhs + lane) * Nd + i]);
    if (!a.K || !a.values)
        return;
    const int64_t  rb = a.row_ptr[row], re = a.row_ptr[row + 1];
    const double*  Kr = a.K + (e * Nd + i) * int64_t(Nd);
    unsigned       missing = 0;
    for (int j = lane; j < Nd; j += 64)
    {
        const int64_t col = int64_t(en[j / a.U]) * a.dpn + a.field_inds[j % a.U];
        if (a.skip_dirichlet && a.dirichlet && a.dirichlet[col])
            continue;
        int64_t lo = rb, hi = re; // first position with col_ind >= col
        while (lo < hi)
        {
            const int64_t mid = (lo + hi) >> 1;
            if (a.col_ind[mid] < col)
                lo = mid + 1;
            else
                hi = mid;
        }
        if (lo < re && a.col_ind[lo] == col)
            unsafeAtomicAdd(a.values + lo, Kr[j]);
        else
            ++missing; // (Tpetra's sumIntoLocalValues skips entries outside the graph and reports how many it took)
    }
    if (missing && a.n_missing)
        atomicAdd(a.n_missing, static_cast< unsigned long long >(missing));
}
} // namespace

extern "C" {
int l3k_assembled_scatter(l3k_mf* mf, int64_t first, int64_t count, const double* d_K, const double* d_F, const int64_t* d_row_ptr,
                          const int32_t* d_col_ind, double* d_values, double* d_rhs, size_t ldr, int skip_dirichlet,
                          int64_t* n_missing)
{
    if (!mf)
    {
        setError("null mf");
        return -1;
    }
    const l3k_mesh* m = mf->mesh;
    if (first < 0 || count < 0 || first + count > m->n_elems)
    {
        setError("element range [%lld, %lld) outside [0, %lld)", (long long)first, (long long)(first + count), (long long)m->n_elems);
        return -1;
    }
    if ((d_K != nullptr) != (d_values != nullptr) || (d_K && (!d_row_ptr || !d_col_ind)))
    {
        setError("l3k_assembled_scatter: K needs values, row_ptr and col_ind (and the other way round)");
        return -1;
    }
    if ((d_F != nullptr) != (d_rhs != nullptr))
    {
        setError("l3k_assembled_scatter: F and rhs go together");
        return -1;
    }
    const int64_t n_local_dofs = (m->n_owned_nodes + m->n_ghost_nodes) * m->dofs_per_node;
    if (d_rhs && ldr < size_t(n_local_dofs))
    {
        setError("rhs leading dimension smaller than the number of local dofs");
        return -1;
    }
    if (n_missing)
        *n_missing = 0;
    if (count == 0 || (!d_K && !d_F))
        return 0;
    L3K_HIP(hipSetDevice(mf->ctx->device));
    const int     N1 = m->order + 1, NN = N1 * N1 * N1, Nd = NN * mf->kp.n_unknowns;
    const int64_t blocks = count * Nd;
    if (blocks > int64_t(0x7fffffff))
    {
        setError("batch too large: %lld element rows in one launch", (long long)blocks);
        return -1;
    }
    hipStream_t         s       = mf->ctx->stream;
    unsigned long long* d_count = nullptr;
    if (n_missing)
    {
        L3K_HIP(hipMalloc(reinterpret_cast< void** >(&d_count), sizeof(unsigned long long)));
        L3K_HIP(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), s));
    }
    ScatterArgs a{};
    a.elem_nodes     = m->elem_nodes.ptr;
    a.dirichlet      = m->dirichlet.ptr;
    a.K              = d_K;
    a.F              = d_F;
    a.row_ptr        = d_row_ptr;
    a.col_ind        = d_col_ind;
    a.values         = d_values;
    a.rhs            = d_rhs;
    a.ldr            = ldr;
    a.n_missing      = d_count;
    a.first          = first;
    a.count          = count;
    a.NN             = NN;
    a.U              = mf->kp.n_unknowns;
    a.dpn            = m->dofs_per_node;
    a.n_rhs          = mf->n_rhs;
    a.skip_dirichlet = skip_dirichlet;
    for (int u = 0; u < l3k::dev::max_unknowns; ++u)
        a.field_inds[u] = mf->field_inds[u];
    hipLaunchKernelGGL(assembledScatterKernel, dim3(unsigned(blocks)), dim3(64), 0, s, a);
    L3K_HIP(hipGetLastError());
    if (n_missing)
    {
        unsigned long long h = 0;
        L3K_HIP(hipMemcpyAsync(&h, d_count, sizeof h, hipMemcpyDeviceToHost, s));
        L3K_HIP(hipStreamSynchronize(s));
        L3K_HIP(hipFree(d_count));
        *n_missing = int64_t(h);
    }
    return 0;
}
} // extern "C"
