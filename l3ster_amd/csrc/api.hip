// api.hip -- the extern "C" surface of libl3k.so (include/l3k.h) plus the small vector kernels around the element
// kernels (scale, Dirichlet rows, pack / unpack-add).  No CPU fallback: without a usable HIP device every device entry
// point fails with an error.
#include "objects.hpp"

#include <algorithm>
#include <cmath>
#include <unordered_map>

using l3k::api::KernelMeta;
using l3k::api::findKernel;
using l3k::api::findResidual;

// ------------------------------------------------------------------------------------------------ small kernels
namespace
{
// y <- beta*y (beta == 0: y <- 0, NaN-safe like putScalar(0.), algsys/MatrixFreeSystem.hpp:1038)
__global__ void scaleKernel(double* __restrict__ y, size_t ld, int64_t rows, int ncols, double beta)
{
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int c = 0; c < ncols; ++c)
    {
        double* col = y + ld * c;
        for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < rows; i += stride)
            col[i] = beta == 0. ? 0. : col[i] * beta;
    }
}
// y[d] += alpha*x[d] on owned Dirichlet rows (algsys/MatrixFreeSystem.hpp:1087-1098)
__global__ void dirichletRowsKernel(const int64_t* __restrict__ rows, int64_t n, const double* __restrict__ x, size_t ldx,
                                    double* __restrict__ y, size_t ldy, int ncols, double alpha)
{
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    {
        const int64_t d = rows[i];
        for (int c = 0; c < ncols; ++c)
            y[d + ldy * c] += alpha * x[d + ldx * c];
    }
}
// diag = 1, rhs = g on owned Dirichlet rows (algsys/MatrixFreeSystem.hpp:911-915)
__global__ void dirichletFinalizeKernel(const int64_t* __restrict__ rows, int64_t n, const double* __restrict__ g, size_t ldg,
                                        double* __restrict__ diag, double* __restrict__ rhs, size_t ldr, int ncols)
{
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    {
        const int64_t d = rows[i];
        diag[d]         = 1.;
        for (int c = 0; c < ncols; ++c)
            rhs[d + ldr * c] = g ? g[d + ldg * c] : 0.;
    }
}
// comm::Import pack / comm::Export unpack (comm/ImportExport.hpp:356-372, 448-470)
__global__ void packKernel(const double* __restrict__ src, size_t ld, int ncols, const int32_t* __restrict__ idx, int64_t n,
                           double* __restrict__ dst)
{
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    {
        const int64_t r = idx[i];
        for (int c = 0; c < ncols; ++c)
            dst[i + n * c] = src[r + ld * c];
    }
}
__global__ void unpackAddKernel(const double* __restrict__ src, int64_t n, const int32_t* __restrict__ idx,
                                double* __restrict__ dst, size_t ld, int ncols)
{
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    {
        const int64_t r = idx[i];
        for (int c = 0; c < ncols; ++c)
            dst[r + ld * c] += src[i + n * c];
    }
}
} // namespace

namespace
{
const std::vector< KernelMeta >& kernelMetas()
{
    static const std::vector< KernelMeta > metas = [] {
        std::vector< KernelMeta > m;
#define L3K_X(id, T, name)                                                                                             \
    m.push_back({id, {T::params.dimension, T::params.n_equations, T::params.n_unknowns, T::params.n_fields, T::params.n_rhs}, \
                 name, std::is_empty_v< T > ? size_t{0} : sizeof(T)});
        L3K_FOR_EACH_KERNEL(L3K_X)
#undef L3K_X
#define L3K_X(id, T, name)                                                                                             \
    m.push_back({id, {T::params.dimension, T::params.n_equations, T::params.n_unknowns, T::params.n_fields, T::params.n_rhs}, \
                 name, std::is_empty_v< T > ? size_t{0} : sizeof(T), true});
        L3K_FOR_EACH_BOUNDARY_KERNEL(L3K_X)
#undef L3K_X
        return m;
    }();
    return metas;
}
const std::vector< KernelMeta >& residualMetas()
{
    static const std::vector< KernelMeta > metas = [] {
        std::vector< KernelMeta > m;
#define L3K_X(id, T, name)                                                                                             \
    m.push_back({id, {T::params.dimension, T::params.n_equations, T::params.n_unknowns, T::params.n_fields, T::params.n_rhs}, \
                 name, std::is_empty_v< T > ? size_t{0} : sizeof(T)});
        L3K_FOR_EACH_RESIDUAL_KERNEL(L3K_X)
#undef L3K_X
        return m;
    }();
    return metas;
}
// kernels announced by plugins (l3k_plugin_load): metadata materialised on first sight
const KernelMeta* fromPlugin(int id, bool residual)
{
    struct Seen
    {
        KernelMeta meta;
        bool       residual;
    };
    static std::vector< std::unique_ptr< Seen > > seen;
    for (const auto& k : seen)
        if (k->meta.id == id && k->residual == residual)
            return &k->meta;
    const auto* pk = l3k::dev::findPluginKernel(id, residual);
    if (!pk)
        return nullptr;
    seen.push_back(std::make_unique< Seen >(Seen{KernelMeta{pk->id,
                                                            {pk->dimension, pk->n_equations, pk->n_unknowns, pk->n_fields, pk->n_rhs},
                                                            pk->name, pk->param_bytes, pk->kind == 1},
                                                 residual}));
    return &seen.back()->meta;
}
} // namespace

namespace l3k::api
{
const KernelMeta* findResidual(int id)
{
    for (const auto& k : residualMetas())
        if (k.id == id)
            return &k;
    return fromPlugin(id, true);
}
const KernelMeta* findKernel(int id)
{
    for (const auto& k : kernelMetas())
        if (k.id == id)
            return &k;
    return fromPlugin(id, false);
}
} // namespace l3k::api

#ifdef L3K_ABLATION
// stage timeline of the single-wave kernel (tools/kbench.py --stamps): 256 iterations x 16 cycle counters of workgroup 0
namespace
{
constexpr int n_stamps = 256 * 16 + 2 * 4096; // + start / end clock of every workgroup (up to 4096)
long long*    debugStamps()
{
    static long long* buf = [] {
        long long* p = nullptr;
        if (std::getenv("L3K_STAMPS") && hipMalloc(&p, n_stamps * sizeof(long long)) == hipSuccess)
            (void)hipMemset(p, 0, n_stamps * sizeof(long long));
        return p;
    }();
    return buf;
}
} // namespace
extern "C" int l3k_debug_stamps(long long* host, int n)
{
    long long* p = debugStamps();
    if (!p || n > n_stamps)
        return -1;
    return hipMemcpy(host, p, n * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}
#endif

namespace
{

// ElemArgs::scratch_alloc: the context's scratch arena, at least `bytes` large (kernels in flight on the context's stream may
// still use the old arena when it has to grow: hipFree waits for them)
double* contextScratch(void* owner, size_t bytes)
{
    auto* ctx = static_cast< l3k_ctx* >(owner);
    if (bytes <= ctx->scratch_bytes)
        return ctx->scratch;
    if (ctx->scratch)
        (void)hipFree(ctx->scratch);
    ctx->scratch       = nullptr;
    ctx->scratch_bytes = 0;
    if (hipMalloc(reinterpret_cast< void** >(&ctx->scratch), bytes) != hipSuccess)
        return nullptr;
    ctx->scratch_bytes = bytes;
    return ctx->scratch;
}

int fillArgs(l3k_mf* mf, int which, int ncols, l3k::dev::ElemArgs& a)
{
    const l3k_mesh* m = mf->mesh;
    a                 = {};
    a.elem_nodes      = m->elem_nodes.ptr;
    a.elem_verts      = m->elem_verts.ptr;
    a.dirichlet       = m->dirichlet.ptr;
    a.elem_flags      = m->elem_flags.ptr;
    a.exclusive_node_begin = m->exclusive_begin;
    a.exclusive_node_end   = m->exclusive_end;
    a.slot_tab             = m->slot_tab.ptr;
    a.all_affine           = m->all_affine ? 1 : 0;
    a.tune                 = &mf->ctx->tune;
    a.scratch_alloc        = &contextScratch;
    a.scratch_owner        = mf->ctx;
    // dynamic batch distribution of the single-wave kernel (l3k_tuning::static_deal: the static deal)
    a.work_counters        = mf->ctx->tune.static_deal ? nullptr : mf->ctx->work_counters;
    a.energy        = ncols == 1 ? mf->energy_target : nullptr;
    a.energy_done   = &mf->energy_done;
    a.n_shell              = m->n_shell;
    a.tables          = mf->tables.ptr;
    a.tables_host     = mf->tables_host.data();
    a.fields          = mf->fields;
    a.ldf             = mf->ldf;
    a.n_owned_dofs    = m->nOwnedDofs();
    a.time            = mf->time;
    a.dofs_per_node   = m->dofs_per_node;
#ifdef L3K_ABLATION // (tools/kbench.py's switches exist in the ablation build only)
    static const int dbg_flags = [] {
        const char* e = std::getenv("L3K_DEBUG_FLAGS");
        return e ? std::atoi(e) : 0;
    }();
    a.dbg    = dbg_flags;
    a.stamps = debugStamps();
#else
    a.dbg    = 0;
    a.stamps = nullptr;
#endif
    a.dense  = mf->dense;
    a.ref_z0 = mf->ctx->reference_z0 ? 1 : 0;
    for (int u = 0; u < l3k::dev::max_unknowns; ++u)
        a.field_inds[u] = mf->field_inds[u];
    switch (which)
    {
    case 0:
        a.elem_begin = 0;
        a.elem_count = m->n_interior;
        break;
    case 1:
        a.elem_begin = m->n_interior;
        a.elem_count = m->n_elems - m->n_interior;
        break;
    case 2:
        a.elem_begin = 0;
        a.elem_count = m->n_elems;
        break;
    case 3: // first / second half of the interior elements: one half overlaps the import, the other the export
        a.elem_begin = 0;
        a.elem_count = m->n_interior / 2;
        break;
    case 4:
        a.elem_begin = m->n_interior / 2;
        a.elem_count = m->n_interior - m->n_interior / 2;
        break;
    default:
        setError("which must be 0 (interior), 1 (border), 2 (all), 3 or 4 (first / second half of the interior)");
        return -1;
    }
    if (mf->kp.n_fields > 0 && !mf->fields)
    {
        setError("kernel reads %d external fields but l3k_mf_set_fields was not called", mf->kp.n_fields);
        return -1;
    }
    if (ncols < 1 || ncols > mf->n_rhs)
    {
        // algsys/MatrixFreeSystem.hpp:1035-1037
        setError("number of columns (%d) must be in [1, n_rhs = %d]", ncols, mf->n_rhs);
        return -1;
    }
    return 0;
}
const l3k::dev::Instance* instanceFor(const l3k_mf* mf, int ncols)
{
    const auto* inst = l3k::dev::findInstance(mf->kernel_id, mf->mesh->order, mf->nq, ncols);
    if (!inst)
        setError("no device instantiation for kernel %d, order %d, nq %d, ncols %d: add it to L3K_FOR_EACH_INSTANCE "
                 "(l3ster_amd/csrc/user_kernels.hpp) and rebuild",
                 mf->kernel_id, mf->mesh->order, mf->nq, ncols);
    return inst;
}
} // namespace

namespace
{
int bndArgs(const l3k_bnd* b, int which, int ncols, l3k::dev::ElemArgs& a)
{
    const l3k_mesh* m = b->mesh;
    a                 = {};
    a.elem_nodes      = m->elem_nodes.ptr;
    a.elem_verts      = m->elem_verts.ptr;
    a.dirichlet       = m->dirichlet.ptr;
    a.tables          = b->tables.ptr;
    a.fields          = b->fields;
    a.ldf             = b->ldf;
    a.n_owned_dofs    = m->nOwnedDofs();
    a.time            = b->time;
    a.dofs_per_node   = m->dofs_per_node;
    a.face_elem       = b->face_elem.ptr;
    a.face_side       = b->face_side.ptr;
    for (int u = 0; u < l3k::dev::max_unknowns; ++u)
        a.field_inds[u] = b->field_inds[u];
    switch (which)
    {
    case 0:
        a.face_begin = 0;
        a.face_count = b->n_interior_faces;
        break;
    case 1:
        a.face_begin = b->n_interior_faces;
        a.face_count = b->n_faces - b->n_interior_faces;
        break;
    case 2:
        a.face_begin = 0;
        a.face_count = b->n_faces;
        break;
    default:
        setError("which must be 0 (sides of interior elements), 1 (of border elements) or 2 (all)");
        return -1;
    }
    if (b->kp.n_fields > 0 && !b->fields)
    {
        setError("boundary kernel reads %d external fields but l3k_bnd_set_fields was not called", b->kp.n_fields);
        return -1;
    }
    if (ncols < 1 || ncols > b->n_rhs)
    {
        setError("number of columns (%d) must be in [1, n_rhs = %d]", ncols, b->n_rhs);
        return -1;
    }
    return 0;
}
// the launch ranges of a side call: the caller's range, or in deterministic mode one launch per colour of the classes it
// covers (the ELEMENTS of the sides of one colour share no node: the order of the additions to a row is the order of the launches)
template < typename F >
int forEachSideRange(const l3k_bnd* b, int which, l3k::dev::ElemArgs& a, F&& launch)
{
    if (!b->ctx->deterministic)
        return launch(a);
    if (b->det_ptr[0].empty())
    {
        setError("deterministic mode was enabled after this boundary term was created: create it with the mode on");
        return -1;
    }
    const bool classes[2] = {which == 0 || which == 2, which == 1 || which == 2};
    for (int cls = 0; cls < 2; ++cls)
        if (classes[cls])
            for (size_t c = 0; c + 1 < b->det_ptr[cls].size(); ++c)
            {
                a.face_begin = b->det_ptr[cls][c];
                a.face_count = b->det_ptr[cls][c + 1] - b->det_ptr[cls][c];
                if (a.face_count > 0)
                    if (int rc = launch(a))
                        return rc;
            }
    return 0;
}
const l3k::dev::BoundaryInstance* bndInstance(const l3k_bnd* b, int ncols)
{
    const auto* inst = l3k::dev::findBoundaryInstance(b->kernel_id, b->mesh->order, b->nq, ncols);
    if (!inst)
        setError("no device instantiation for boundary kernel %d, order %d, nq %d, ncols %d: add it to "
                 "L3K_FOR_EACH_BOUNDARY_INSTANCE (l3ster_amd/csrc/user_kernels.hpp) and rebuild",
                 b->kernel_id, b->mesh->order, b->nq, ncols);
    return inst;
}
int bndApplyImpl(l3k_bnd* b, int which, const double* d_x, size_t ldx, const double* d_xghost, size_t ldxg, double* d_y,
                 size_t ldy, double* d_yghost, size_t ldyg, int ncols, double alpha)
{
    l3k::dev::ElemArgs a;
    if (int rc = bndArgs(b, which, ncols, a))
        return rc;
    const l3k_mesh* m = b->mesh;
    if (ldx < size_t(m->nOwnedDofs()) || ldy < size_t(m->nOwnedDofs()))
    {
        setError("leading dimension smaller than the number of owned dofs");
        return -1;
    }
    if (m->n_ghost_nodes > 0 && which != 0 && (!d_xghost || !d_yghost))
    {
        setError("mesh has ghost nodes: sides of border elements need the ghost import/export buffers");
        return -1;
    }
    a.x = d_x, a.xg = d_xghost, a.y = d_y, a.yg = d_yghost;
    a.ldx = ldx, a.ldxg = ldxg, a.ldy = ldy, a.ldyg = ldyg;
    a.alpha = alpha;
    const auto* inst = bndInstance(b, ncols);
    if (!inst)
        return -4;
    return forEachSideRange(b, which, a, [&](l3k::dev::ElemArgs& r) { return inst->apply(r, b->blob.empty() ? nullptr : b->blob.data(), b->ctx->stream); });
}
int bndDiagRhsImpl(l3k_bnd* b, int which, const double* d_dirichlet_vals, size_t ldg, double* d_diag, double* d_rhs,
                   size_t ldr, double* d_diag_ghost, double* d_rhs_ghost, size_t ldrg)
{
    l3k::dev::ElemArgs a;
    if (int rc = bndArgs(b, which, b->n_rhs, a))
        return rc;
    if (b->mesh->n_ghost_nodes > 0 && which != 0 && (!d_rhs_ghost || (d_diag && !d_diag_ghost)))
    {
        setError("mesh has ghost nodes: sides of border elements need the ghost diag / rhs buffers");
        return -1;
    }
    a.dirichlet_vals = d_dirichlet_vals;
    a.ldg            = ldg;
    a.y = d_rhs, a.ldy = ldr, a.yg = d_rhs_ghost, a.ldyg = ldrg;
    a.diag = d_diag, a.diag_g = d_diag_ghost;
    const auto* inst = bndInstance(b, b->n_rhs);
    if (!inst)
        return -4;
    return forEachSideRange(b, which, a, [&](l3k::dev::ElemArgs& r) { return inst->diag_rhs(r, b->blob.empty() ? nullptr : b->blob.data(), b->ctx->stream); });
}
} // namespace

// ------------------------------------------------------------------------------------------------ C ABI

namespace
{
// Greedy colouring of the elements by their corner nodes (two conforming hexes share a node iff they share a corner), class
// by class in element order; the element arrays are copied in (class, colour, element) order.  A launch over one colour of
// one class adds at most once to any row, so the order of the additions to a row is the order of the launches.
int buildDeterministicPlan(l3k_mesh& m, const l3k_mesh_desc* d, const std::vector< uint8_t >& flags)
{
    const int     n1 = d->order + 1;
    const int64_t N  = int64_t(n1) * n1 * n1;
    int           corner[8];
    for (int v = 0; v < 8; ++v)
        corner[v] = ((v & 1) ? n1 - 1 : 0) + n1 * (((v >> 1) & 1) ? n1 - 1 : 0) + n1 * n1 * (((v >> 2) & 1) ? n1 - 1 : 0);
    std::unordered_map< uint32_t, uint64_t > used; // corner node -> colours taken by the elements around it
    used.reserve(static_cast< size_t >(d->n_elems) * 2);
    std::vector< uint8_t > colour(static_cast< size_t >(d->n_elems));
    int                    n_colours = 0;
    for (int64_t e = 0; e < d->n_elems; ++e)
    {
        uint64_t taken = 0;
        for (int v = 0; v < 8; ++v)
        {
            const auto it = used.find(d->elem_nodes[e * N + corner[v]]);
            if (it != used.end())
                taken |= it->second;
        }
        int c = 0;
        while (c < 64 && ((taken >> c) & 1u))
            ++c;
        if (c == 64)
        {
            setError("deterministic mode: more than 64 colours needed");
            return -1;
        }
        colour[e] = uint8_t(c);
        n_colours = std::max(n_colours, c + 1);
        for (int v = 0; v < 8; ++v)
            used[d->elem_nodes[e * N + corner[v]]] |= uint64_t(1) << c;
    }
    m.det_corner_nodes.resize(static_cast< size_t >(d->n_elems) * 8);
    for (int64_t e = 0; e < d->n_elems; ++e)
        for (int v = 0; v < 8; ++v)
            m.det_corner_nodes[size_t(e) * 8 + v] = d->elem_nodes[e * N + corner[v]];
    std::vector< int64_t > order(static_cast< size_t >(d->n_elems));
    for (int64_t e = 0; e < d->n_elems; ++e)
        order[e] = e;
    const int64_t n_int = d->n_interior_elems;
    auto          by_colour = [&](int64_t x, int64_t y) { return colour[x] < colour[y]; };
    std::stable_sort(order.begin(), order.begin() + n_int, by_colour);
    std::stable_sort(order.begin() + n_int, order.end(), by_colour);
    for (int cls = 0; cls < 2; ++cls)
    {
        const int64_t b = cls ? n_int : 0, e = cls ? d->n_elems : n_int;
        m.det_ptr[cls].assign(static_cast< size_t >(n_colours) + 1, e);
        int64_t i = b;
        for (int c = 0; c < n_colours; ++c)
        {
            m.det_ptr[cls][c] = i;
            while (i < e && colour[order[i]] == c)
                ++i;
        }
    }
    std::vector< uint32_t > nodes(static_cast< size_t >(d->n_elems * N));
    std::vector< double >   verts(static_cast< size_t >(d->n_elems) * 24);
    std::vector< uint8_t >  fl(flags.empty() ? 0 : static_cast< size_t >(d->n_elems));
    for (int64_t i = 0; i < d->n_elems; ++i)
    {
        const int64_t e = order[i];
        std::copy_n(d->elem_nodes + e * N, N, nodes.begin() + i * N);
        std::copy_n(d->elem_verts + e * 24, 24, verts.begin() + i * 24);
        if (!fl.empty())
            fl[i] = flags[e];
    }
    hipStream_t s = m.ctx->stream;
    if (int rc = m.det_elem_nodes.upload(nodes.data(), nodes.size(), s))
        return rc;
    if (int rc = m.det_elem_verts.upload(verts.data(), verts.size(), s))
        return rc;
    if (!fl.empty())
        if (int rc = m.det_elem_flags.upload(fl.data(), fl.size(), s))
            return rc;
    L3K_HIP(hipStreamSynchronize(s)); // (the staging vectors are locals)
    m.det_built = true;
    return 0;
}
// the launch ranges of an element call: the caller's range as it is, or in deterministic mode the colours of the classes
// it covers, on the permuted element arrays (the two halves of the interior, which = 3 / 4, become: all of it / nothing)
template < typename F >
int forEachLaunchRange(const l3k_mf* mf, int which, l3k::dev::ElemArgs& a, F&& launch)
{
    const l3k_mesh* m = mf->mesh;
    if (!mf->ctx->deterministic)
        return launch(a);
    if (!m->det_built)
    {
        setError("deterministic mode was enabled after this mesh was created: create the mesh with the mode on");
        return -1;
    }
    a.elem_nodes = m->det_elem_nodes.ptr;
    a.elem_verts = m->det_elem_verts.ptr;
    a.elem_flags = m->det_elem_flags.ptr;
    a.energy     = nullptr; // (the fused <x, A x> is an atomic accumulation: the caller takes the fixed-order dot product)
    const bool classes[2] = {which == 0 || which == 2 || which == 3, which == 1 || which == 2};
    for (int cls = 0; cls < 2; ++cls)
        if (classes[cls])
            for (size_t c = 0; c + 1 < m->det_ptr[cls].size(); ++c)
            {
                a.elem_begin = m->det_ptr[cls][c];
                a.elem_count = m->det_ptr[cls][c + 1] - m->det_ptr[cls][c];
                if (a.elem_count > 0)
                    if (int rc = launch(a))
                        return rc;
            }
    return 0;
}
} // namespace

extern "C" {

int l3k_version(void)
{
    return L3K_VERSION;
}
const char* l3k_last_error(void)
{
    return l3k::dev::lastError();
}

int l3k_gll_nodes(int n, double* x)
{
    if (n < 2 || !x)
    {
        setError("l3k_gll_nodes: need n >= 2");
        return -1;
    }
    const auto v = l3k::host::gllNodes(n);
    std::copy(v.begin(), v.end(), x);
    return 0;
}
int l3k_gl_rule(int nq, double* x, double* w)
{
    if (nq < 1 || !x || !w)
    {
        setError("l3k_gl_rule: need nq >= 1");
        return -1;
    }
    std::vector< double > xv, wv;
    l3k::host::glRule(nq, xv, wv);
    std::copy(xv.begin(), xv.end(), x);
    std::copy(wv.begin(), wv.end(), w);
    return 0;
}
int l3k_n_qps1d(int p, int value_order, int derivative_order)
{
    return value_order * p + derivative_order * (p - 1) + 1;
}
int l3k_basis_1d(int p, int nq, double* I, double* D)
{
    if (p < 1 || nq < 1 || !I || !D)
    {
        setError("l3k_basis_1d: bad arguments");
        return -1;
    }
    std::vector< double > Iv, Dv;
    l3k::host::basis1d(p, nq, Iv, Dv);
    std::copy(Iv.begin(), Iv.end(), I);
    std::copy(Dv.begin(), Dv.end(), D);
    return 0;
}
int l3k_colloc_deriv(int nq, double* C)
{
    if (nq < 1 || !C)
    {
        setError("l3k_colloc_deriv: bad arguments");
        return -1;
    }
    const auto v = l3k::host::collocDeriv(nq);
    std::copy(v.begin(), v.end(), C);
    return 0;
}

int l3k_kernel_info(int kernel_id, l3k_kparams* params, const char** name, size_t* param_bytes)
{
    const auto* k = findKernel(kernel_id);
    if (!k)
    {
        setError("unknown kernel id %d", kernel_id);
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
int l3k_plugin_load(const char* path)
{
    if (!path)
    {
        setError("l3k_plugin_load: null path");
        return -1;
    }
    // the plugin's static registrars call registerPluginKernel / registerInstance / ... of THIS library
    void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h)
    {
        setError("l3k_plugin_load: %s", dlerror());
        return -1;
    }
    return 0;
}
int l3k_instance_count(void)
{
    return l3k::dev::instanceCount();
}
int l3k_instance_info(int i, int* kernel_id, int* order, int* nq, int* ncols)
{
    const auto* inst = l3k::dev::instanceAt(i);
    if (!inst)
    {
        setError("instance index %d out of range", i);
        return -1;
    }
    *kernel_id = inst->kernel_id;
    *order     = inst->order;
    *nq        = inst->nq;
    *ncols     = inst->ncols;
    return 0;
}

int l3k_ctx_create(int hip_device, void* hip_stream, l3k_ctx** out)
{
    if (!out)
    {
        setError("l3k_ctx_create: out is null");
        return -1;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1)
    {
        setError("no HIP device available: libl3k has no CPU fallback (the device path is the product)");
        return -2;
    }
    if (hip_device < 0 || hip_device >= count)
    {
        setError("hip_device %d outside [0,%d)", hip_device, count);
        return -1;
    }
    L3K_HIP(hipSetDevice(hip_device));
    auto* ctx = new l3k_ctx{hip_device, static_cast< hipStream_t >(hip_stream)};
    // the ONLY place the product reads the environment: initial values of the context's settings (include/l3k.h: l3k_tuning)
    if (const char* e = std::getenv("L3K_DETERMINISTIC"))
        ctx->deterministic = std::atoi(e) != 0;
    const auto flag = [](const char* name, int& field) {
        if (const char* e = std::getenv(name))
            field = *e != '\0' && std::strcmp(e, "0") != 0;
    };
    if (const char* e = std::getenv("L3K_GENERIC_BELOW"))
        ctx->tune.generic_below = std::atoll(e);
    if (const char* e = std::getenv("L3K_FAST_WAVES_PER_CU"))
        ctx->tune.waves_per_cu = std::atoi(e) > 0 ? std::atoi(e) : 0;
    flag("L3K_FAST_STATIC", ctx->tune.static_deal);
    flag("L3K_NO_AFFINE", ctx->tune.no_affine);
    flag("L3K_COLUMN_BY_COLUMN", ctx->tune.column_by_column);
    flag("L3K_ASSEMBLE_DENSE", ctx->tune.assemble_dense);
    flag("L3K_ASM_TWO_LAUNCHES", ctx->tune.assemble_two_launches);
    flag("L3K_SCATTER_PER_ENTRY", ctx->tune.scatter_per_entry);
    flag("L3K_ASM_DIRECT_STORE", ctx->tune.assemble_direct_store);
    flag("L3K_ASM_NO_SYMMETRISE", ctx->tune.assemble_no_symmetrise);
    // the context's own device buffers are allocated here, with its device current: a later call may come from a thread
    // whose current device is another one (several contexts in one process: thread-emulated ranks, a multi-GPU C++ host)
    if (hipMalloc(reinterpret_cast< void** >(&ctx->work_counters), 9 * 128) != hipSuccess ||
        hipMalloc(reinterpret_cast< void** >(&ctx->red_ws), sizeof(double) * 2 * l3k_cg_blocks) != hipSuccess)
    {
        delete ctx;
        setError("l3k_ctx_create: hipMalloc of the context buffers failed");
        return -3;
    }
    *out = ctx;
    return 0;
}
int l3k_ctx_set_stream(l3k_ctx* ctx, void* hip_stream)
{
    if (!ctx)
    {
        setError("null ctx");
        return -1;
    }
    ctx->stream = static_cast< hipStream_t >(hip_stream);
    return 0;
}
int l3k_ctx_set_deterministic(l3k_ctx* ctx, int on)
{
    if (!ctx)
    {
        setError("null ctx");
        return -1;
    }
    ctx->deterministic = on != 0;
    return 0;
}
int l3k_ctx_set_reference_z0(l3k_ctx* ctx, int on)
{
    if (!ctx)
    {
        setError("null context");
        return -1;
    }
    ctx->reference_z0 = on != 0;
    return 0;
}
int l3k_ctx_get_tuning(const l3k_ctx* ctx, l3k_tuning* out)
{
    if (!ctx || !out)
    {
        setError("l3k_ctx_get_tuning: null argument");
        return -1;
    }
    *out = ctx->tune;
    return 0;
}
int l3k_ctx_set_tuning(l3k_ctx* ctx, const l3k_tuning* in)
{
    if (!ctx || !in)
    {
        setError("l3k_ctx_set_tuning: null argument");
        return -1;
    }
    if (in->generic_below < 0 || in->waves_per_cu < 0 || in->waves_per_cu > 64)
    {
        setError("l3k_ctx_set_tuning: generic_below must be >= 0 and waves_per_cu in [0, 64]");
        return -1;
    }
    ctx->tune = *in;
    return 0;
}
int l3k_ctx_synchronize(l3k_ctx* ctx)
{
    if (!ctx)
    {
        setError("null ctx");
        return -1;
    }
    L3K_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}
int l3k_ctx_destroy(l3k_ctx* ctx)
{
    delete ctx;
    return 0;
}

int l3k_mesh_create(l3k_ctx* ctx, const l3k_mesh_desc* d, l3k_mesh** out)
{
    if (!ctx || !d || !out)
    {
        setError("l3k_mesh_create: null argument");
        return -1;
    }
    if (d->dim != 3)
    {
        setError("device kernels exist for hex elements only (dim = 3); quads are covered by the CPU oracle");
        return -1;
    }
    if (d->order < 1 || d->n_elems < 0 || d->n_interior_elems < 0 || d->n_interior_elems > d->n_elems ||
        d->dofs_per_node < 1 || d->n_owned_nodes < 0 || d->n_ghost_nodes < 0 || !d->elem_nodes || !d->elem_verts)
    {
        setError("l3k_mesh_create: inconsistent descriptor");
        return -1;
    }
    const int64_t N       = int64_t(d->order + 1) * (d->order + 1) * (d->order + 1);
    const int64_t n_nodes = d->n_owned_nodes + d->n_ghost_nodes;
    if (n_nodes * d->dofs_per_node >= (int64_t(1) << 31) * 8)
    {
        setError("too many local dofs");
        return -1;
    }
    // operand shapes must match what the kernels assume: every node id inside [0, n_nodes)
    for (int64_t i = 0; i < d->n_elems * N; ++i)
        if (d->elem_nodes[i] >= static_cast< uint64_t >(n_nodes))
        {
            setError("elem_nodes[%lld] = %u is outside the local node range [0,%lld)", (long long)i, d->elem_nodes[i],
                     (long long)n_nodes);
            return -1;
        }
    L3K_HIP(hipSetDevice(ctx->device));
    auto m           = std::make_unique< l3k_mesh >();
    m->ctx           = ctx;
    m->dim           = d->dim;
    m->order         = d->order;
    m->dofs_per_node = d->dofs_per_node;
    m->n_elems       = d->n_elems;
    m->n_interior    = d->n_interior_elems;
    m->n_owned_nodes = d->n_owned_nodes;
    m->n_ghost_nodes = d->n_ghost_nodes;
    if (int rc = m->elem_nodes.upload(d->elem_nodes, size_t(d->n_elems * N), ctx->stream))
        return rc;
    if (int rc = m->elem_verts.upload(d->elem_verts, size_t(d->n_elems) * 24, ctx->stream))
        return rc;
    // nodes referenced by exactly one element (the element-internal nodes of the reference's numbering,
    // mesh/LocalMeshView.hpp:425-458) can be scattered with plain stores: find the maximal owned tail range
    {
        std::vector< uint8_t > count(static_cast< size_t >(n_nodes), 0);
        for (int64_t i = 0; i < d->n_elems * N; ++i)
            if (count[d->elem_nodes[i]] < 2)
                ++count[d->elem_nodes[i]];
        int64_t b = d->n_owned_nodes;
        while (b > 0 && count[b - 1] == 1) // exactly one: a node no element touches still needs the beta scaling
            --b;
        m->exclusive_begin = b;
        m->exclusive_end   = d->n_owned_nodes;
        // The single-wave kernel decides "exclusive" by the LOCAL position (element-internal or not) instead of testing
        // every node id: shrink the range until it holds no node that sits on an element's shell, and drop it altogether
        // if some element-internal node lies outside it (numberings that do not put the internal nodes last).
        const int n1 = d->order + 1;
        auto      internal = [&](int64_t i) {
            const int ix = int(i % n1), iy = int((i / n1) % n1), iz = int(i / (n1 * n1));
            return ix > 0 && ix < n1 - 1 && iy > 0 && iy < n1 - 1 && iz > 0 && iz < n1 - 1;
        };
        int64_t max_shell = -1, min_internal = n_nodes;
        for (int64_t e = 0; e < d->n_elems; ++e)
            for (int64_t i = 0; i < N; ++i)
            {
                const int64_t id = d->elem_nodes[e * N + i];
                if (internal(i))
                    min_internal = std::min(min_internal, id);
                else if (id < d->n_owned_nodes)
                    max_shell = std::max(max_shell, id);
            }
        m->exclusive_begin = std::max(m->exclusive_begin, max_shell + 1);
        if (d->n_elems == 0 || d->order < 2 || min_internal < m->exclusive_begin)
            m->exclusive_begin = m->exclusive_end; // empty: every node is scattered with atomics
        // scatter slots: local nodes of a typical element (the middle one of the traversal) in ascending id order --
        // rows that are contiguous in y become contiguous slots; internal positions last
        if (d->n_elems > 0 && n1 <= 8)
        {
            const uint32_t*        ids = d->elem_nodes + (d->n_elems / 2) * N;
            std::vector< int32_t > order(static_cast< size_t >(N));
            for (int64_t i = 0; i < N; ++i)
                order[i] = int32_t(i);
            std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
                const bool ix = internal(x), iy = internal(y);
                return ix != iy ? iy : ids[x] < ids[y];
            });
            std::vector< uint16_t > tab(static_cast< size_t >(n1 * n1 * 8), 0);
            int                     n_shell = 0;
            for (int64_t s = 0; s < N; ++s)
            {
                const int32_t i                           = order[s];
                tab[(i % (n1 * n1)) * 8 + i / (n1 * n1)] = uint16_t(s);
                n_shell += !internal(i);
            }
            m->n_shell = n_shell;
            if (int rc = m->slot_tab.upload(tab.data(), tab.size(), ctx->stream))
                return rc;
            L3K_HIP(hipStreamSynchronize(ctx->stream)); // (tab is a local)
        }
    }
    std::vector< int64_t > rows;
    // per-element flags: bit 0 = the element touches a Dirichlet dof, bit 1 = its tri-linear map is affine (a
    // parallelepiped: the bilinear and trilinear coefficient vectors of the 8 vertices vanish), so its Jacobian is one
    // matrix -- the element kernel then inverts it once per element instead of once per quadrature point
    std::vector< uint8_t > flags(static_cast< size_t >(d->n_elems), 0); // (host staging: outlives the synchronisation below)
    for (int64_t e = 0; e < d->n_elems; ++e)
    {
        const double* v = d->elem_verts + e * 24;
        double        scale = 0., dev = 0.;
        for (int s = 0; s < 3; ++s)
        {
            auto c = [&](int sx, int sy, int sz) { // coefficient of xi^sx eta^sy zeta^sz (x 8)
                double acc = 0.;
                for (int k = 0; k < 8; ++k)
                    acc += ((sx && !(k & 1)) ? -1. : 1.) * ((sy && !(k & 2)) ? -1. : 1.) * ((sz && !(k & 4)) ? -1. : 1.) * v[k * 3 + s];
                return acc;
            };
            scale = std::max({scale, std::fabs(c(1, 0, 0)), std::fabs(c(0, 1, 0)), std::fabs(c(0, 0, 1))});
            dev   = std::max({dev, std::fabs(c(1, 1, 0)), std::fabs(c(1, 0, 1)), std::fabs(c(0, 1, 1)), std::fabs(c(1, 1, 1))});
        }
        if (dev <= 1e-14 * scale)
            flags[e] |= 2;
    }
    m->all_affine = d->n_elems > 0;
    for (uint8_t f : flags)
        m->all_affine = m->all_affine && (f & 2);
    if (d->dirichlet)
    {
        for (int64_t e = 0; e < d->n_elems; ++e)
            for (int64_t i = 0; i < N && !(flags[e] & 1); ++i)
                for (int k = 0; k < d->dofs_per_node; ++k)
                    if (d->dirichlet[int64_t(d->elem_nodes[e * N + i]) * d->dofs_per_node + k])
                    {
                        flags[e] |= 1;
                        break;
                    }
        if (int rc = m->dirichlet.upload(d->dirichlet, size_t(n_nodes * d->dofs_per_node), ctx->stream))
            return rc;
        for (int64_t i = 0; i < d->n_owned_nodes * d->dofs_per_node; ++i) // getOwnedDirichletDofs
            if (d->dirichlet[i])
                rows.push_back(i);
        if (int rc = m->owned_dirichlet_rows.upload(rows.data(), rows.size(), ctx->stream))
            return rc;
    }
    if (int rc = m->elem_flags.upload(flags.data(), flags.size(), ctx->stream))
        return rc;
    if (ctx->deterministic)
        if (int rc = buildDeterministicPlan(*m, d, flags))
            return rc;
    L3K_HIP(hipStreamSynchronize(ctx->stream)); // host arrays may be freed by the caller after return
    *out = m.release();
    return 0;
}
int l3k_mesh_destroy(l3k_mesh* mesh)
{
    delete mesh;
    return 0;
}

int l3k_mf_create(l3k_ctx* ctx, l3k_mesh* mesh, int kernel_id, const void* kparam_blob, size_t kparam_bytes,
                  const l3k_asmopts* opts, const int* field_inds, int n_rhs, l3k_mf** out)
{
    if (!ctx || !mesh || !out)
    {
        setError("l3k_mf_create: null argument");
        return -1;
    }
    const auto* k = findKernel(kernel_id);
    if (!k)
    {
        setError("unknown kernel id %d", kernel_id);
        return -1;
    }
    if (k->boundary)
    {
        setError("kernel %s is a boundary equation kernel: use l3k_bnd_create", k->name);
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
    if (n_rhs < 1)
    {
        setError("n_rhs must be >= 1");
        return -1;
    }
    const l3k_asmopts o  = opts ? *opts : l3k_asmopts{1, 0, 0};
    const int         nq = l3k_n_qps1d(mesh->order, o.value_order, o.derivative_order);
    if (nq < mesh->order + 1)
    {
        setError("nq = %d < p+1 = %d: the collocation-derivative device algorithm needs nq >= p+1", nq, mesh->order + 1);
        return -1;
    }
    auto mf       = std::make_unique< l3k_mf >();
    mf->ctx       = ctx;
    mf->mesh      = mesh;
    mf->kernel_id = kernel_id;
    mf->nq        = nq;
    mf->n_rhs     = n_rhs;
    mf->kp        = k->kp;
    if (kparam_blob)
        mf->blob.assign(static_cast< const char* >(kparam_blob), static_cast< const char* >(kparam_blob) + kparam_bytes);
    for (int u = 0; u < l3k::dev::max_unknowns; ++u)
        mf->field_inds[u] = 0;
    if (k->kp.n_unknowns > l3k::dev::max_unknowns)
    {
        setError("n_unknowns > %d unsupported", l3k::dev::max_unknowns);
        return -1;
    }
    for (int u = 0; u < k->kp.n_unknowns; ++u)
    {
        const int fi = field_inds ? field_inds[u] : u;
        if (fi < 0 || fi >= mesh->dofs_per_node)
        {
            setError("field_inds[%d] = %d outside [0, dofs_per_node = %d)", u, fi, mesh->dofs_per_node);
            return -1;
        }
        mf->field_inds[u] = fi;
    }
    mf->dense = k->kp.n_unknowns == mesh->dofs_per_node;
    for (int u = 0; u < k->kp.n_unknowns; ++u)
        mf->dense = mf->dense && mf->field_inds[u] == u;
    // rows of nodes touched by one element only are written (not accumulated) by the element kernel, see l3k_mf_scale
    mf->fuse = mf->dense && mesh->exclusive_end > mesh->exclusive_begin;
    L3K_HIP(hipSetDevice(ctx->device));
    mf->tables_host  = l3k::host::deviceTableBlock(mesh->order, nq);
    const auto& block = mf->tables_host;
    if (int rc = mf->tables.upload(block.data(), block.size(), ctx->stream))
        return rc;
    L3K_HIP(hipStreamSynchronize(ctx->stream));
    *out = mf.release();
    return 0;
}
int l3k_mf_destroy(l3k_mf* mf)
{
    delete mf;
    return 0;
}
int l3k_mf_set_fields(l3k_mf* mf, const double* d_soa, size_t ld)
{
    if (!mf)
    {
        setError("null mf");
        return -1;
    }
    if (d_soa && ld < size_t(mf->mesh->n_owned_nodes + mf->mesh->n_ghost_nodes))
    {
        setError("field leading dimension %zu < number of local nodes", ld);
        return -1;
    }
    mf->fields = d_soa;
    mf->ldf    = ld;
    return 0;
}
int l3k_mf_set_time(l3k_mf* mf, double time)
{
    if (!mf)
    {
        setError("null mf");
        return -1;
    }
    mf->time = time;
    return 0;
}

int l3k_mf_route(l3k_mf* mf, int which, int ncols, int with_energy, char* buf, size_t n)
{
    if (!mf || !buf || n == 0)
    {
        setError("l3k_mf_route: null argument");
        return -1;
    }
    buf[0] = '\0';
    l3k::dev::ElemArgs a;
    double* const      saved = mf->energy_target;
    if (with_energy)
        mf->energy_target = reinterpret_cast< double* >(mf->ctx->red_ws); // (any non-null device pointer: nothing is launched)
    const int rc      = fillArgs(mf, which, ncols, a);
    mf->energy_target = saved;
    if (rc)
        return rc;
    if (mf->ctx->deterministic)
        a.energy = nullptr; // (forEachLaunchRange)
    const l3k_mesh* m = mf->mesh;
    // the pointer relations the launcher looks at: ghost rows in buffers of their own whenever the mesh has ghost nodes and the
    // launch covers border elements (l3k_mf_apply_elems is given separate import / export buffers)
    static const double dummy[2] = {0., 0.};
    a.x = a.y = nullptr;
    a.xg = a.yg = (m->n_ghost_nodes > 0 && (which == 1 || which == 2)) ? const_cast< double* >(dummy) : nullptr;
    a.fuse_beta = mf->fuse;
    const auto* inst = l3k::dev::findInstance(mf->kernel_id, m->order, mf->nq, ncols);
    bool        looped = false;
    if (!inst)
    {
        inst = instanceFor(mf, 1);
        if (!inst)
            return -4;
        if (inst->apply_cols && mf->dense && !mf->ctx->deterministic && !mf->ctx->tune.column_by_column)
            a.n_cols = ncols;
        else
            looped = ncols > 1;
    }
    if (!inst->route)
    {
        setError("this instance carries no route description");
        return -4;
    }
    if (int rc2 = inst->route(a, buf, n))
        return rc2;
    const size_t len = std::strlen(buf);
    if (const auto* meta = l3k::api::findKernel(mf->kernel_id))
        std::snprintf(buf + len, n - len, " [kernel %d \"%s\"%s%s%s]", mf->kernel_id, meta->name, looped ? ", column by column" : "",
                      mf->ctx->deterministic ? ", deterministic: one launch per colour" : "", mf->ctx->reference_z0 ? ", reference z=0" : "");
    return 0;
}

int l3k_mf_scale(l3k_mf* mf, double* d_y, size_t ldy, int ncols, double beta)
{
    if (!mf || !d_y)
    {
        setError("l3k_mf_scale: null argument");
        return -1;
    }
    const l3k_mesh* m    = mf->mesh;
    const int64_t   rows = m->nOwnedDofs();
    if (beta == 1. || rows == 0)
        return 0;
    // rows of exclusive nodes are left to the element kernel (it writes alpha*A*x + beta*y there)
    const int64_t r0 = mf->fuse ? m->exclusive_begin * m->dofs_per_node : rows;
    const int64_t r1 = mf->fuse ? m->exclusive_end * m->dofs_per_node : rows;
    if (r0 > 0)
        hipLaunchKernelGGL(scaleKernel, dim3(gridFor(r0)), dim3(256), 0, mf->ctx->stream, d_y, ldy, r0, ncols, beta);
    if (rows > r1)
        hipLaunchKernelGGL(scaleKernel, dim3(gridFor(rows - r1)), dim3(256), 0, mf->ctx->stream, d_y + r1, ldy, rows - r1,
                           ncols, beta);
    L3K_HIP(hipGetLastError());
    return 0;
}

int l3k_mf_apply_elems(l3k_mf* mf, int which, const double* d_x, size_t ldx, const double* d_xghost, size_t ldxg,
                       double* d_y, size_t ldy, double* d_yghost, size_t ldyg, int ncols, double alpha, double beta)
{
    if (!mf || !d_x || !d_y)
    {
        setError("l3k_mf_apply_elems: null argument");
        return -1;
    }
    int cur_dev = -1;
    if (hipGetDevice(&cur_dev) != hipSuccess || cur_dev != mf->ctx->device) // (several contexts in one process)
        L3K_HIP(hipSetDevice(mf->ctx->device));
    l3k::dev::ElemArgs a;
    if (int rc = fillArgs(mf, which, ncols, a))
        return rc;
    if (a.elem_count > 0 && (a.energy || mf->energy_target)) // (armed but ncols > 1: a.energy is null, never "fused")
        ++mf->energy_expected;
    const l3k_mesh* m = mf->mesh;
    if (ldx < size_t(m->nOwnedDofs()) || ldy < size_t(m->nOwnedDofs()))
    {
        setError("leading dimension smaller than the number of owned dofs");
        return -1;
    }
    if (m->n_ghost_nodes > 0 && (which == 1 || which == 2) && (!d_xghost || !d_yghost))
    {
        setError("mesh has ghost nodes: border elements need the ghost import/export buffers");
        return -1;
    }
    // the kernels move a node's dofs with 16-byte accesses (and run 18 % slower when a node's row straddles a 32-byte
    // boundary: profiles/r01_kbench_vector_alignment.log)
    const auto misaligned = [](const void* p, size_t ld, int nc) {
        return reinterpret_cast< uintptr_t >(p) % 16 != 0 || (nc > 1 && (ld * sizeof(double)) % 16 != 0);
    };
    // only the one-wave kernel uses 16-byte accesses: it takes dense dof layouts with an even number of unknowns
    // (FastCfg::feasible); every other shape runs the generic kernel with 8-byte accesses
    if (mf->dense && mf->kp.n_unknowns % 2 == 0 &&
        (misaligned(d_x, ldx, ncols) || misaligned(d_y, ldy, ncols) ||
         (m->n_ghost_nodes > 0 && d_xghost && misaligned(d_xghost, ldxg, ncols)) ||
         (m->n_ghost_nodes > 0 && d_yghost && misaligned(d_yghost, ldyg, ncols))))
    {
        setError("vectors must be 16-byte aligned (columns too: even leading dimensions); 32-byte alignment is faster");
        return -1;
    }
    a.x     = d_x;
    a.xg    = d_xghost;
    a.y     = d_y;
    a.yg    = d_yghost;
    a.ldx   = ldx;
    a.ldxg  = ldxg;
    a.ldy   = ldy;
    a.ldyg  = ldyg;
    a.alpha     = alpha;
    a.beta      = beta;
    a.fuse_beta = mf->fuse;
    const void* blob = mf->blob.empty() ? nullptr : mf->blob.data();
    const auto* inst = l3k::dev::findInstance(mf->kernel_id, mf->mesh->order, mf->nq, ncols);
    if (inst)
    {
        if (int rc = forEachLaunchRange(mf, which, a, [&](l3k::dev::ElemArgs& r) { return inst->apply(r, blob, mf->ctx->stream); }))
            return rc;
    }
    else
    {
        // no ncols-column instantiation: column by column with the single-column one, as the reference does when
        // fewer columns than n_rhs are passed (algsys/MatrixFreeSystem.hpp:1124-1138)
        inst = instanceFor(mf, 1);
        if (!inst)
            return -4;
        if (inst->apply_cols && mf->dense && !mf->ctx->deterministic && !mf->ctx->tune.column_by_column) // (the switch: cross-check)
        {
            // dense dof layout: all columns in one pass over the elements (node ids, flags and the work ticket once per
            // element), MatrixFreeSystem.hpp:678-688
            l3k::dev::ElemArgs ac = a;
            ac.n_cols             = ncols;
            ac.energy             = nullptr;
            if (int rc = inst->apply_cols(ac, blob, mf->ctx->stream))
                return rc;
        }
        else
        for (int c = 0; c < ncols; ++c)
        {
            l3k::dev::ElemArgs ac = a;
            ac.x  = d_x + ldx * c;
            ac.xg = d_xghost ? d_xghost + ldxg * c : nullptr;
            ac.y  = d_y + ldy * c;
            ac.yg = d_yghost ? d_yghost + ldyg * c : nullptr;
            if (int rc = forEachLaunchRange(mf, which, ac, [&](l3k::dev::ElemArgs& r) { return inst->apply(r, blob, mf->ctx->stream); }))
                return rc;
        }
    }
    // boundary equation kernels registered on this system act on the sides of the same element range
    // (the halves of the interior: all sides of interior elements go with the first half)
    if (which != 4)
        for (l3k_bnd* b : mf->boundary_terms)
            if (int rc = bndApplyImpl(b, which == 3 ? 0 : which, d_x, ldx, d_xghost, ldxg, d_y, ldy, d_yghost, ldyg, ncols, alpha))
                return rc;
    return 0;
}

int l3k_mf_dirichlet_rows(l3k_mf* mf, const double* d_x, size_t ldx, double* d_y, size_t ldy, int ncols, double alpha)
{
    if (!mf || !d_x || !d_y)
    {
        setError("l3k_mf_dirichlet_rows: null argument");
        return -1;
    }
    const auto& rows = mf->mesh->owned_dirichlet_rows;
    if (rows.n == 0)
        return 0;
    hipLaunchKernelGGL(dirichletRowsKernel, dim3(gridFor(int64_t(rows.n))), dim3(256), 0, mf->ctx->stream, rows.ptr,
                       int64_t(rows.n), d_x, ldx, d_y, ldy, ncols, alpha);
    L3K_HIP(hipGetLastError());
    return 0;
}

// s[1] += sum over the owned Dirichlet rows of x_d^2 (their share of x^T A x: those rows of the operator are the identity)
__global__ void dirichletEnergyKernel(const int64_t* __restrict__ rows, int64_t n, const double* __restrict__ x, double* __restrict__ s1)
{
    __shared__ double sh[256];
    double            acc = 0.;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256)
        acc += x[rows[i]] * x[rows[i]];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1)
    {
        if (int(threadIdx.x) < w)
            sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0 && sh[0] != 0.)
        unsafeAtomicAdd(s1, sh[0]);
}

int l3k_mf_energy_begin(l3k_mf* mf, double* d_s)
{
    if (!mf || !d_s)
    {
        setError("l3k_mf_energy_begin: null argument");
        return -1;
    }
    L3K_HIP(hipMemsetAsync(d_s + 1, 0, sizeof(double), mf->ctx->stream));
    // the element kernel accumulates x^T A x only on its single-wave route and only for domain kernels
    mf->energy_target   = mf->boundary_terms.empty() ? d_s + 1 : nullptr;
    mf->energy_done     = 0;
    mf->energy_expected = 0;
    return 0;
}
int l3k_mf_energy_end(l3k_mf* mf, const double* d_x, int* fused)
{
    if (!mf || !d_x || !fused)
    {
        setError("l3k_mf_energy_end: null argument");
        return -1;
    }
    double* const s1  = mf->energy_target;
    *fused            = s1 != nullptr && mf->energy_done == mf->energy_expected;
    mf->energy_target = nullptr;
    const auto& rows  = mf->mesh->owned_dirichlet_rows;
    if (*fused && rows.n > 0)
    {
        const int64_t nr = int64_t(rows.n);
        hipLaunchKernelGGL(dirichletEnergyKernel, dim3(unsigned(std::min< int64_t >((nr + 255) / 256, 1024))), dim3(256), 0,
                           mf->ctx->stream, rows.ptr, nr, d_x, s1);
        L3K_HIP(hipGetLastError());
    }
    return 0;
}
int l3k_mf_apply_energy(l3k_mf* mf, const double* d_x, double* d_y, double* d_s)
{
    if (!mf || !d_x || !d_y || !d_s)
    {
        setError("l3k_mf_apply_energy: null argument");
        return -1;
    }
    const size_t n = size_t(mf->mesh->nOwnedDofs());
    if (int rc = l3k_mf_energy_begin(mf, d_s))
        return rc;
    const int rc = l3k_mf_apply(mf, d_x, n, d_y, n, 1, 1., 0.);
    int       fused = 0;
    if (int rc2 = l3k_mf_energy_end(mf, d_x, &fused))
        return rc2;
    if (rc)
        return rc;
    return fused ? 0 : l3k_cg_dot_pap(mf->ctx, d_x, d_y, int64_t(n), d_s); // (the dot product overwrites s[1])
}

int l3k_mf_apply(l3k_mf* mf, const double* d_x, size_t ldx, double* d_y, size_t ldy, int ncols, double alpha, double beta)
{
    if (!mf)
    {
        setError("null mf");
        return -1;
    }
    if (mf->mesh->n_ghost_nodes != 0)
    {
        setError("l3k_mf_apply is the single-rank form; this mesh has ghost nodes: use the split-phase entry points");
        return -1;
    }
    if (int rc = l3k_mf_scale(mf, d_y, ldy, ncols, beta))
        return rc;
    if (int rc = l3k_mf_apply_elems(mf, 2, d_x, ldx, nullptr, 0, d_y, ldy, nullptr, 0, ncols, alpha, beta))
        return rc;
    return l3k_mf_dirichlet_rows(mf, d_x, ldx, d_y, ldy, ncols, alpha);
}

int l3k_pack_rows(l3k_ctx* ctx, const double* d_src, size_t ld, int ncols, const int32_t* d_idx, int64_t n, double* d_dst)
{
    if (!ctx || (n > 0 && (!d_src || !d_idx || !d_dst)))
    {
        setError("l3k_pack_rows: null argument");
        return -1;
    }
    if (n <= 0)
        return 0;
    hipLaunchKernelGGL(packKernel, dim3(gridFor(n)), dim3(256), 0, ctx->stream, d_src, ld, ncols, d_idx, n, d_dst);
    L3K_HIP(hipGetLastError());
    return 0;
}
int l3k_unpack_add_rows(l3k_ctx* ctx, const double* d_src, int64_t n, const int32_t* d_idx, double* d_dst, size_t ld,
                        int ncols)
{
    if (!ctx || (n > 0 && (!d_src || !d_idx || !d_dst)))
    {
        setError("l3k_unpack_add_rows: null argument");
        return -1;
    }
    if (n <= 0)
        return 0;
    hipLaunchKernelGGL(unpackAddKernel, dim3(gridFor(n)), dim3(256), 0, ctx->stream, d_src, n, d_idx, d_dst, ld, ncols);
    L3K_HIP(hipGetLastError());
    return 0;
}

int l3k_mf_diag_rhs(l3k_mf* mf, int which, const double* d_dirichlet_vals, size_t ldg, double* d_diag, double* d_rhs,
                    size_t ldr, double* d_diag_ghost, double* d_rhs_ghost, size_t ldrg, int finalize)
{
    if (!mf || !d_rhs)
    {
        setError("l3k_mf_diag_rhs: null argument");
        return -1;
    }
    l3k::dev::ElemArgs a;
    if (int rc = fillArgs(mf, which, mf->n_rhs, a))
        return rc;
    a.dirichlet_vals = d_dirichlet_vals;
    a.ldg            = ldg;
    a.y              = d_rhs;
    a.ldy            = ldr;
    a.yg             = d_rhs_ghost;
    a.ldyg           = ldrg;
    a.diag           = d_diag;
    a.diag_g         = d_diag_ghost;
    if (mf->mesh->n_ghost_nodes > 0 && which != 0 && (!d_rhs_ghost || (d_diag && !d_diag_ghost)))
    {
        setError("mesh has ghost nodes: border elements need the ghost diag / rhs buffers");
        return -1;
    }
    const auto* inst = instanceFor(mf, mf->n_rhs);
    if (!inst)
        return -4;
    if (int rc = forEachLaunchRange(mf, which, a, [&](l3k::dev::ElemArgs& r) {
            return inst->diag_rhs(r, mf->blob.empty() ? nullptr : mf->blob.data(), mf->ctx->stream);
        }))
        return rc;
    for (l3k_bnd* b : mf->boundary_terms)
        if (int rc = bndDiagRhsImpl(b, which, d_dirichlet_vals, ldg, d_diag, d_rhs, ldr, d_diag_ghost, d_rhs_ghost, ldrg))
            return rc;
    if (finalize && mf->mesh->owned_dirichlet_rows.n > 0 && d_diag)
    {
        const auto& rows = mf->mesh->owned_dirichlet_rows;
        hipLaunchKernelGGL(dirichletFinalizeKernel, dim3(gridFor(int64_t(rows.n))), dim3(256), 0, mf->ctx->stream, rows.ptr,
                           int64_t(rows.n), d_dirichlet_vals, ldg, d_diag, d_rhs, ldr, mf->n_rhs);
        L3K_HIP(hipGetLastError());
    }
    return 0;
}

int l3k_mf_dirichlet_finalize(l3k_mf* mf, const double* d_dirichlet_vals, size_t ldg, double* d_diag, double* d_rhs, size_t ldr)
{
    if (!mf || !d_diag || !d_rhs)
    {
        setError("l3k_mf_dirichlet_finalize: null argument");
        return -1;
    }
    const auto& rows = mf->mesh->owned_dirichlet_rows;
    if (rows.n == 0)
        return 0;
    hipLaunchKernelGGL(dirichletFinalizeKernel, dim3(gridFor(int64_t(rows.n))), dim3(256), 0, mf->ctx->stream, rows.ptr,
                       int64_t(rows.n), d_dirichlet_vals, ldg, d_diag, d_rhs, ldr, mf->n_rhs);
    L3K_HIP(hipGetLastError());
    return 0;
}

// K_e of [first, first + count) into d_K (row-major) through the tiled layout: sub-batches formed on the context's stream into one
// of the system's two buffers, turned on the second stream (events order the reuse of the buffers, as in l3k_assemble_global)
static int assembleRowMajorViaTiled(l3k_mf* mf, const l3k::dev::Instance* inst, int64_t first, int64_t count, double* d_K)
{
    const l3k_mesh* m  = mf->mesh;
    const int       N1 = m->order + 1, U = mf->kp.n_unknowns, Nd = N1 * N1 * N1 * U;
    const size_t    mat = size_t(Nd) * Nd; // doubles per matrix
    // sub-batches of <= 1.5 GiB of tiled matrices per buffer (large launches: the kernels of the route are all memory-bound, so their
    // overlap on the two streams buys little and small sub-batches only add launch tails: tools/r04_stored_subbatch.py)
    int64_t nb = int64_t((size_t(1536) << 20) / (mat * sizeof(double)));
    nb         = nb < 1 ? 1 : nb;
    if (mf->ctx->tune.assemble_sub_batch > 0)
        nb = mf->ctx->tune.assemble_sub_batch;
    nb = nb > count ? count : nb;
    const size_t kd = size_t(nb) * mat, wd = inst->assemble_ws_doubles * size_t(nb) + 1;
    auto&        g  = mf->gasm;
    L3K_HIP(hipSetDevice(mf->ctx->device));
    if (g.doubles < kd + wd)
    {
        for (int k = 0; k < 2; ++k)
        {
            if (g.buf[k])
                L3K_HIP(hipFree(g.buf[k]));
            g.buf[k] = nullptr;
        }
        g.doubles = 0;
        for (int k = 0; k < 2; ++k)
            L3K_HIP(hipMalloc(reinterpret_cast< void** >(&g.buf[k]), (kd + wd) * sizeof(double)));
        g.doubles = kd + wd;
    }
    if (!g.second)
    {
        L3K_HIP(hipStreamCreateWithFlags(&g.second, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k)
        {
            L3K_HIP(hipEventCreateWithFlags(&g.formed[k], hipEventDisableTiming));
            L3K_HIP(hipEventCreateWithFlags(&g.consumed[k], hipEventDisableTiming));
        }
    }
    hipStream_t sa   = mf->ctx->stream;
    const void* blob = mf->blob.empty() ? nullptr : mf->blob.data();
    // bitwise symmetric matrices, as the reference returns them: the x-major tiled layout and the one-pass mirroring transposition
    // (api_assembled.hip).  l3k_tuning::assemble_no_symmetrise: the plain tiled layout and the plain transposition (K[i][j] and
    // K[j][i] then differ by rounding) -- the cross-check of the former
    const bool sym = !mf->ctx->tune.assemble_no_symmetrise;
    for (int k = 0; k < 2; ++k) // (the flags of degenerate elements: the trailing double of each coefficient workspace)
        L3K_HIP(hipMemsetAsync(g.buf[k] + kd + inst->assemble_ws_doubles * size_t(nb), 0, sizeof(double), sa));
    int64_t done  = 0;
    int     n_sub = 0;
    for (int i = 0; done < count; ++i, ++n_sub)
    {
        const int     k = i & 1;
        const int64_t n = count - done < nb ? count - done : nb;
        if (i >= 2)
            L3K_HIP(hipStreamWaitEvent(sa, g.consumed[k], 0)); // the transposition of sub-batch i - 2 has read this buffer
        l3k::dev::ElemArgs a;
        if (int rc = fillArgs(mf, 2, mf->n_rhs, a))
            return rc;
        a.elem_begin     = first + done;
        a.elem_count     = n;
        a.elem_begin_out = 0;
        a.K              = g.buf[k];
        a.K_tiled        = sym ? 2 : 1;
        a.workspace      = g.buf[k] + kd + size_t(nb - n) * inst->assemble_ws_doubles; // (the flag keeps one position per buffer)
        if (int rc = inst->assemble(a, blob, sa))
            return rc;
        L3K_HIP(hipEventRecord(g.formed[k], sa));
        L3K_HIP(hipStreamWaitEvent(g.second, g.formed[k], 0));
        if (int rc = sym ? launchTiledXToRowMajorSym(U, N1, n, g.buf[k], d_K + size_t(done) * mat, g.second)
                         : launchTiledToRowMajor(U, N1, n, g.buf[k], d_K + size_t(done) * mat, g.second))
            return rc;
        L3K_HIP(hipEventRecord(g.consumed[k], g.second));
        done += n;
    }
    for (int k = 0; k < 2 && k < n_sub; ++k)
        L3K_HIP(hipStreamWaitEvent(sa, g.consumed[k], 0)); // later work on the context's stream sees the finished matrices
    double flags[2] = {0., 0.};
    for (int k = 0; k < 2; ++k)
        L3K_HIP(hipMemcpyAsync(&flags[k], g.buf[k] + kd + inst->assemble_ws_doubles * size_t(nb), sizeof(double), hipMemcpyDeviceToHost, sa));
    L3K_HIP(hipStreamSynchronize(sa));
    if (flags[0] != 0. || flags[1] != 0.)
    {
        setError("Encountered degenerate element ( |J| <= 0 )"); // algsys/AssembleLocalSystem.hpp:249
        return -2;
    }
    return 0;
}

int l3k_local_assemble(l3k_mf* mf, int64_t first, int64_t count, double* d_K, double* d_F, double* d_checksum)
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
    if (count == 0)
        return 0;
    const auto* inst = instanceFor(mf, mf->n_rhs);
    if (!inst)
        return -4;
    l3k::dev::ElemArgs a;
    if (int rc = fillArgs(mf, 2, mf->n_rhs, a))
        return rc;
    a.elem_begin     = first;
    a.elem_count     = count;
    a.elem_begin_out = 0;
    a.K              = d_K;
    a.F              = d_F;
    a.checksum       = d_checksum;
    hipStream_t s    = mf->ctx->stream;
    const void* blob = mf->blob.empty() ? nullptr : mf->blob.data();
    // Stored row-major matrices: formed in the tiled layout (coalesced stores) and turned by a transposition kernel on a second
    // stream, sub-batch by sub-batch -- the assembly kernels cannot fill the 64-byte lines of a row-major K_e (four workgroups and
    // two iterations per line: 3.9 x write traffic).  l3k_tuning::assemble_direct_store keeps the direct store (cross-check).
    const l3k_tuning& tune = mf->ctx->tune;
    // (orders >= 4, or on request: below, a matrix is a few KB and the direct store wins -- order 2: 12.9 M against 6.7 M matrices/s,
    // profiles/r04_stored_assembly.jsonl)
    const bool via_tiled = d_K && inst->assemble_tiled && mf->kp.n_unknowns <= 4 && m->order <= 7 && !tune.assemble_dense &&
                           !tune.assemble_two_launches && !tune.assemble_direct_store && (m->order >= 4 || tune.assemble_sub_batch > 0);
    if (via_tiled)
    {
        if (int rc = assembleRowMajorViaTiled(mf, inst, first, count, d_K))
            return rc;
        a.K = nullptr; // (a checksum asked for beside K comes from a streaming pass below)
    }
    if (a.K || d_checksum)
    {
        const size_t need = inst->assemble_ws_doubles * size_t(count) + 1;
        if (need > mf->ws_doubles)
        {
            if (mf->ws)
                L3K_HIP(hipFree(mf->ws));
            mf->ws = nullptr;
            L3K_HIP(hipMalloc(reinterpret_cast< void** >(&mf->ws), need * sizeof(double)));
            mf->ws_doubles = need;
        }
        a.workspace = mf->ws;
        double* flag = mf->ws + inst->assemble_ws_doubles * size_t(count);
        L3K_HIP(hipMemsetAsync(flag, 0, sizeof(double), s));
        if (d_checksum)
            L3K_HIP(hipMemsetAsync(d_checksum, 0, sizeof(double) * size_t(count), s));
        if (int rc = inst->assemble(a, blob, s))
            return rc;
        double degenerate = 0.;
        L3K_HIP(hipMemcpyAsync(&degenerate, flag, sizeof(double), hipMemcpyDeviceToHost, s));
        L3K_HIP(hipStreamSynchronize(s));
        if (degenerate != 0.)
        {
            setError("Encountered degenerate element ( |J| <= 0 )"); // algsys/AssembleLocalSystem.hpp:249
            return -2;
        }
    }
    if (d_F)
    {
        // F_e = sum_q w detJ B_q^T f_q: the sum-factorised RHS-mode kernel without Dirichlet lifting, element-local output
        a.dirichlet      = nullptr;
        a.elem_flags     = nullptr;
        a.dirichlet_vals = nullptr;
        a.diag           = nullptr;
        a.local_out      = 1;
        a.y              = d_F; // unused for addressing in local_out mode, must be non-null
        if (int rc = inst->diag_rhs(a, blob, s))
            return rc;
    }
    return 0;
}
// K_e of the elements [first, first + count) in the TILED layout (include/l3k.h): what l3k_assemble_global keeps between its two kernels
int l3k_local_assemble_tiled(l3k_mf* mf, int64_t first, int64_t count, double* d_Kt)
{
    if (!mf || !d_Kt)
    {
        setError("l3k_local_assemble_tiled: null argument");
        return -1;
    }
    const l3k_mesh* m = mf->mesh;
    if (first < 0 || count < 0 || first + count > m->n_elems)
    {
        setError("element range [%lld, %lld) outside [0, %lld)", (long long)first, (long long)(first + count), (long long)m->n_elems);
        return -1;
    }
    if (count == 0)
        return 0;
    const auto* inst = instanceFor(mf, mf->n_rhs);
    if (!inst)
        return -4;
    if (!inst->assemble_tiled)
    {
        setError("l3k_local_assemble_tiled: this shape has no sum-factorised assembly kernel");
        return -1;
    }
    l3k::dev::ElemArgs a;
    if (int rc = fillArgs(mf, 2, mf->n_rhs, a))
        return rc;
    a.elem_begin     = first;
    a.elem_count     = count;
    a.elem_begin_out = 0;
    a.K              = d_Kt;
    a.K_tiled        = 1;
    hipStream_t  s    = mf->ctx->stream;
    const size_t need = inst->assemble_ws_doubles * size_t(count) + 1;
    if (need > mf->ws_doubles)
    {
        if (mf->ws)
            L3K_HIP(hipFree(mf->ws));
        mf->ws = nullptr;
        L3K_HIP(hipMalloc(reinterpret_cast< void** >(&mf->ws), need * sizeof(double)));
        mf->ws_doubles = need;
    }
    a.workspace  = mf->ws;
    double* flag = mf->ws + inst->assemble_ws_doubles * size_t(count);
    L3K_HIP(hipMemsetAsync(flag, 0, sizeof(double), s));
    if (int rc = inst->assemble(a, mf->blob.empty() ? nullptr : mf->blob.data(), s))
        return rc;
    double degenerate = 0.;
    L3K_HIP(hipMemcpyAsync(&degenerate, flag, sizeof(double), hipMemcpyDeviceToHost, s));
    L3K_HIP(hipStreamSynchronize(s));
    if (degenerate != 0.)
    {
        setError("Encountered degenerate element ( |J| <= 0 )"); // algsys/AssembleLocalSystem.hpp:249
        return -2;
    }
    return 0;
}
// assembleGlobalSystem (algsys/AssembleGlobalSystem.hpp:20-53: per element assembleLocalSystem -> scatterLocalSystem) for the
// elements [first, first + count) as ONE call: sub-batches of element systems are formed into one of two workspace buffers on the
// context's stream while the previous sub-batch is summed into the CSR values on a second stream -- the assembly kernels are
// bound by the FP64 pipe, the scatter by the memory-side atomic units, so the two overlap (events order buffer reuse).
int l3k_assemble_global(l3k_mf* mf, int64_t first, int64_t count, const int64_t* d_row_ptr, const int32_t* d_col_ind,
                        double* d_values, double* d_rhs, size_t ldr, int skip_dirichlet, size_t workspace_bytes, int64_t* n_missing)
{
    if (!mf || !d_row_ptr || !d_col_ind || !d_values)
    {
        setError("l3k_assemble_global: null argument");
        return -1;
    }
    const l3k_mesh* m = mf->mesh;
    if (first < 0 || count < 0 || first + count > m->n_elems)
    {
        setError("element range [%lld, %lld) outside [0, %lld)", (long long)first, (long long)(first + count), (long long)m->n_elems);
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
    if (count == 0)
        return 0;
    const auto* inst = instanceFor(mf, mf->n_rhs);
    if (!inst)
        return -4;
    L3K_HIP(hipSetDevice(mf->ctx->device));
    // element matrices between the two kernels in the tiled layout where the sum-factorised assembly kernel exists (its stores
    // coalesce, the scatter transposes a row through LDS); the dense cross-check kernel writes the reference's row-major layout
    const int    tiled = inst->assemble_tiled && mf->kp.n_unknowns <= 4 && m->order <= 7 && !mf->ctx->tune.assemble_dense ? 1 : 0;
    const int    N1 = m->order + 1, Nd = N1 * N1 * N1 * mf->kp.n_unknowns, R = mf->n_rhs;
    const size_t per_elem = sizeof(double) * (size_t(Nd) * Nd + size_t(Nd) * R + inst->assemble_ws_doubles);
    if (workspace_bytes == 0)
        workspace_bytes = size_t(1) << 30;
    int64_t nb = int64_t(workspace_bytes / 2 / per_elem);
    nb         = nb < 1 ? 1 : (nb > count ? count : nb);
    // at least ~8 sub-batches per call where the batch stays large enough for full launches (the overlap needs several)
    if (const int64_t eighth = (count + 7) / 8; nb > eighth && eighth >= 512)
        nb = eighth;
    while (nb * Nd > int64_t(0x7fffffff) / 64 && nb > 1) // (launch-size limits of the kernels)
        nb /= 2;
    const size_t kd = size_t(nb) * Nd * Nd, fd = d_rhs ? size_t(nb) * Nd * R : 0, wd = inst->assemble_ws_doubles * size_t(nb) + 1;
    auto&        g  = mf->gasm;
    if (g.doubles < kd + fd + wd)
    {
        for (int k = 0; k < 2; ++k)
        {
            if (g.buf[k])
                L3K_HIP(hipFree(g.buf[k]));
            g.buf[k] = nullptr;
        }
        g.doubles = 0;
        for (int k = 0; k < 2; ++k)
            L3K_HIP(hipMalloc(reinterpret_cast< void** >(&g.buf[k]), (kd + fd + wd) * sizeof(double)));
        g.doubles = kd + fd + wd;
    }
    if (!g.second)
    {
        L3K_HIP(hipStreamCreateWithFlags(&g.second, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k)
        {
            L3K_HIP(hipEventCreateWithFlags(&g.formed[k], hipEventDisableTiming));
            L3K_HIP(hipEventCreateWithFlags(&g.consumed[k], hipEventDisableTiming));
        }
    }
    struct Side
    {
        double *   K, *F, *ws;
        hipEvent_t formed, consumed;
    } side[2];
    for (int k = 0; k < 2; ++k)
        side[k] = Side{g.buf[k], g.buf[k] + kd, g.buf[k] + kd + fd, g.formed[k], g.consumed[k]};
    struct
    {
        hipStream_t s;
    } second{g.second};
    hipStream_t         sa      = mf->ctx->stream;
    const void*         blob    = mf->blob.empty() ? nullptr : mf->blob.data();
    unsigned long long* d_count = n_missing ? mf->ctx->missCounter() : nullptr;
    if (d_count)
        L3K_HIP(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), sa));
    // (the flags of degenerate elements: the trailing double of each coefficient workspace, set by the kernels, never cleared here)
    for (auto& sd : side)
        L3K_HIP(hipMemsetAsync(sd.ws + inst->assemble_ws_doubles * size_t(nb), 0, sizeof(double), sa));
    int64_t done  = 0;
    int     n_sub = 0;
    for (int i = 0; done < count; ++i, ++n_sub)
    {
        Side&         sd = side[i & 1];
        const int64_t n  = count - done < nb ? count - done : nb;
        if (i >= 2)
            L3K_HIP(hipStreamWaitEvent(sa, sd.consumed, 0)); // the scatter of sub-batch i - 2 has read this buffer
        l3k::dev::ElemArgs a;
        if (int rc = fillArgs(mf, 2, mf->n_rhs, a))
            return rc;
        a.elem_begin     = first + done;
        a.elem_count     = n;
        a.elem_begin_out = 0;
        a.K              = sd.K;
        a.K_tiled        = tiled;
        a.F              = d_rhs ? sd.F : nullptr;
        a.checksum       = nullptr;
        // (the kernels keep the degenerate-element flag behind the coefficients of THEIR elem_count elements: a short last
        // sub-batch works in the tail of the buffer, so that the flag has one position per buffer)
        a.workspace      = sd.ws + size_t(nb - n) * inst->assemble_ws_doubles;
        if (int rc = inst->assemble(a, blob, sa))
            return rc;
        if (d_rhs)
        {
            L3K_HIP(hipMemsetAsync(sd.F, 0, sizeof(double) * size_t(n) * Nd * R, sa));
            a.dirichlet      = nullptr;
            a.elem_flags     = nullptr;
            a.dirichlet_vals = nullptr;
            a.diag           = nullptr;
            a.local_out      = 1;
            a.y              = sd.F;
            if (int rc = inst->diag_rhs(a, blob, sa))
                return rc;
        }
        L3K_HIP(hipEventRecord(sd.formed, sa));
        L3K_HIP(hipStreamWaitEvent(second.s, sd.formed, 0));
        if (int rc = launchAssembledScatter(mf, first + done, n, sd.K, d_rhs ? sd.F : nullptr, d_row_ptr, d_col_ind, d_values, d_rhs,
                                            ldr, skip_dirichlet, d_count, second.s, tiled))
            return rc;
        L3K_HIP(hipEventRecord(sd.consumed, second.s));
        done += n;
    }
    for (int k = 0; k < 2 && k < n_sub; ++k)
        L3K_HIP(hipStreamWaitEvent(sa, side[k].consumed, 0)); // later work on the context's stream sees the finished values
    double             flags[2] = {0., 0.};
    unsigned long long h        = 0;
    for (int k = 0; k < 2; ++k)
        L3K_HIP(hipMemcpyAsync(&flags[k], side[k].ws + inst->assemble_ws_doubles * size_t(nb), sizeof(double), hipMemcpyDeviceToHost, sa));
    if (d_count)
        L3K_HIP(hipMemcpyAsync(&h, d_count, sizeof h, hipMemcpyDeviceToHost, sa));
    L3K_HIP(hipStreamSynchronize(sa));
    if (n_missing)
        *n_missing = int64_t(h);
    if (flags[0] != 0. || flags[1] != 0.)
    {
        setError("Encountered degenerate element ( |J| <= 0 )"); // algsys/AssembleLocalSystem.hpp:249
        return -2;
    }
    return 0;
}
// ------------------------------------------------------------------------------------------------ boundary terms
int l3k_bnd_create(l3k_ctx* ctx, l3k_mesh* mesh, int kernel_id, const void* kparam_blob, size_t kparam_bytes,
                   const l3k_asmopts* opts, const int* field_inds, int n_rhs, int64_t n_faces, const int64_t* face_elem,
                   const uint8_t* face_side, l3k_bnd** out)
{
    if (!ctx || !mesh || !out || n_faces < 0 || (n_faces > 0 && (!face_elem || !face_side)))
    {
        setError("l3k_bnd_create: bad argument");
        return -1;
    }
    const auto* k = findKernel(kernel_id);
    if (!k || !k->boundary)
    {
        setError("kernel id %d is not a boundary equation kernel", kernel_id);
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
    if (n_rhs < 1 || k->kp.n_unknowns > l3k::dev::max_unknowns)
    {
        setError("n_rhs must be >= 1 and n_unknowns <= %d", l3k::dev::max_unknowns);
        return -1;
    }
    for (int64_t i = 0; i < n_faces; ++i) // operand shapes must match what the kernel assumes
        if (face_elem[i] < 0 || face_elem[i] >= mesh->n_elems || face_side[i] >= 6)
        {
            setError("side %lld = (element %lld, side %d) is outside the mesh", (long long)i, (long long)face_elem[i],
                     int(face_side[i]));
            return -1;
        }
    const l3k_asmopts o = opts ? *opts : l3k_asmopts{1, 0, 0};
    auto              b = std::make_unique< l3k_bnd >();
    b->ctx = ctx, b->mesh = mesh, b->kernel_id = kernel_id, b->n_rhs = n_rhs, b->kp = k->kp;
    b->nq = l3k_n_qps1d(mesh->order, o.value_order, o.derivative_order);
    if (kparam_blob)
        b->blob.assign(static_cast< const char* >(kparam_blob), static_cast< const char* >(kparam_blob) + kparam_bytes);
    for (int u = 0; u < l3k::dev::max_unknowns; ++u)
        b->field_inds[u] = 0;
    for (int u = 0; u < k->kp.n_unknowns; ++u)
    {
        const int fi = field_inds ? field_inds[u] : u;
        if (fi < 0 || fi >= mesh->dofs_per_node)
        {
            setError("field_inds[%d] = %d outside [0, dofs_per_node = %d)", u, fi, mesh->dofs_per_node);
            return -1;
        }
        b->field_inds[u] = fi;
    }
    // sides of interior elements first (they need no ghost data, like the interior elements themselves)
    std::vector< int64_t > fe;
    std::vector< uint8_t > fs;
    for (int pass = 0; pass < 2; ++pass)
    {
        for (int64_t i = 0; i < n_faces; ++i)
            if ((face_elem[i] >= mesh->n_interior) == (pass == 1))
            {
                fe.push_back(face_elem[i]);
                fs.push_back(face_side[i]);
            }
        if (pass == 0)
            b->n_interior_faces = int64_t(fe.size());
    }
    b->n_faces = n_faces;
    if (ctx->deterministic)
    {
        // greedy colouring of the sides by ALL EIGHT corner nodes of their elements: the side kernel scatter-adds over every
        // node of the element (the normal derivative couples all of them, device/boundary.hpp), so two sides may share a
        // colour only if their elements share no node -- for conforming hexes: no corner.  (Colouring by the side's own four
        // corners let the z- side of a corner element and the x- side of the element stacked on it into one launch.)
        // Class by class; the lists are stored in (class, colour) order
        if (!mesh->det_built)
        {
            setError("deterministic mode was enabled after this mesh was created: create the mesh with the mode on");
            return -1;
        }
        std::unordered_map< uint32_t, uint64_t > used;
        std::vector< uint8_t >                   colour(fe.size());
        int                                      n_colours = 0;
        for (size_t i = 0; i < fe.size(); ++i)
        {
            uint64_t taken = 0;
            for (int v = 0; v < 8; ++v)
            {
                const auto it = used.find(mesh->det_corner_nodes[size_t(fe[i]) * 8 + v]);
                if (it != used.end())
                    taken |= it->second;
            }
            int c = 0;
            while (c < 64 && ((taken >> c) & 1u))
                ++c;
            if (c == 64)
            {
                setError("deterministic mode: more than 64 colours needed for the boundary sides");
                return -1;
            }
            colour[i] = uint8_t(c);
            n_colours = std::max(n_colours, c + 1);
            for (int v = 0; v < 8; ++v)
                used[mesh->det_corner_nodes[size_t(fe[i]) * 8 + v]] |= uint64_t(1) << c;
        }
        std::vector< size_t > order(fe.size());
        for (size_t i = 0; i < order.size(); ++i)
            order[i] = i;
        auto by_colour = [&](size_t x, size_t y) { return colour[x] < colour[y]; };
        std::stable_sort(order.begin(), order.begin() + b->n_interior_faces, by_colour);
        std::stable_sort(order.begin() + b->n_interior_faces, order.end(), by_colour);
        std::vector< int64_t > fe2(fe.size());
        std::vector< uint8_t > fs2(fs.size());
        for (size_t i = 0; i < order.size(); ++i)
            fe2[i] = fe[order[i]], fs2[i] = fs[order[i]];
        for (int cls = 0; cls < 2; ++cls)
        {
            const int64_t lo = cls ? b->n_interior_faces : 0, hi = cls ? n_faces : b->n_interior_faces;
            b->det_ptr[cls].assign(static_cast< size_t >(n_colours) + 1, hi);
            int64_t i = lo;
            for (int c = 0; c < n_colours; ++c)
            {
                b->det_ptr[cls][c] = i;
                while (i < hi && colour[order[size_t(i)]] == c)
                    ++i;
            }
        }
        fe.swap(fe2);
        fs.swap(fs2);
    }
    L3K_HIP(hipSetDevice(ctx->device));
    const auto block = l3k::host::deviceTableBlock(mesh->order, b->nq);
    if (int rc = b->tables.upload(block.data(), block.size(), ctx->stream))
        return rc;
    if (int rc = b->face_elem.upload(fe.data(), fe.size(), ctx->stream))
        return rc;
    if (int rc = b->face_side.upload(fs.data(), fs.size(), ctx->stream))
        return rc;
    L3K_HIP(hipStreamSynchronize(ctx->stream));
    *out = b.release();
    return 0;
}
int l3k_bnd_destroy(l3k_bnd* bnd)
{
    delete bnd;
    return 0;
}
int l3k_bnd_set_fields(l3k_bnd* bnd, const double* d_soa, size_t ld)
{
    if (!bnd)
    {
        setError("null bnd");
        return -1;
    }
    if (d_soa && ld < size_t(bnd->mesh->n_owned_nodes + bnd->mesh->n_ghost_nodes))
    {
        setError("field leading dimension %zu < number of local nodes", ld);
        return -1;
    }
    bnd->fields = d_soa;
    bnd->ldf    = ld;
    return 0;
}
int l3k_bnd_set_time(l3k_bnd* bnd, double time)
{
    if (!bnd)
    {
        setError("null bnd");
        return -1;
    }
    bnd->time = time;
    return 0;
}
int l3k_bnd_apply(l3k_bnd* bnd, int which, const double* d_x, size_t ldx, const double* d_xghost, size_t ldxg, double* d_y,
                  size_t ldy, double* d_yghost, size_t ldyg, int ncols, double alpha)
{
    if (!bnd || !d_x || !d_y)
    {
        setError("l3k_bnd_apply: null argument");
        return -1;
    }
    return bndApplyImpl(bnd, which, d_x, ldx, d_xghost, ldxg, d_y, ldy, d_yghost, ldyg, ncols, alpha);
}
int l3k_bnd_diag_rhs(l3k_bnd* bnd, int which, const double* d_dirichlet_vals, size_t ldg, double* d_diag, double* d_rhs,
                     size_t ldr, double* d_diag_ghost, double* d_rhs_ghost, size_t ldrg)
{
    if (!bnd || !d_rhs)
    {
        setError("l3k_bnd_diag_rhs: null argument");
        return -1;
    }
    return bndDiagRhsImpl(bnd, which, d_dirichlet_vals, ldg, d_diag, d_rhs, ldr, d_diag_ghost, d_rhs_ghost, ldrg);
}
int l3k_mf_attach_boundary(l3k_mf* mf, l3k_bnd* bnd)
{
    if (!mf || !bnd)
    {
        setError("l3k_mf_attach_boundary: null argument");
        return -1;
    }
    if (bnd->mesh != mf->mesh)
    {
        setError("boundary term and system live on different meshes");
        return -1;
    }
    if (bnd->n_rhs != mf->n_rhs)
    {
        setError("boundary term has n_rhs = %d, system has %d", bnd->n_rhs, mf->n_rhs);
        return -1;
    }
    mf->boundary_terms.push_back(bnd);
    return 0;
}

} // extern "C"
