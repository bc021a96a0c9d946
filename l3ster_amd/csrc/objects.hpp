// objects.hpp -- the objects behind the opaque handles of include/l3k.h and the helpers the api_*.hip files share.
#ifndef L3K_OBJECTS_HPP
#define L3K_OBJECTS_HPP

#include "l3k.h"

#include "device/common.hpp"
#include "host/tables.hpp"
#include "user_kernels.hpp"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <type_traits>
#include <string>
#include <utility>
#include <vector>

namespace l3k::dev
{
const char* lastError();
}
using l3k::dev::setError;

#define L3K_HIP(call)                                                                                                  \
    do                                                                                                                 \
    {                                                                                                                  \
        const hipError_t err_ = (call);                                                                                \
        if (err_ != hipSuccess)                                                                                        \
        {                                                                                                              \
            setError("%s failed: %s (%s:%d)", #call, hipGetErrorString(err_), __FILE__, __LINE__);                     \
            return -3;                                                                                                 \
        }                                                                                                              \
    } while (0)


namespace l3k::api
{
inline unsigned gridFor(int64_t n, int block = 256)
{
    const int64_t g = (n + block - 1) / block;
    return static_cast< unsigned >(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

template < typename T >
struct DevBuf
{
    T*     ptr = nullptr;
    size_t n   = 0;
    DevBuf()   = default;
    DevBuf(const DevBuf&)            = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : ptr{o.ptr}, n{o.n}
    {
        o.ptr = nullptr;
        o.n   = 0;
    }
    DevBuf& operator=(DevBuf&& o) noexcept
    {
        if (this != &o)
        {
            if (ptr)
                (void)hipFree(ptr);
            ptr   = o.ptr;
            n     = o.n;
            o.ptr = nullptr;
            o.n   = 0;
        }
        return *this;
    }
    ~DevBuf()
    {
        if (ptr)
            (void)hipFree(ptr);
    }
    int alloc(size_t count)
    {
        n = count;
        if (count == 0)
            return 0;
        L3K_HIP(hipMalloc(reinterpret_cast< void** >(&ptr), count * sizeof(T)));
        return 0;
    }
    int upload(const T* host, size_t count, hipStream_t s)
    {
        n = count;
        if (count == 0)
            return 0;
        L3K_HIP(hipMalloc(reinterpret_cast< void** >(&ptr), count * sizeof(T)));
        L3K_HIP(hipMemcpyAsync(ptr, host, count * sizeof(T), hipMemcpyHostToDevice, s));
        return 0;
    }
};
} // namespace l3k::api
using l3k::api::DevBuf;
using l3k::api::gridFor;

inline constexpr int l3k_cg_blocks = 1024; // blocks of the PCG's two-stage reductions (api_solver.hip)
struct l3k_ctx
{
    int         device;
    hipStream_t stream;
    // bitwise-reproducible mode (L3K_DETERMINISTIC=1 or l3k_ctx_set_deterministic): element launches go colour by colour
    // (no two elements of a launch share a node), so every row of y receives its contributions in a fixed order
    bool        deterministic = false;
    // l3k_ctx_set_reference_z0: matrix-free applies hand domain kernels Point{x, y, 0.} as the reference's hex sum-factorisation
    // path does (algsys/SumFactorization.hpp:732); default: the true point (what its local-element path passes)
    bool        reference_z0 = false;
    l3k_tuning  tune = l3k::dev::defaultTuning(); // launch-route settings (the environment is read once, in l3k_ctx_create)
    double*     red_ws = nullptr; // per-block partial sums of the PCG dot products (cg_blocks * 2 doubles)
    uint32_t*   work_counters = nullptr; // batch counters of the single-wave element kernel (8 x 128 bytes) + one more line:
    // the counter of l3k_assembled_scatter's entries outside the graph (no allocation per call)
    unsigned long long* missCounter() const { return reinterpret_cast< unsigned long long* >(work_counters + 8 * 32); }
    // global-memory working sets of the element kernels whose buffers exceed the LDS (ElemArgs::scratch): grown on demand
    double* scratch       = nullptr;
    size_t  scratch_bytes = 0;
    ~l3k_ctx()
    {
        if (red_ws)
            (void)hipFree(red_ws);
        if (work_counters)
            (void)hipFree(work_counters);
        if (scratch)
            (void)hipFree(scratch);
    }
};
struct l3k_mesh
{
    l3k_ctx*            ctx;
    int                 dim, order, dofs_per_node;
    int64_t             n_elems, n_interior, n_owned_nodes, n_ghost_nodes;
    DevBuf< uint32_t >  elem_nodes;
    DevBuf< double >    elem_verts;
    DevBuf< uint8_t >   dirichlet;
    DevBuf< int64_t >   owned_dirichlet_rows;
    DevBuf< uint8_t >   elem_flags;
    int64_t             exclusive_begin = 0, exclusive_end = 0;
    // scatter order of the single-wave kernel: slot_tab[lane * 8 + k] = scatter slot of local node lane + (p+1)^2 * k; slots
    // [0, n_shell) are the element's non-internal nodes in ascending node-id order of a typical element (runs of contiguous
    // rows in y), slots [n_shell, N) its internal nodes = exactly the nodes of [exclusive_begin, exclusive_end)
    DevBuf< uint16_t >  slot_tab;
    int                 n_shell = 0;
    bool                all_affine = false; // every element is a parallelepiped (flags bit 1 of all elements)
    // deterministic mode: copies of the element arrays with the elements of each class (interior, border) sorted by colour;
    // det_ptr[0][c] .. det_ptr[0][c + 1] = interior elements of colour c, det_ptr[1][...] the border ones (positions in the
    // permuted arrays, which keep the interior elements first)
    bool                   det_built = false;
    DevBuf< uint32_t >     det_elem_nodes;
    DevBuf< double >       det_elem_verts;
    DevBuf< uint8_t >      det_elem_flags;
    std::vector< int64_t > det_ptr[2];
    std::vector< uint32_t > det_corner_nodes; // host, [n_elems][8] in the ORIGINAL element order: colouring of boundary sides
    int64_t nOwnedDofs() const { return n_owned_nodes * dofs_per_node; }
    int64_t nLocalDofs() const { return (n_owned_nodes + n_ghost_nodes) * dofs_per_node; }
};
struct l3k_bnd;
struct l3k_mf
{
    std::vector< l3k_bnd* > boundary_terms; // attached boundary equation kernels (not owned)
    l3k_ctx*            ctx;
    l3k_mesh*           mesh;
    int                 kernel_id, nq, n_rhs;
    l3k_kparams         kp;
    std::vector< char > blob;
    int                 field_inds[l3k::dev::max_unknowns];
    DevBuf< double >    tables;
    std::vector< double > tables_host;
    const double*       fields = nullptr;
    size_t              ldf    = 0;
    double              time   = 0.;
    bool                dense = false, fuse = false;
    double*             energy_target = nullptr; // l3k_mf_apply_energy: where the element kernel adds x^T A x (device)
    int                 energy_done   = 0;       // ... how many element launches did (the launcher counts)
    int                 energy_expected = 0;     // ... of how many non-empty ones: equal = fused, else the caller takes a dot product
    double*             ws = nullptr; // LocalAssembly workspace (grown on demand)
    size_t              ws_doubles = 0;
    // l3k_assemble_global: two halves of element-system buffers, a second stream and the events that order their reuse; kept
    // across calls (allocating gigabytes per call cost more than the pipeline saved)
    struct GlobalAsm
    {
        double*     buf[2]      = {nullptr, nullptr};
        size_t      doubles     = 0; // per half
        hipStream_t second      = nullptr;
        hipEvent_t  formed[2]   = {nullptr, nullptr}, consumed[2] = {nullptr, nullptr};
    } gasm;
    ~l3k_mf()
    {
        if (ws)
            (void)hipFree(ws);
        for (int k = 0; k < 2; ++k)
        {
            if (gasm.buf[k])
                (void)hipFree(gasm.buf[k]);
            if (gasm.formed[k])
                (void)hipEventDestroy(gasm.formed[k]);
            if (gasm.consumed[k])
                (void)hipEventDestroy(gasm.consumed[k]);
        }
        if (gasm.second)
            (void)hipStreamDestroy(gasm.second);
    }
};

// a boundary equation kernel on a list of element sides (assembleProblem(kernel, boundary_ids) of the reference)
struct l3k_bnd
{
    l3k_ctx*              ctx;
    l3k_mesh*             mesh;
    int                   kernel_id, nq, n_rhs;
    l3k_kparams           kp;
    std::vector< char >   blob;
    int                   field_inds[l3k::dev::max_unknowns];
    DevBuf< double >      tables;
    DevBuf< int64_t >     face_elem; // sides of interior elements first
    DevBuf< uint8_t >     face_side;
    int64_t               n_faces = 0, n_interior_faces = 0;
    // deterministic mode: the two classes of sides sorted by colour (sides of one colour share no node); det_ptr[cls][c] ..
    // det_ptr[cls][c + 1] = positions of colour c in face_elem / face_side
    std::vector< int64_t > det_ptr[2];
    const double*         fields = nullptr;
    size_t                ldf    = 0;
    double                time   = 0.;
};


namespace l3k::api
{
struct KernelMeta
{
    int         id;
    l3k_kparams kp;
    const char* name;
    size_t      bytes;
    bool        boundary = false;
};
// registered equation kernels (built in or announced by a plugin) / residual kernels by id; nullptr if unknown
const KernelMeta* findKernel(int id);
const KernelMeta* findResidual(int id);
} // namespace l3k::api
// api_assembled.hip: `count` element matrices from the tiled layout of l3k_local_assemble_tiled to the row-major one, on stream s
int launchTiledToRowMajor(int U, int N1, int64_t count, const double* d_Kt, double* d_K, hipStream_t s);
// ... and the upper triangles overwritten by the mirrored lower ones (bitwise symmetric matrices, as the reference's)
int launchTiledXToRowMajorSym(int U, int N1, int64_t count, const double* d_Kt, double* d_K, hipStream_t s);
// api_assembled.hip: the batch scatter of element systems into CSR values on a given stream
int launchAssembledScatter(l3k_mf* mf, int64_t first, int64_t count, const double* d_K, const double* d_F, const int64_t* d_row_ptr,
                           const int32_t* d_col_ind, double* d_values, double* d_rhs, size_t ldr, int skip_dirichlet,
                           unsigned long long* d_count, hipStream_t s, int tiled);
#endif
