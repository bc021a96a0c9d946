// common.hpp -- shared declarations of the device layer (argument blocks, instance registry).
#ifndef L3K_DEVICE_COMMON_HPP
#define L3K_DEVICE_COMMON_HPP

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "l3k.h"
#include "l3k/kernel_interface.hpp"

// Ablation switches (tools/kbench.py) exist only in a library built with L3K_ABLATION=1 (python -m l3ster_amd.build
// with that variable set writes lib/libl3k_ablation.so); in the product build they fold away at compile time.
#ifdef L3K_ABLATION
#define L3K_DBG(a) ((a).dbg)
#else
#define L3K_DBG(a) 0
#endif

namespace l3k::dev
{
inline constexpr int max_unknowns = 8;

// 1-D tables of one (p, nq) pair in HBM, layout: I[n][nq] | C[nq][nq] | qw[nq] | qx[nq] | D[n][nq] | gll[n] |
// product tables II | ID | DD (diagonal kernel) after the
// even-odd tables of I (n->nq), C (nq->nq), I^T (nq->n), C^T (nq->nq): for W (nin x nout) with the symmetry
// W[nin-1-b][nout-1-q] = s*W[b][q] (s = +1 interpolation, -1 derivative) two row-major arrays
// We, Wo of ceil(nin/2) x ceil(nout/2) each (see host/tables.cpp:evenOddTables and device/sumfact_fast.hpp:sweepEO).
struct TableLayout
{
    int n, nq;
    constexpr int offI() const { return 0; }
    constexpr int offC() const { return n * nq; }
    constexpr int offW() const { return offC() + nq * nq; }
    constexpr int offX() const { return offW() + nq; }
    constexpr int offD() const { return offX() + nq; }
    constexpr int offG() const { return offD() + n * nq; }
    constexpr int hn() const { return (n + 1) / 2; }
    constexpr int hq() const { return (nq + 1) / 2; }
    constexpr int offEoI() const { return offG() + n; }                    // We | Wo, each hn x hq
    constexpr int offEoC() const { return offEoI() + 2 * hn() * hq(); }    // each hq x hq
    constexpr int offEoIt() const { return offEoC() + 2 * hq() * hq(); }   // each hq x hn
    constexpr int offEoCt() const { return offEoIt() + 2 * hq() * hn(); }  // each hq x hq
    constexpr int offII() const { return offEoCt() + 2 * hq() * hq(); }   // I*I, I*D, D*D elementwise, n x nq each
    constexpr int offID() const { return offII() + n * nq; }
    constexpr int offDD() const { return offID() + n * nq; }
    constexpr int offE() const { return offDD() + n * nq; } // phi_k'(-1) [n] | phi_k'(+1) [n] (boundary kernels)
    constexpr int offDG() const { return offE() + 2 * n; }  // phi_b'(gll_q), [b][q], n x n (values at the nodes)
    constexpr int offEoDt() const { return offDG() + n * n; } // even-odd tables of D^T (nq -> n, antisymmetric): We | Wo, each hq x hn
    constexpr int size() const { return offEoDt() + 2 * hq() * hn(); }
};

// Everything an element kernel needs; passed by value as the kernel argument (scalar loads).
struct ElemArgs
{
    const uint32_t* elem_nodes; // [n_elems][N]
    const double*   elem_verts; // [n_elems][8][3]
    const uint8_t*  dirichlet;  // [n_local_dofs] or nullptr
    const uint8_t*  elem_flags; // [n_elems] bit 0: element touches a Dirichlet dof, bit 1: affine element (one Jacobian)
    int64_t         exclusive_node_begin, exclusive_node_end; // nodes in [begin,end) belong to exactly one element
    const double*   tables;     // TableLayout (device)
    const double*   tables_host; // the same block in host memory (copied into the kernel arguments of the fast path)
    const double*   fields;     // SoA [F][ldf] or nullptr
    size_t          ldf;
    const double*   x;  // owned rows
    const double*   xg; // ghost rows (or nullptr)
    double*         y;
    double*         yg;
    size_t          ldx, ldxg, ldy, ldyg;
    int64_t         n_owned_dofs;
    int64_t         elem_begin, elem_count;
    double          alpha, beta, time;
    int             fuse_beta; // rows of exclusive nodes are written as alpha*A*x + beta*y by the element kernel
    double*         energy; // single-wave kernel: if set, x^T A x of the launch's elements is ADDED here (one atomic per wave)
    int*            energy_done; // HOST flag: set to 1 by the launcher that ran the energy-accumulating kernel variant
    uint32_t*       work_counters; // single-wave kernel: 8 batch counters, 128 bytes apart, zeroed by the launcher (or nullptr: static deal)
    const uint16_t* slot_tab;  // single-wave kernel: [ (p+1)^2 ][8] scatter slots of a lane's local nodes (objects.hpp:l3k_mesh)
    int             n_shell;   // slots [0, n_shell) are scattered with atomics, [n_shell, N) are exclusive nodes (plain stores)
    int             dofs_per_node;
    int             field_inds[max_unknowns];
    // diag/rhs mode
    const double* dirichlet_vals; // [n_local_dofs][R] (ld = ldg) or nullptr
    size_t        ldg;
    double*       diag;
    double*       diag_g;
    // local assembly
    double* K;         // [count][Nd][Nd] row-major or nullptr
    int     K_tiled;   // K in the tiled layout of l3k_local_assemble_tiled (all U x U blocks, no mirroring) instead of row-major
    double* F;         // [count][Nd][R] column-major per element (RHS-mode kernel with local_out)
    double* checksum;  // [count] or nullptr
    double* workspace; // per-QP coefficients of the batch + 1 trailing flag (degenerate element)
    int64_t elem_begin_out; // output slot of the first element of the batch
    int     local_out;      // RHS-mode kernel writes element-local F_e instead of scattering
    int     all_affine; // every element of the mesh is a parallelepiped (one Jacobian per element)
    int     n_cols;     // single-wave kernel, multi-column variant: columns applied per element pass (0 / 1: one)
    const l3k_tuning* tune; // HOST: the context's launch-route settings (read by the launchers; nullptr: defaults)
    // Shapes whose per-element buffers exceed the 160 KiB of LDS (e.g. the reference's NS3D benchmark kernel at order 4: 14 fields x
    // 8^3 points x 5 buffers = 286 KB) run the same kernels on a per-workgroup working set in GLOBAL memory: `scratch` is the device
    // arena (workgroup b uses [b * bytes, (b + 1) * bytes)), obtained by the launcher from the HOST callback (the context owns and
    // grows it)
    double* scratch;
    double* (*scratch_alloc)(void* owner, size_t bytes);
    void*   scratch_owner;
    int     ref_z0;     // applies pass z = 0 to the domain kernel (the reference's evalAtHexQPs, SumFactorization.hpp:732) instead of the true z
    int     dense; // dofs_per_node == n_unknowns and field_inds = identity: a node's unknowns are one contiguous vector
    int     dbg; // ablation switches for tools/kbench.py (env L3K_DEBUG_FLAGS): read only by L3K_ABLATION builds
    long long* stamps; // L3K_ABLATION builds with env L3K_STAMPS: per-stage cycle counters of workgroup 0 ([iteration][16])
    // boundary terms / integrals: element sides [face_begin, face_begin + face_count) of the list (device arrays)
    const int64_t* face_elem;
    const uint8_t* face_side;
    int64_t        face_begin, face_count;
    double*        partial; // integrals: [n_blocks][E] per-block partial sums
    int            square;  // integrate the squared residual (L2 norm)
    double*        node_sum;   // values at nodes: [n_local_dofs] accumulated kernel values ...
    double*        node_count; // ... and number of contributions
};

using LaunchFn = int (*)(const ElemArgs&, const void* kparam_blob, hipStream_t stream);
using RouteFn  = int (*)(const ElemArgs&, char* buf, size_t n); // describes the kernel a launch with these arguments takes

// the launch-route settings of a launch: the context's, or the defaults (host/registry.cpp)
const l3k_tuning& defaultTuning();
inline const l3k_tuning& tuneOf(const ElemArgs& a)
{
    return a.tune ? *a.tune : defaultTuning();
}

struct Instance
{
    int      kernel_id, order, nq, ncols;
    LaunchFn apply;
    LaunchFn diag_rhs;
    LaunchFn assemble;
    size_t   assemble_ws_doubles; // workspace doubles per element for `assemble`
    LaunchFn apply_cols = nullptr; // ncols == 1 instances: applies a.n_cols columns in one pass over the elements, or nullptr
    bool     assemble_tiled = false; // `assemble` can write the tiled layout (ElemArgs::K_tiled): the sum-factorised kernel fits
    RouteFn  route = nullptr; // text description of the kernel `apply` (or `apply_cols`, with a.n_cols > 1) launches
};

// boundary equation kernel on element sides (device/boundary.hpp)
struct BoundaryInstance
{
    int      kernel_id, order, nq, ncols;
    LaunchFn apply;    // y += alpha * A_b x
    LaunchFn diag_rhs; // diag += diag(A_b), rhs += B_b^T W (f_b - B_b g)
};
// residual kernel integral (device/integral.hpp); `blocks` = number of partial sums the launch writes
struct IntegralInstance
{
    int      residual_id, order, nq;
    LaunchFn domain, boundary;
    LaunchFn at_nodes; // computeValuesAtNodes (independent of nq; registered with every instance of the kernel / order)
};
void                    registerBoundaryInstance(const BoundaryInstance& inst);
const BoundaryInstance* findBoundaryInstance(int kernel_id, int order, int nq, int ncols);
void                    registerIntegralInstance(const IntegralInstance& inst);
const IntegralInstance* findIntegralInstance(int residual_id, int order, int nq);

// Kernel metadata of functors that are not compiled into libl3k.so: kernel plugins (shared libraries built from a
// user's functor by l3ster_amd/plugin.py, loaded with l3k_plugin_load) announce themselves here from static registrars.
struct PluginKernel
{
    int         id, kind; // kind 0 domain equation kernel, 1 boundary equation kernel, 2 residual kernel
    int         dimension, n_equations, n_unknowns, n_fields, n_rhs;
    const char* name;
    size_t      param_bytes;
};
void                registerPluginKernel(const PluginKernel& k);
const PluginKernel* findPluginKernel(int id, bool residual);

void            registerInstance(const Instance& inst);
const Instance* findInstance(int kernel_id, int order, int nq, int ncols);
int             instanceCount();
const Instance* instanceAt(int i);

template < typename K >
struct KernelId;
template < typename K >
struct ResidualId;

void setError(const char* fmt, ...);
// compute units of the current device (cached per device)
int deviceComputeUnits();
} // namespace l3k::dev
#endif
