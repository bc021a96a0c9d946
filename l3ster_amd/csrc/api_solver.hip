// api_solver.hip -- Jacobi-preconditioned conjugate gradients of libl3k.so: fused vector kernels and the single-rank driver.
#include "objects.hpp"

namespace
{
// ---- fused vector kernels of the Jacobi-PCG iteration (solve/BelosSolvers.hpp:116-122 "Block CG" with one column +
// solve/NativePreconditioners.hpp:36-96).  Scalars live in a device array s: 0 <r,z>, 1 <p,Ap>, 2 <r,z> new, 3 <r,r>.
// Every dot product is a two-stage reduction in a fixed order (bitwise reproducible for a given grid).
constexpr int cg_threads = 256, cg_blocks = l3k_cg_blocks;
__device__ __forceinline__ double blockSum(double v, double* sh)
{
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int w = cg_threads / 2; w > 0; w >>= 1)
    {
        if (threadIdx.x < w)
            sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    return sh[0];
}
__global__ __launch_bounds__(cg_threads) void cgDotKernel(const double* __restrict__ u, const double* __restrict__ v, int64_t n,
                                                          double* __restrict__ partial)
{
    __shared__ double sh[cg_threads];
    double            acc = 0.;
    for (int64_t i = int64_t(blockIdx.x) * cg_threads + threadIdx.x; i < n; i += int64_t(gridDim.x) * cg_threads)
        acc += __builtin_nontemporal_load(u + i) * __builtin_nontemporal_load(v + i);
    const double t = blockSum(acc, sh);
    if (threadIdx.x == 0)
        partial[blockIdx.x] = t;
}
// s[dst0] = sum partial[0][:], s[dst1] = sum partial[1][:] (dst1 < 0: one row); shift != 0: s[0] = s[2] first
__global__ __launch_bounds__(cg_threads) void cgFinishKernel(const double* __restrict__ partial, int n_blocks, double* __restrict__ s,
                                                             int dst0, int dst1, int shift)
{
    __shared__ double sh[cg_threads];
    for (int row = 0; row < (dst1 >= 0 ? 2 : 1); ++row)
    {
        double acc = 0.;
        for (int i = threadIdx.x; i < n_blocks; i += cg_threads)
            acc += partial[row * n_blocks + i];
        __syncthreads();
        const double t = blockSum(acc, sh);
        if (threadIdx.x == 0)
            s[row == 0 ? dst0 : dst1] = t;
    }
    if (shift && threadIdx.x == 0)
        s[0] = s[2];
}
// The iteration keeps the PRECONDITIONED residual z = M^-1 r instead of r (Jacobi: r = z / minv element-wise), and x moves in the
// p pass: 9 vector passes per iteration instead of 11 (the vector kernels run at the HBM rate, so passes are what counts):
//   z pass:  alpha = s[0]/s[1]; z -= alpha minv Ap; partial <r, z>, <r, r> with r = z / minv     reads z, Ap, minv; writes z
//   p pass:  x += alpha p; beta = s[2]/s[0]; p = z + beta p                                       reads z, p, x;   writes p, x
// (round 2: x += alpha p and r -= alpha Ap in one pass over x, r, p, Ap, minv, then p = minv r + beta p over r, minv, p.)
// The same iterates in exact arithmetic; in floating point z is updated where r was (one rounding of minv * Ap more, one of
// minv * r less).  Rows with minv == 0 (a preconditioner zeroed on constrained dofs, l3k_jacobi_inverse with damping 0) are FROZEN:
// z = p = 0 there, x keeps its initial value, and -- since r cannot be recovered from z = 0 -- they are left out of <r, r>, i.e.
// the convergence test runs over the rows the iteration can change (include/l3k.h: l3k_pcg_solve).
__global__ __launch_bounds__(cg_threads) void cgUpdateZKernel(double* __restrict__ z, const double* __restrict__ ap,
                                                              const double* __restrict__ minv, int64_t n,
                                                              const double* __restrict__ s, double* __restrict__ partial)
{
    __shared__ double sh[cg_threads];
    const double      alpha = s[0] / s[1];
    double            rz = 0., rr = 0.;
    for (int64_t i = int64_t(blockIdx.x) * cg_threads + threadIdx.x; i < n; i += int64_t(gridDim.x) * cg_threads)
    {
        // (every array is streamed once and is far larger than the caches: non-temporal loads and stores, +3-6 % of HBM rate)
        const double m  = minv ? __builtin_nontemporal_load(minv + i) : 1.;
        const double zi = __builtin_nontemporal_load(z + i) - alpha * (m * __builtin_nontemporal_load(ap + i));
        const double ri = minv ? (m != 0. ? zi / m : 0.) : zi; // (0 / 0 on a frozen row would poison both sums)
        __builtin_nontemporal_store(zi, z + i);
        rz += ri * zi;
        rr += ri * ri;
    }
    const double a = blockSum(rz, sh);
    __syncthreads();
    const double b = blockSum(rr, sh);
    if (threadIdx.x == 0)
    {
        partial[blockIdx.x]             = a;
        partial[gridDim.x + blockIdx.x] = b;
    }
}
__global__ __launch_bounds__(cg_threads) void cgUpdatePXKernel(double* __restrict__ p, double* __restrict__ x, const double* __restrict__ z,
                                                               int64_t n, const double* __restrict__ s)
{
    const double alpha = s[0] / s[1], beta = s[2] / s[0];
    for (int64_t i = int64_t(blockIdx.x) * cg_threads + threadIdx.x; i < n; i += int64_t(gridDim.x) * cg_threads)
    {
        const double pi = __builtin_nontemporal_load(p + i);
        __builtin_nontemporal_store(__builtin_nontemporal_load(x + i) + alpha * pi, x + i);
        __builtin_nontemporal_store(__builtin_nontemporal_load(z + i) + beta * pi, p + i);
    }
}
// z = minv (b - r) (r holds A x0 on entry and z on return); p = z; partial <r, z>, <r, r>
__global__ __launch_bounds__(cg_threads) void cgInitKernel(double* __restrict__ r, const double* __restrict__ b,
                                                           double* __restrict__ p, const double* __restrict__ minv, int64_t n,
                                                           double* __restrict__ partial)
{
    __shared__ double sh[cg_threads];
    double            rz = 0., rr = 0.;
    for (int64_t i = int64_t(blockIdx.x) * cg_threads + threadIdx.x; i < n; i += int64_t(gridDim.x) * cg_threads)
    {
        const double m  = minv ? minv[i] : 1.;
        const double ri = m != 0. ? __builtin_nontemporal_load(b + i) - __builtin_nontemporal_load(r + i) : 0.; // (frozen rows: out of the residual norm from the start, as in the z pass)
        const double zi = m * ri;
        __builtin_nontemporal_store(zi, r + i);
        __builtin_nontemporal_store(zi, p + i);
        rz += ri * zi;
        rr += ri * ri;
    }
    const double a = blockSum(rz, sh);
    __syncthreads();
    const double c = blockSum(rr, sh);
    if (threadIdx.x == 0)
    {
        partial[blockIdx.x]             = a;
        partial[gridDim.x + blockIdx.x] = c;
    }
}
// NativeJacobiImpl::init (solve/NativePreconditioners.hpp:75-96): sign(d) * damping / max(|d|, threshold)
__global__ void jacobiInverseKernel(const double* __restrict__ d, int64_t n, double damping, double threshold, double* __restrict__ out)
{
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
    {
        const double v = d[i], a = fabs(v);
        out[i]         = (v < 0. ? -damping : damping) / (a > threshold ? a : threshold);
    }
}
inline int cgGrid(int64_t n)
{
    const int64_t g = (n + cg_threads - 1) / cg_threads;
    return int(g < 1 ? 1 : (g > cg_blocks ? cg_blocks : g));
}
} // namespace

extern "C" {

// ------------------------------------------------------------------------------------------------ Jacobi-PCG
static int cgWorkspace(l3k_ctx* ctx)
{
    if (!ctx->red_ws) // (allocated by l3k_ctx_create on the context's device)
    {
        setError("context without reduction workspace");
        return -3;
    }
    return 0;
}
int l3k_jacobi_inverse(l3k_ctx* ctx, const double* d_diag, int64_t n, double damping, double threshold, double* d_minv)
{
    if (!ctx || (n > 0 && (!d_diag || !d_minv)))
    {
        setError("l3k_jacobi_inverse: null argument");
        return -1;
    }
    if (n > 0)
        hipLaunchKernelGGL(jacobiInverseKernel, dim3(gridFor(n)), dim3(256), 0, ctx->stream, d_diag, n, damping, threshold, d_minv);
    L3K_HIP(hipGetLastError());
    return 0;
}
int l3k_cg_init(l3k_ctx* ctx, double* d_r, const double* d_b, double* d_p, const double* d_minv, int64_t n, double* d_s)
{
    if (!ctx || !d_r || !d_b || !d_p || !d_s)
    {
        setError("l3k_cg_init: null argument");
        return -1;
    }
    if (int rc = cgWorkspace(ctx))
        return rc;
    const int g = cgGrid(n);
    hipLaunchKernelGGL(cgInitKernel, dim3(g), dim3(cg_threads), 0, ctx->stream, d_r, d_b, d_p, d_minv, n, ctx->red_ws);
    hipLaunchKernelGGL(cgFinishKernel, dim3(1), dim3(cg_threads), 0, ctx->stream, ctx->red_ws, g, d_s, 2, 3, 1);
    L3K_HIP(hipGetLastError());
    return 0;
}
int l3k_cg_dot_pap(l3k_ctx* ctx, const double* d_p, const double* d_ap, int64_t n, double* d_s)
{
    if (!ctx || !d_p || !d_ap || !d_s)
    {
        setError("l3k_cg_dot_pap: null argument");
        return -1;
    }
    if (int rc = cgWorkspace(ctx))
        return rc;
    const int g = cgGrid(n);
    hipLaunchKernelGGL(cgDotKernel, dim3(g), dim3(cg_threads), 0, ctx->stream, d_p, d_ap, n, ctx->red_ws);
    hipLaunchKernelGGL(cgFinishKernel, dim3(1), dim3(cg_threads), 0, ctx->stream, ctx->red_ws, g, d_s, 1, -1, 0);
    L3K_HIP(hipGetLastError());
    return 0;
}
int l3k_cg_update_z(l3k_ctx* ctx, double* d_z, const double* d_ap, const double* d_minv, int64_t n, double* d_s)
{
    if (!ctx || !d_z || !d_ap || !d_s)
    {
        setError("l3k_cg_update_z: null argument");
        return -1;
    }
    if (int rc = cgWorkspace(ctx))
        return rc;
    const int g = cgGrid(n);
    hipLaunchKernelGGL(cgUpdateZKernel, dim3(g), dim3(cg_threads), 0, ctx->stream, d_z, d_ap, d_minv, n, d_s, ctx->red_ws);
    hipLaunchKernelGGL(cgFinishKernel, dim3(1), dim3(cg_threads), 0, ctx->stream, ctx->red_ws, g, d_s, 2, 3, 0);
    L3K_HIP(hipGetLastError());
    return 0;
}
int l3k_cg_update_px(l3k_ctx* ctx, double* d_p, double* d_x, const double* d_z, int64_t n, double* d_s)
{
    if (!ctx || !d_p || !d_x || !d_z || !d_s)
    {
        setError("l3k_cg_update_px: null argument");
        return -1;
    }
    const int g = cgGrid(n);
    hipLaunchKernelGGL(cgUpdatePXKernel, dim3(g), dim3(cg_threads), 0, ctx->stream, d_p, d_x, d_z, n, d_s);
    // <r,z> of this iteration becomes the old one: after every block has read alpha and beta
    hipLaunchKernelGGL(cgFinishKernel, dim3(1), dim3(cg_threads), 0, ctx->stream, ctx->red_ws, 0, d_s, 4, -1, 1);
    L3K_HIP(hipGetLastError());
    return 0;
}
int l3k_pcg_solve(l3k_mf* mf, const double* d_b, double* d_x, const double* d_minv, const l3k_cg_opts* opts,
                  l3k_cg_result* result)
{
    if (!mf || !d_b || !d_x || !result)
    {
        setError("l3k_pcg_solve: null argument");
        return -1;
    }
    if (mf->mesh->n_ghost_nodes != 0)
    {
        setError("l3k_pcg_solve is the single-rank solver; partitioned systems iterate with the l3k_cg_* pieces and an "
                 "all-reduce of the scalar block between them (l3ster_amd/solve.py)");
        return -1;
    }
    const l3k_cg_opts o = opts ? *opts : l3k_cg_opts{1e-6, 10000, 0, 1};
    l3k_ctx*          ctx = mf->ctx;
    hipStream_t       st  = ctx->stream;
    const int64_t     n   = mf->mesh->nOwnedDofs();
    DevBuf< double >  work; // z (the preconditioned residual, in the array named r) | p | ap | s[8]
    work.n = size_t(3 * n + 8);
    L3K_HIP(hipMalloc(reinterpret_cast< void** >(&work.ptr), work.n * sizeof(double)));
    double *r = work.ptr, *p = r + n, *ap = p + n, *s = ap + n;
    double  h[4];
    const auto scalars = [&]() -> int {
        L3K_HIP(hipMemcpyAsync(h, s, sizeof h, hipMemcpyDeviceToHost, st));
        L3K_HIP(hipStreamSynchronize(st));
        return 0;
    };
    // z = M^-1 (b - A x0), p = z
    if (int rc = l3k_mf_apply(mf, d_x, size_t(n), r, size_t(n), 1, 1., 0.))
        return rc;
    if (int rc = l3k_cg_init(ctx, r, d_b, p, d_minv, n, s))
        return rc;
    double scale = 1.;
    if (o.residual_scaling == 2)
    {
        if (int rc = l3k_cg_dot_pap(ctx, d_b, d_b, n, s)) // s[1] = <b, b> (scratch use of the slot)
            return rc;
    }
    if (int rc = scalars())
        return rc;
    const double rr0 = std::sqrt(h[3]);
    if (o.residual_scaling == 1)
        scale = rr0 > 0. ? rr0 : 1.;
    else if (o.residual_scaling == 2)
        scale = std::sqrt(h[1]) > 1e-300 ? std::sqrt(h[1]) : 1e-300;
    double res = rr0 / scale;
    int    it  = 0;
    const int every = o.check_every > 0 ? o.check_every : 1;
    while (res > o.tol && it < o.max_iters)
    {
        if (int rc = l3k_mf_apply_energy(mf, p, ap, s)) // ap = A p, s[1] = <p, A p>
            return rc;
        if (int rc = l3k_cg_update_z(ctx, r, ap, d_minv, n, s)) // (r holds z = M^-1 r)
            return rc;
        if (int rc = l3k_cg_update_px(ctx, p, d_x, r, n, s))
            return rc;
        ++it;
        if (it % every == 0 || it == o.max_iters)
        {
            if (int rc = scalars())
                return rc;
            res = std::sqrt(h[3]) / scale;
        }
    }
    result->achieved_tol = res;
    result->iterations   = it;
    result->converged    = res <= o.tol;
    return 0;
}
// the same for a multivector of right-hand sides (the reference's systems carry n_rhs columns: Belos "Block CG" with block size 1
// iterates them one after the other, solve/BelosSolvers.hpp:116-122): column c of d_b / d_x at + c * ld; results[ncols]
int l3k_pcg_solve_cols(l3k_mf* mf, const double* d_b, size_t ldb, double* d_x, size_t ldx, int ncols, const double* d_minv,
                       const l3k_cg_opts* opts, l3k_cg_result* results)
{
    if (!mf || !d_b || !d_x || !results || ncols < 1)
    {
        setError("l3k_pcg_solve_cols: bad argument");
        return -1;
    }
    const size_t n = size_t(mf->mesh->nOwnedDofs());
    if (ncols > 1 && (ldb < n || ldx < n))
    {
        setError("l3k_pcg_solve_cols: leading dimension smaller than the number of owned dofs");
        return -1;
    }
    for (int c = 0; c < ncols; ++c)
        if (int rc = l3k_pcg_solve(mf, d_b + ldb * c, d_x + ldx * c, d_minv, opts, results + c))
            return rc;
    return 0;
}
} // extern "C"
