// integral.hpp -- integrals of residual kernels over elements / element sides (computeIntegral, computeNormL2).
//
// Reference: post/Integral.hpp:11-111 (evalElementIntegral, evalElementBoundaryIntegral, evalLocalIntegral) and
// post/NormL2.hpp:11-62 (squared residual, doubled quadrature orders).  The reference evaluates the full element basis
// at every quadrature point; here the fields are taken to the quadrature points with sum-factorised sweeps.  Norms use
// nq = 2p+1 points per direction (13 at p = 6), so the quadrature grid is streamed plane by plane: x and y
// interpolation once, then per z-plane the values, the z derivative (phi'), and the in-plane x / y derivatives by the
// collocation matrix on the Gauss points -- the LDS footprint is O(nq^2 n) instead of O(nq^3).
//
// One element (or side) per workgroup; each workgroup writes E partial sums, reduced in a fixed order by
// reducePartialsKernel in api_post.hip (bitwise reproducible results).
#ifndef L3K_DEVICE_INTEGRAL_HPP
#define L3K_DEVICE_INTEGRAL_HPP

#include "boundary.hpp"

namespace l3k::dev
{
inline constexpr int integral_threads = 256;

// block-wide sum of `vals[NV]` in a fixed order -> out[NV] written by thread 0
template < int NV >
__device__ __forceinline__ void blockReduceStore(double (&vals)[NV], double* scratch /*[NT]*/, double* out)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int v = 0; v < NV; ++v)
    {
        __syncthreads();
        scratch[tid] = vals[v];
        __syncthreads();
        for (int s = integral_threads / 2; s > 0; s >>= 1)
        {
            if (tid < s)
                scratch[tid] += scratch[tid + s];
            __syncthreads();
        }
        if (tid == 0)
            out[v] = scratch[0];
    }
}

template < typename K, int P, int NQ >
constexpr size_t integralLdsDoubles()
{
    constexpr int F = K::params.n_fields > 0 ? K::params.n_fields : 1, N1 = P + 1;
    return size_t(F) * N1 * N1 * N1 + size_t(F) * NQ * N1 * N1 + size_t(F) * NQ * NQ * N1 + 4 * size_t(F) * NQ * NQ + 24 +
           integral_threads;
}

template < typename K, int P, int NQ >
__global__ __launch_bounds__(integral_threads) void integralDomainKernel(const ElemArgs a, const K kern)
{
    constexpr KernelParams params = K::params;
    constexpr int          E = params.n_equations, F = params.n_fields, FA = F > 0 ? F : 1;
    constexpr int          N1 = P + 1, NN = N1 * N1 * N1, NQ2 = NQ * NQ, NT = integral_threads;
    constexpr TableLayout  TL{N1, NQ};
    using Iface = KernelInterface< params >;

    extern __shared__ double lds[];
    double* const            xs = lds;                    // [F][N1][N1][N1]   (ix fastest)
    double* const            A1 = xs + FA * NN;           // [F][N1 (z)][N1 (y)][NQ (x)]
    double* const            A2 = A1 + FA * NQ * N1 * N1; // [F][N1 (z)][NQ (y)][NQ (x)]
    double* const            pl = A2 + FA * NQ2 * N1;     // [4][F][NQ2]: value, d/dxi, d/deta, d/dzeta of the plane
    double* const            vs = pl + 4 * FA * NQ2;
    double* const            red = vs + 24;

    const int       tid  = threadIdx.x;
    const int64_t   e    = a.elem_begin + blockIdx.x;
    const uint32_t* en   = a.elem_nodes + e * NN;
    const double*   tabI = a.tables + TL.offI();
    const double*   tabD = a.tables + TL.offD();
    const double*   tabC = a.tables + TL.offC();
    const double*   qw   = a.tables + TL.offW();
    const double*   qp   = a.tables + TL.offX();

    if (tid < 24)
        vs[tid] = a.elem_verts[e * 24 + tid];
    for (int t = tid; t < NN * F; t += NT) // FieldAccess::fill, post/FieldAccess.hpp:21-30
    {
        const int fl = t / NN, i = t - fl * NN;
        xs[fl * NN + i] = a.fields[en[i] + fl * a.ldf];
    }
    __syncthreads();
    // x then y interpolation, field by field (source and destination have different per-field strides)
    for (int fl = 0; fl < F; ++fl)
        sweep< 0, N1, NQ, false, false, N1, N1, N1, 1, NT >(xs + fl * NN, A1 + fl * NQ * N1 * N1, 0, tabI, tid);
    __syncthreads();
    for (int fl = 0; fl < F; ++fl)
        sweep< 1, N1, NQ, false, false, NQ, N1, N1, 1, NT >(A1 + fl * NQ * N1 * N1, A2 + fl * NQ2 * N1, 0, tabI, tid);
    __syncthreads();

    double acc[E];
#pragma unroll
    for (int i = 0; i < E; ++i)
        acc[i] = 0.;
    for (int qz = 0; qz < NQ; ++qz)
    {
        // plane values and zeta derivative
        for (int t = tid; t < F * NQ2; t += NT)
        {
            const int     fl = t / NQ2, ab = t - fl * NQ2;
            const double* col = A2 + fl * NQ2 * N1 + ab;
            double        v = 0., dz = 0.;
#pragma unroll
            for (int k = 0; k < N1; ++k)
            {
                v += tabI[k * NQ + qz] * col[k * NQ2];
                dz += tabD[k * NQ + qz] * col[k * NQ2];
            }
            pl[(0 * FA + fl) * NQ2 + ab] = v;
            pl[(3 * FA + fl) * NQ2 + ab] = dz;
        }
        __syncthreads();
        // in-plane derivatives on the Gauss grid (collocation matrix, exact for nq >= p+1)
        for (int t = tid; t < F * NQ2; t += NT)
        {
            const int     fl = t / NQ2, ab = t - fl * NQ2, qb = ab / NQ, qa = ab - qb * NQ;
            const double* pv = pl + (0 * FA + fl) * NQ2;
            double        dx = 0., dy = 0.;
#pragma unroll
            for (int m = 0; m < NQ; ++m)
            {
                dx += tabC[m * NQ + qa] * pv[m + NQ * qb];
                dy += tabC[m * NQ + qb] * pv[qa + NQ * m];
            }
            pl[(1 * FA + fl) * NQ2 + ab] = dx;
            pl[(2 * FA + fl) * NQ2 + ab] = dy;
        }
        __syncthreads();
        for (int qi = tid; qi < NQ2; qi += NT)
        {
            const int qb = qi / NQ, qa = qi - qb * NQ;
            double    G[6][3], Jm[3][3], Ji[3][3], xyz[3];
            hexPencilGeom(vs, qp[qb], qp[qz], G);
            hexPointOnPencil(G, qp[qa], Jm, xyz);
            const double det = inverse3(Jm, Ji);
            const double wgt = qw[qa] * qw[qb] * qw[qz] * det;
            typename Iface::DomainInput in;
#pragma unroll
            for (int fl = 0; fl < F; ++fl)
            {
                in.field_vals[fl] = pl[(0 * FA + fl) * NQ2 + qi];
                const double d0 = pl[(1 * FA + fl) * NQ2 + qi], d1 = pl[(2 * FA + fl) * NQ2 + qi],
                             d2 = pl[(3 * FA + fl) * NQ2 + qi];
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    in.field_ders[s][fl] = Ji[0][s] * d0 + Ji[1][s] * d1 + Ji[2][s] * d2;
            }
            in.point = SpaceTimePoint{Point3{{xyz[0], xyz[1], xyz[2]}}, a.time};
            typename Iface::Rhs out{};
            kern(in, out);
#pragma unroll
            for (int i = 0; i < E; ++i)
                acc[i] += wgt * (a.square ? out[i] * out[i] : out[i]);
        }
        __syncthreads();
    }
    blockReduceStore< E >(acc, red, a.partial + int64_t(blockIdx.x) * E);
}

template < typename K, int P, int NQ >
constexpr size_t integralSideLdsDoubles()
{
    constexpr int F = K::params.n_fields > 0 ? K::params.n_fields : 1, N1 = P + 1;
    return size_t(F) * N1 * N1 * N1 + 2 * size_t(F) * N1 * N1 + 24 + integral_threads;
}

template < typename K, int P, int NQ >
__global__ __launch_bounds__(integral_threads) void integralSideKernel(const ElemArgs a, const K kern)
{
    constexpr KernelParams params = K::params;
    constexpr int          E = params.n_equations, F = params.n_fields, FA = F > 0 ? F : 1;
    constexpr int          N1 = P + 1, NN = N1 * N1 * N1, N2 = N1 * N1, NQ2 = NQ * NQ, NT = integral_threads;
    constexpr TableLayout  TL{N1, NQ};
    using Iface = KernelInterface< params >;

    extern __shared__ double lds[];
    double* const            xs  = lds;           // [F][NN]
    double* const            pv  = xs + FA * NN;  // [2][F][N2]
    double* const            vs  = pv + 2 * FA * N2;
    double* const            red = vs + 24;

    const int       tid  = threadIdx.x;
    const int64_t   f    = a.face_begin + blockIdx.x;
    const int64_t   e    = a.face_elem[f];
    const SideAxes  sa   = sideAxes(a.face_side[f]);
    const uint32_t* en   = a.elem_nodes + e * NN;
    const double*   tabI = a.tables + TL.offI();
    const double*   tabD = a.tables + TL.offD();
    const double*   tabE = a.tables + TL.offE() + sa.upper * N1;
    const double*   qw   = a.tables + TL.offW();
    const double*   qp   = a.tables + TL.offX();
    const int       str[3] = {1, N1, N2};
    const int       st1 = str[sa.t1], st2 = str[sa.t2], stn = str[sa.n];
    const int       kface = sa.upper ? P : 0;

    if (tid < 24)
        vs[tid] = a.elem_verts[e * 24 + tid];
    for (int t = tid; t < NN * F; t += NT)
    {
        const int fl = t / NN, i = t - fl * NN;
        xs[fl * NN + i] = a.fields[en[i] + fl * a.ldf];
    }
    __syncthreads();
    for (int t = tid; t < F * N2; t += NT)
    {
        const int     fl = t / N2, ij = t - fl * N2, j = ij / N1, i = ij - j * N1;
        const double* col = xs + fl * NN + i * st1 + j * st2;
        double        dn  = 0.;
#pragma unroll
        for (int k = 0; k < N1; ++k)
            dn += tabE[k] * col[k * stn];
        pv[fl * N2 + ij]           = col[kface * stn];
        pv[FA * N2 + fl * N2 + ij] = dn;
    }
    __syncthreads();
    double acc[E];
#pragma unroll
    for (int i = 0; i < E; ++i)
        acc[i] = 0.;
    for (int qi = tid; qi < NQ2; qi += NT)
    {
        const int qb = qi / NQ, qa = qi - qb * NQ;
        double    Jm[3][3], Ji[3][3], xyz[3], nrm[3];
        const double jac = sidePointGeom(vs, sa, qp[qa], qp[qb], Jm, Ji, xyz, nrm);
        const double wgt = qw[qa] * qw[qb] * jac;
        typename Iface::BoundaryInput in;
#pragma unroll
        for (int fl = 0; fl < F; ++fl)
        {
            const double* p0 = pv + fl * N2;
            const double* p1 = pv + FA * N2 + fl * N2;
            double        val = 0., d1 = 0., d2 = 0., dn = 0.;
            for (int j = 0; j < N1; ++j)
            {
                double s0 = 0., s1 = 0., sn = 0.;
#pragma unroll
                for (int i = 0; i < N1; ++i)
                {
                    const double Ia = tabI[i * NQ + qa];
                    s0 += Ia * p0[i + N1 * j];
                    s1 += tabD[i * NQ + qa] * p0[i + N1 * j];
                    sn += Ia * p1[i + N1 * j];
                }
                const double Ib = tabI[j * NQ + qb];
                val += Ib * s0;
                d1 += Ib * s1;
                d2 += tabD[j * NQ + qb] * s0;
                dn += Ib * sn;
            }
            const double r0 = sa.n == 0 ? dn : d1, r1 = sa.n == 1 ? dn : (sa.n == 0 ? d1 : d2), r2 = sa.n == 2 ? dn : d2;
            in.field_vals[fl] = val;
#pragma unroll
            for (int s = 0; s < 3; ++s)
                in.field_ders[s][fl] = Ji[0][s] * r0 + Ji[1][s] * r1 + Ji[2][s] * r2;
        }
        in.point  = SpaceTimePoint{Point3{{xyz[0], xyz[1], xyz[2]}}, a.time};
        in.normal = {{nrm[0], nrm[1], nrm[2]}};
        typename Iface::Rhs out{};
        kern(in, out);
#pragma unroll
        for (int i = 0; i < E; ++i)
            acc[i] += wgt * (a.square ? out[i] * out[i] : out[i]);
    }
    blockReduceStore< E >(acc, red, a.partial + int64_t(blockIdx.x) * E);
}

// computeValuesAtNodes (algsys/ComputeValuesAtNodes.hpp:371-448 domain, :508-594 boundary): the residual kernel evaluated
// AT THE NODES of an element side (or of the whole element): field values are the nodal values, reference derivatives
// come from the GLL differentiation matrix phi_b'(gll_q); equation e is accumulated into dof field_inds[e] of the node
// together with a contribution count; l3k_average_values divides (averageElementContributions :112-154).
template < typename K, int P, int NQ, bool SIDE >
__global__ __launch_bounds__(integral_threads) void valuesAtNodesKernel(const ElemArgs a, const K kern)
{
    constexpr KernelParams params = K::params;
    constexpr int          E = params.n_equations, F = params.n_fields, FA = F > 0 ? F : 1;
    constexpr int          N1 = P + 1, NN = N1 * N1 * N1, N2 = N1 * N1, NT = integral_threads;
    constexpr TableLayout  TL{N1, NQ};
    using Iface = KernelInterface< params >;

    extern __shared__ double lds[];
    double* const            xs = lds;            // [F][NN]
    double* const            vs = xs + FA * NN;   // [8][3]

    const int       tid  = threadIdx.x;
    const int64_t   f    = SIDE ? a.face_begin + blockIdx.x : 0;
    const int64_t   e    = SIDE ? a.face_elem[f] : a.elem_begin + blockIdx.x;
    const SideAxes  sa   = sideAxes(SIDE ? a.face_side[f] : 0);
    const uint32_t* en   = a.elem_nodes + e * NN;
    const double*   gll  = a.tables + TL.offG();
    const double*   tabG = a.tables + TL.offDG();
    if (tid < 24)
        vs[tid] = a.elem_verts[e * 24 + tid];
    for (int t = tid; t < NN * F; t += NT)
    {
        const int fl = t / NN, i = t - fl * NN;
        xs[fl * NN + i] = a.fields[en[i] + fl * a.ldf];
    }
    __syncthreads();
    const int str[3] = {1, N1, N2};
    for (int t = tid; t < (SIDE ? N2 : NN); t += NT)
    {
        int c[3];
        if constexpr (SIDE) // getSideNodeInds: the normal coordinate sits at the side's end
        {
            c[sa.t1] = t % N1;
            c[sa.t2] = t / N1;
            c[sa.n]  = sa.upper ? P : 0;
        }
        else
        {
            c[0] = t % N1, c[1] = (t / N1) % N1, c[2] = t / N2;
        }
        const int i = c[0] + N1 * c[1] + N2 * c[2];
        double    G[6][3], Jm[3][3], Ji[3][3], xyz[3], nrm[3] = {0., 0., 0.};
        if constexpr (SIDE)
            sidePointGeom(vs, sa, gll[c[sa.t1]], gll[c[sa.t2]], Jm, Ji, xyz, nrm);
        else
        {
            hexPencilGeom(vs, gll[c[1]], gll[c[2]], G);
            hexPointOnPencil(G, gll[c[0]], Jm, xyz);
            inverse3(Jm, Ji);
        }
        typename Iface::BoundaryInput in;
#pragma unroll
        for (int fl = 0; fl < F; ++fl)
        {
            const double* xf = xs + fl * NN;
            in.field_vals[fl] = xf[i];
            double dr[3] = {0., 0., 0.};
            for (int d = 0; d < 3; ++d)
                for (int b = 0; b < N1; ++b)
                    dr[d] += tabG[b * N1 + c[d]] * xf[i + (b - c[d]) * str[d]];
#pragma unroll
            for (int s = 0; s < 3; ++s)
                in.field_ders[s][fl] = Ji[0][s] * dr[0] + Ji[1][s] * dr[1] + Ji[2][s] * dr[2];
        }
        in.point  = SpaceTimePoint{Point3{{xyz[0], xyz[1], xyz[2]}}, a.time};
        in.normal = {{nrm[0], nrm[1], nrm[2]}};
        typename Iface::Rhs out{};
        kern(in, out);
        const int64_t node = en[i];
#pragma unroll
        for (int eq = 0; eq < E; ++eq)
        {
            const int64_t dof = node * a.dofs_per_node + a.field_inds[eq];
            unsafeAtomicAdd(a.node_sum + dof, out[eq]);
            unsafeAtomicAdd(a.node_count + dof, 1.);
        }
    }
}

template < typename K, int P, int NQ, bool SIDE >
int launchValuesAtNodes(const ElemArgs& a, const void* kparam_blob, hipStream_t stream)
{
    const int64_t count = SIDE ? a.face_count : a.elem_count;
    if (count <= 0)
        return 0;
    constexpr int    FA  = K::params.n_fields > 0 ? K::params.n_fields : 1;
    constexpr size_t lds = sizeof(double) * (size_t(FA) * (P + 1) * (P + 1) * (P + 1) + 24);
    K kern{};
    if (kparam_blob)
        __builtin_memcpy(&kern, kparam_blob, sizeof(K));
    hipLaunchKernelGGL((valuesAtNodesKernel< K, P, NQ, SIDE >), dim3(static_cast< unsigned >(count)), dim3(integral_threads), lds,
                       stream, a, kern);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess)
    {
        setError("valuesAtNodesKernel launch failed: %s", hipGetErrorString(err));
        return -3;
    }
    return 0;
}
// side = a.face_count > 0 selects the boundary form
template < typename K, int P, int NQ >
int launchValuesAtNodesAny(const ElemArgs& a, const void* kparam_blob, hipStream_t stream)
{
    return a.face_elem ? launchValuesAtNodes< K, P, NQ, true >(a, kparam_blob, stream)
                       : launchValuesAtNodes< K, P, NQ, false >(a, kparam_blob, stream);
}

template < typename K, int P, int NQ, bool SIDE >
int launchIntegral(const ElemArgs& a, const void* kparam_blob, hipStream_t stream)
{
    const int64_t count = SIDE ? a.face_count : a.elem_count;
    if (count <= 0)
        return 0;
    constexpr size_t lds = sizeof(double) * (SIDE ? integralSideLdsDoubles< K, P, NQ >() : integralLdsDoubles< K, P, NQ >());
    static_assert(lds <= lds_limit_bytes, "integral working set exceeds 160 KiB of LDS");
    K kern{};
    if (kparam_blob)
        __builtin_memcpy(&kern, kparam_blob, sizeof(K));
    auto kernel = [] {
        if constexpr (SIDE)
            return integralSideKernel< K, P, NQ >;
        else
            return integralDomainKernel< K, P, NQ >;
    }();
    static bool attr_set = false;
    if (!attr_set)
    {
        if (hipFuncSetAttribute(reinterpret_cast< const void* >(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                int(lds)) != hipSuccess)
        {
            setError("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu) failed", lds);
            return -3;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(kernel, dim3(static_cast< unsigned >(count)), dim3(integral_threads), lds, stream, a, kern);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess)
    {
        setError("integral kernel launch failed: %s", hipGetErrorString(err));
        return -3;
    }
    return 0;
}
} // namespace l3k::dev
#endif
