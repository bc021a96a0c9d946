// boundary.hpp -- boundary equation kernels on element sides (matrix-free term, diagonal and rhs).
//
// Reference: the BoundaryEquationKernel overloads of evaluateLocalOperator / precomputeOperatorDiagonalAndRhs
// (algsys/EvaluateLocalOperator.hpp:238-274,303-330) with map::mapBoundary (mapping/MapReferenceToPhysical.hpp:44-89,
// mapping/BoundaryNormal.hpp:8-64, mapping/BoundaryIntegralJacobian.hpp:9-29) and the side quadratures of
// basisfun/ReferenceElementBasisAtQuadrature.hpp:57-97.  The reference evaluates the full element basis (n^3 functions)
// at every side quadrature point; here the tensor structure is used: on a side the basis VALUES are those of the n^2
// side nodes only, the two tangential derivatives act inside the side plane, and the normal derivative is one
// contraction of the n^3 nodal values with phi_k'(+-1).  The set of side quadrature points (tensor Gauss rule on the two
// tangential axes) is the reference's; their order differs, sums agree to rounding.
//
// One side per workgroup.  Surface work is O(n^2) of the volume work, so this kernel is written for clarity, not peak.
#ifndef L3K_DEVICE_BOUNDARY_HPP
#define L3K_DEVICE_BOUNDARY_HPP

#include "sumfact_apply.hpp"

namespace l3k::dev
{
// side s of a hex (mesh/ElementTraits.hpp:84-95): 0 z-, 1 z+, 2 y-, 3 y+, 4 x-, 5 x+
struct SideAxes
{
    int    n, t1, t2; // normal axis and the two tangential axes (ascending)
    double nsign;     // outward normal = nsign * (dx/dxi_t1 x dx/dxi_t2) / |.|   (mapping/BoundaryNormal.hpp:39-61)
    int    upper;     // side at xi_n = +1
};
__device__ __forceinline__ SideAxes sideAxes(int side)
{
    SideAxes s;
    s.upper = side & 1;
    if (side < 2)
    {
        s.n = 2, s.t1 = 0, s.t2 = 1;
        s.nsign = s.upper ? 1. : -1.;
    }
    else if (side < 4)
    {
        s.n = 1, s.t1 = 0, s.t2 = 2;
        s.nsign = s.upper ? -1. : 1.;
    }
    else
    {
        s.n = 0, s.t1 = 1, s.t2 = 2;
        s.nsign = s.upper ? 1. : -1.;
    }
    return s;
}

// geometry of one side point: Jm[s][d] = dx_s/dxi_d, its inverse, the physical point, surface jacobian and normal
__device__ __forceinline__ double sidePointGeom(const double* __restrict__ vs, const SideAxes& sa, double c1, double c2,
                                                double Jm[3][3], double Ji[3][3], double xyz[3], double nrm[3])
{
    const double cn    = sa.upper ? 1. : -1.;
    const double xi[3] = {sa.n == 0 ? cn : c1, sa.n == 1 ? cn : (sa.n == 0 ? c1 : c2), sa.n == 2 ? cn : c2};
    double       G[6][3];
    hexPencilGeom(vs, xi[1], xi[2], G);
    hexPointOnPencil(G, xi[0], Jm, xyz);
    inverse3(Jm, Ji);
    double a[3], b[3];
#pragma unroll
    for (int s = 0; s < 3; ++s)
    {
        a[s] = sa.n == 0 ? Jm[s][1] : Jm[s][0];
        b[s] = sa.n == 2 ? Jm[s][1] : Jm[s][2];
    }
    const double cr[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    const double len   = sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
    const double inv   = sa.nsign / len;
#pragma unroll
    for (int s = 0; s < 3; ++s)
        nrm[s] = cr[s] * inv;
    return len;
}

template < typename K, int P, int NQ, int R, bool RHS_MODE >
constexpr size_t faceLdsDoubles()
{
    constexpr int U = K::params.n_unknowns, E = K::params.n_equations, F = K::params.n_fields, NF = U * R + F;
    constexpr int N1 = P + 1, NN = N1 * N1 * N1, N2 = N1 * N1, NQ2 = NQ * NQ;
    return size_t(NF) * NN + 2 * size_t(NF) * N2 + 4 * size_t(NF) * NQ2 + 24 + (RHS_MODE ? size_t(NQ2) * (1 + 4 * E * U) : 0);
}

// RHS_MODE == false: y += alpha * A_b x;   RHS_MODE == true: rhs += B_b^T W (f_b - B_b g), diag += diag(A_b)
template < typename K, int P, int NQ, int R, bool RHS_MODE >
__global__ __launch_bounds__(256) void faceKernel(const ElemArgs a, const K kern)
{
    constexpr KernelParams params = K::params;
    constexpr int          U = params.n_unknowns, E = params.n_equations, F = params.n_fields, OPS = U * R, NF = OPS + F;
    constexpr int          N1 = P + 1, NN = N1 * N1 * N1, N2 = N1 * N1, NQ2 = NQ * NQ, NT = 256;
    constexpr int          CS = 1 + 4 * E * U; // per-point coefficient record of the diagonal
    constexpr TableLayout  TL{N1, NQ};
    using Iface = KernelInterface< KernelParams{params.dimension, E, U, F, R} >;

    extern __shared__ double lds[];
    double* const            xs   = lds;                 // [NF][NN]
    double* const            pv   = xs + NF * NN;        // [2][NF][N2]
    double* const            q    = pv + 2 * NF * N2;    // [4][NF][NQ2]
    double* const            vs   = q + 4 * NF * NQ2;    // [8][3]
    double* const            coef = vs + 24;             // [NQ2][CS] (RHS_MODE)

    const int       tid  = threadIdx.x;
    const int64_t   f    = a.face_begin + blockIdx.x;
    const int64_t   e    = a.face_elem[f];
    const int       side = a.face_side[f];
    const SideAxes  sa   = sideAxes(side);
    const uint32_t* en   = a.elem_nodes + e * NN;
    const double*   tabI = a.tables + TL.offI();
    const double*   tabD = a.tables + TL.offD();
    const double*   tabE = a.tables + TL.offE() + sa.upper * N1; // phi_k'(+-1)
    const double*   qw   = a.tables + TL.offW();
    const double*   qp   = a.tables + TL.offX();
    const int       str[3] = {1, N1, N2};
    const int       st1 = str[sa.t1], st2 = str[sa.t2], stn = str[sa.n];
    const int       kface = sa.upper ? P : 0;

    if (tid < 24)
        vs[tid] = a.elem_verts[e * 24 + tid];
    // ---- gather (as sumfactApplyKernel)
    for (int t = tid; t < NN * U; t += NT)
    {
        const int     i    = t / U;
        const int     u    = t - i * U;
        const int64_t node = en[i];
        const int64_t dof  = node * a.dofs_per_node + a.field_inds[u];
        const bool    dir  = a.dirichlet != nullptr && a.dirichlet[dof] != 0;
#pragma unroll
        for (int r = 0; r < R; ++r)
        {
            double val;
            if constexpr (RHS_MODE)
                val = (dir && a.dirichlet_vals) ? a.dirichlet_vals[dof + a.ldg * r] : 0.;
            else
                val = dir ? 0. : (dof < a.n_owned_dofs ? a.x[dof + a.ldx * r] : a.xg[(dof - a.n_owned_dofs) + a.ldxg * r]);
            xs[(r * U + u) * NN + i] = val;
        }
    }
    if constexpr (F > 0)
        for (int t = tid; t < NN * F; t += NT)
        {
            const int fl = t / NN, i = t - fl * NN;
            xs[(OPS + fl) * NN + i] = a.fields[en[i] + fl * a.ldf];
        }
    __syncthreads();

    // ---- values and normal reference derivative on the side plane (i along t1, j along t2)
    for (int t = tid; t < NF * N2; t += NT)
    {
        const int     op = t / N2, ij = t - op * N2, j = ij / N1, i = ij - j * N1;
        const double* col = xs + op * NN + i * st1 + j * st2;
        double        dn  = 0.;
#pragma unroll
        for (int k = 0; k < N1; ++k)
            dn += tabE[k] * col[k * stn];
        pv[op * N2 + ij]             = col[kface * stn];
        pv[NF * N2 + op * N2 + ij]   = dn;
    }
    __syncthreads();
    // ---- to the side quadrature points: value, d/dxi_t1, d/dxi_t2, d/dxi_n
    for (int t = tid; t < NF * NQ2; t += NT)
    {
        const int     op = t / NQ2, ab = t - op * NQ2, qb = ab / NQ, qa = ab - qb * NQ;
        const double* p0 = pv + op * N2;
        const double* p1 = pv + NF * N2 + op * N2;
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
        q[(0 * NF + op) * NQ2 + ab] = val;
        q[(1 * NF + op) * NQ2 + ab] = d1;
        q[(2 * NF + op) * NQ2 + ab] = d2;
        q[(3 * NF + op) * NQ2 + ab] = dn;
    }
    __syncthreads();

    // planes of q holding the derivative along reference axis 0, 1, 2
    const int pl0 = sa.n == 0 ? 3 : 1, pl1 = sa.n == 1 ? 3 : (sa.n == 0 ? 1 : 2), pl2 = sa.n == 2 ? 3 : 2;
    // ---- quadrature points: one per thread
    for (int qi = tid; qi < NQ2; qi += NT)
    {
        const int qb = qi / NQ, qa = qi - qb * NQ;
        double    Jm[3][3], Ji[3][3], xyz[3], nrm[3];
        const double jac = sidePointGeom(vs, sa, qp[qa], qp[qb], Jm, Ji, xyz, nrm);
        const double wgt = qw[qa] * qw[qb] * jac;
        double       v[NF], dv[3][NF];
#pragma unroll
        for (int o = 0; o < NF; ++o)
        {
            v[o]     = q[(0 * NF + o) * NQ2 + qi];
            dv[0][o] = q[(pl0 * NF + o) * NQ2 + qi];
            dv[1][o] = q[(pl1 * NF + o) * NQ2 + qi];
            dv[2][o] = q[(pl2 * NF + o) * NQ2 + qi];
        }
        typename Iface::BoundaryInput in;
#pragma unroll
        for (int fl = 0; fl < F; ++fl)
        {
            in.field_vals[fl] = v[OPS + fl];
#pragma unroll
            for (int s = 0; s < 3; ++s)
                in.field_ders[s][fl] = Ji[0][s] * dv[0][OPS + fl] + Ji[1][s] * dv[1][OPS + fl] + Ji[2][s] * dv[2][OPS + fl];
        }
        in.point  = SpaceTimePoint{Point3{{xyz[0], xyz[1], xyz[2]}}, a.time};
        in.normal = {{nrm[0], nrm[1], nrm[2]}};
        typename Iface::Result res{};
        kern(in, res);
        // reference-space operator blocks D_d = sum_s A_{s+1} Ji[d][s]
        double Dm[3][E][U];
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int eq = 0; eq < E; ++eq)
#pragma unroll
                for (int u = 0; u < U; ++u)
                    Dm[d][eq][u] = res.operators[1](eq, u) * Ji[d][0] + res.operators[2](eq, u) * Ji[d][1] +
                                   res.operators[3](eq, u) * Ji[d][2];
        double r0[OPS], rd[3][OPS];
#pragma unroll
        for (int r = 0; r < R; ++r)
        {
            double tq[E];
#pragma unroll
            for (int eq = 0; eq < E; ++eq)
            {
                double acc = 0.;
#pragma unroll
                for (int u = 0; u < U; ++u)
                    acc += res.operators[0](eq, u) * v[r * U + u] + Dm[0][eq][u] * dv[0][r * U + u] +
                           Dm[1][eq][u] * dv[1][r * U + u] + Dm[2][eq][u] * dv[2][r * U + u];
                tq[eq] = RHS_MODE ? wgt * (res.rhs(eq, r) - acc) : wgt * acc;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
            {
                double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
#pragma unroll
                for (int eq = 0; eq < E; ++eq)
                {
                    a0 += res.operators[0](eq, u) * tq[eq];
                    a1 += Dm[0][eq][u] * tq[eq];
                    a2 += Dm[1][eq][u] * tq[eq];
                    a3 += Dm[2][eq][u] * tq[eq];
                }
                r0[r * U + u]    = a0;
                rd[0][r * U + u] = a1;
                rd[1][r * U + u] = a2;
                rd[2][r * U + u] = a3;
            }
        }
#pragma unroll
        for (int o = 0; o < OPS; ++o)
        {
            q[(0 * NF + o) * NQ2 + qi]   = r0[o];
            q[(pl0 * NF + o) * NQ2 + qi] = rd[0][o];
            q[(pl1 * NF + o) * NQ2 + qi] = rd[1][o];
            q[(pl2 * NF + o) * NQ2 + qi] = rd[2][o];
        }
        if constexpr (RHS_MODE)
        {
            double* c = coef + qi * CS;
            c[0]      = wgt;
#pragma unroll
            for (int eq = 0; eq < E; ++eq)
#pragma unroll
                for (int u = 0; u < U; ++u)
                {
                    // plane order: value, t1, t2, n
                    c[1 + (0 * E + eq) * U + u] = res.operators[0](eq, u);
                    c[1 + (1 * E + eq) * U + u] = sa.n == 0 ? Dm[1][eq][u] : Dm[0][eq][u];
                    c[1 + (2 * E + eq) * U + u] = sa.n == 2 ? Dm[1][eq][u] : Dm[2][eq][u];
                    c[1 + (3 * E + eq) * U + u] = sa.n == 0 ? Dm[0][eq][u] : (sa.n == 1 ? Dm[1][eq][u] : Dm[2][eq][u]);
                }
        }
    }
    __syncthreads();

    // ---- transposed: back to the side nodes (w0: through the basis values of the side nodes, w1: through phi_k'(+-1))
    for (int t = tid; t < OPS * N2; t += NT)
    {
        const int op = t / N2, ij = t - op * N2, j = ij / N1, i = ij - j * N1;
        double    w0 = 0., w1 = 0.;
        for (int qb = 0; qb < NQ; ++qb)
        {
            double s0 = 0., s2 = 0., sn = 0.;
#pragma unroll
            for (int qa = 0; qa < NQ; ++qa)
            {
                const int    ab = qa + NQ * qb;
                const double Ia = tabI[i * NQ + qa];
                s0 += Ia * q[(0 * NF + op) * NQ2 + ab] + tabD[i * NQ + qa] * q[(1 * NF + op) * NQ2 + ab];
                s2 += Ia * q[(2 * NF + op) * NQ2 + ab];
                sn += Ia * q[(3 * NF + op) * NQ2 + ab];
            }
            const double Ib = tabI[j * NQ + qb];
            w0 += Ib * s0 + tabD[j * NQ + qb] * s2;
            w1 += Ib * sn;
        }
        pv[op * N2 + ij]           = w0;
        pv[NF * N2 + op * N2 + ij] = w1;
    }
    __syncthreads();

    // ---- scatter-add over all n^3 nodes (the normal derivative couples every node of the element)
    for (int t = tid; t < NN * U; t += NT)
    {
        const int     i    = t / U;
        const int     u    = t - i * U;
        const int     c[3] = {i % N1, (i / N1) % N1, i / N2};
        const int     ci = sa.n == 0 ? c[1] : c[0], cj = sa.n == 2 ? c[1] : c[2], ck = sa.n == 0 ? c[0] : (sa.n == 1 ? c[1] : c[2]);
        const int     ij = ci + N1 * cj;
        const int64_t node = en[i];
        const int64_t dof  = node * a.dofs_per_node + a.field_inds[u];
        const bool    dir  = !RHS_MODE && a.dirichlet != nullptr && a.dirichlet[dof] != 0;
        if (!dir)
        {
#pragma unroll
            for (int r = 0; r < R; ++r)
            {
                const int    op  = r * U + u;
                const double val = (ck == kface ? pv[op * N2 + ij] : 0.) + tabE[ck] * pv[NF * N2 + op * N2 + ij];
                double* dst = dof < a.n_owned_dofs ? a.y + dof + a.ldy * r : a.yg + (dof - a.n_owned_dofs) + a.ldyg * r;
                unsafeAtomicAdd(dst, (RHS_MODE ? 1. : a.alpha) * val);
            }
        }
        if constexpr (RHS_MODE)
            if (a.diag)
            {
                // diag(A_b)[node, u] = sum_q w jac sum_eq (B_q[eq, (node, u)])^2
                double dsum = 0.;
                for (int qb = 0; qb < NQ; ++qb)
                    for (int qa = 0; qa < NQ; ++qa)
                    {
                        const double* cf = coef + (qa + NQ * qb) * CS;
                        const double  Ia = tabI[ci * NQ + qa], Ib = tabI[cj * NQ + qb];
                        const double  bv = ck == kface ? Ia * Ib : 0.;
                        const double  b1 = ck == kface ? tabD[ci * NQ + qa] * Ib : 0.;
                        const double  b2 = ck == kface ? Ia * tabD[cj * NQ + qb] : 0.;
                        const double  bn = Ia * Ib * tabE[ck];
                        double        sq = 0.;
#pragma unroll
                        for (int eq = 0; eq < E; ++eq)
                        {
                            const double B = cf[1 + (0 * E + eq) * U + u] * bv + cf[1 + (1 * E + eq) * U + u] * b1 +
                                             cf[1 + (2 * E + eq) * U + u] * b2 + cf[1 + (3 * E + eq) * U + u] * bn;
                            sq += B * B;
                        }
                        dsum += cf[0] * sq;
                    }
                double* dd = dof < a.n_owned_dofs ? a.diag + dof : a.diag_g + (dof - a.n_owned_dofs);
                unsafeAtomicAdd(dd, dsum);
            }
    }
}

template < typename K, int P, int NQ, int R, bool RHS_MODE >
int launchFace(const ElemArgs& a, const void* kparam_blob, hipStream_t stream)
{
    if (a.face_count <= 0)
        return 0;
    constexpr size_t lds = sizeof(double) * faceLdsDoubles< K, P, NQ, R, RHS_MODE >();
    static_assert(lds <= lds_limit_bytes, "side working set exceeds 160 KiB of LDS");
    K kern{};
    if (kparam_blob)
        __builtin_memcpy(&kern, kparam_blob, sizeof(K));
    auto        kernel   = faceKernel< K, P, NQ, R, RHS_MODE >;
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
    hipLaunchKernelGGL(kernel, dim3(static_cast< unsigned >(a.face_count)), dim3(256), lds, stream, a, kern);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess)
    {
        setError("faceKernel launch failed: %s", hipGetErrorString(err));
        return -3;
    }
    return 0;
}
} // namespace l3k::dev
#endif
