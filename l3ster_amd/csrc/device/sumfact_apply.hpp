// sumfact_apply.hpp -- matrix-free sum-factorised operator apply on hex elements (generic LDS-staged version).
//
// Computes, per element, y_e = sum_q w_q detJ_q B_q^T (B_q x_e) exactly as evalLocalOperatorSumFact of the reference
// (algsys/SumFactorization.hpp:882-917 = sumFactBackHex :469-504 + evalAtHexQPs :678-756 + sumFactForwardHex :784-814)
// fused with the gather / scatter of algsys/MatrixFreeSystem.hpp:421-467,494-537, but in the collocation-derivative
// form: values are interpolated to the Gauss points with 3 sweeps (I), then differentiated ON the Gauss grid with the
// nq x nq collocation matrix C (D = I*C exactly for nq >= p+1), i.e. 6 sweeps each way instead of 9.  The quadrature
// point stage inlines the user functor (zeros of A_i fold away at compile time).
//
// One element per workgroup; 5 LDS buffers of (U*R+F) * max(n,nq)^3 doubles; each sweep is a set of independent
// 1-D pencils (n_in values -> n_out values) held in registers, coefficients come from the scalar cache.
#ifndef L3K_DEVICE_SUMFACT_APPLY_HPP
#define L3K_DEVICE_SUMFACT_APPLY_HPP

#include "common.hpp"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <cstdio>
#include <utility>

namespace l3k::dev
{
constexpr int cmax(int a, int b)
{
    return a > b ? a : b;
}
constexpr int cmin(int a, int b)
{
    return a < b ? a : b;
}

// One family of 1-D contractions along AXIS of an array with dims (DI, DJ, DK) (i fastest), NOPS arrays `op_stride`
// apart: out[q] (+)= sum_b in[b] * W(b, q), W(b,q) = TRANS ? Mat[q*NIN + b] : Mat[b*NOUT + q].
template < int AXIS, int NIN, int NOUT, bool TRANS, bool ACC, int DI, int DJ, int DK, int NOPS, int NT >
__device__ __forceinline__ void
sweep(const double* __restrict__ src, double* __restrict__ dst, int op_stride, const double* __restrict__ Mat, int tid)
{
    constexpr int NA    = AXIS == 0 ? DJ : DI;
    constexpr int NB    = AXIS == 2 ? DJ : DK;
    constexpr int total = NA * NB * NOPS;
    for (int t = tid; t < total; t += NT)
    {
        const int op  = t / (NA * NB);
        const int rem = t - op * (NA * NB);
        const int b   = rem / NA;
        const int a   = rem - b * NA;
        int       soff, doff, sstr, dstr;
        if constexpr (AXIS == 0)
        {
            soff = NIN * (a + DJ * b);
            doff = NOUT * (a + DJ * b);
            sstr = dstr = 1;
        }
        else if constexpr (AXIS == 1)
        {
            soff = a + DI * NIN * b;
            doff = a + DI * NOUT * b;
            sstr = dstr = DI;
        }
        else
        {
            soff = doff = a + DI * b;
            sstr = dstr = DI * DJ;
        }
        const double* s = src + op * op_stride + soff;
        double*       d = dst + op * op_stride + doff;
        double        in[NIN];
#pragma unroll
        for (int i = 0; i < NIN; ++i)
            in[i] = s[i * sstr];
#pragma unroll
        for (int q = 0; q < NOUT; ++q)
        {
            double acc = ACC ? d[q * dstr] : 0.;
#pragma unroll
            for (int i = 0; i < NIN; ++i)
                acc += in[i] * (TRANS ? Mat[q * NIN + i] : Mat[i * NOUT + q]);
            d[q * dstr] = acc;
        }
    }
}

// Geometry of a tri-linear hex (8 vertices, v = i + 2j + 4k).  Along an x-pencil (eta, zeta fixed) the map is linear
// in xi: x = G0 + xi*G1, dx/dxi = G1, dx/deta = G2 + xi*G3, dx/dzeta = G4 + xi*G5 (each a 3-vector), so the 18 pencil
// coefficients are computed once and every quadrature point costs 9 FMAs.
__device__ __forceinline__ void hexPencilGeom(const double* __restrict__ vs /*[8][3]*/, double eta, double zeta, double G[6][3])
{
    const double le[2] = {.5 * (1. - eta), .5 * (1. + eta)}, lz[2] = {.5 * (1. - zeta), .5 * (1. + zeta)};
#pragma unroll
    for (int s = 0; s < 3; ++s)
    {
        // a[j][k] = mean over i, b[j][k] = half difference over i of the vertex coordinate
        double a[2][2], b[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 2; ++k)
            {
                const double c0 = vs[(0 + 2 * j + 4 * k) * 3 + s], c1 = vs[(1 + 2 * j + 4 * k) * 3 + s];
                a[j][k] = .5 * (c0 + c1);
                b[j][k] = .5 * (c1 - c0);
            }
        G[0][s] = (a[0][0] * le[0] + a[1][0] * le[1]) * lz[0] + (a[0][1] * le[0] + a[1][1] * le[1]) * lz[1];
        G[1][s] = (b[0][0] * le[0] + b[1][0] * le[1]) * lz[0] + (b[0][1] * le[0] + b[1][1] * le[1]) * lz[1];
        G[2][s] = .5 * ((a[1][0] - a[0][0]) * lz[0] + (a[1][1] - a[0][1]) * lz[1]);
        G[3][s] = .5 * ((b[1][0] - b[0][0]) * lz[0] + (b[1][1] - b[0][1]) * lz[1]);
        G[4][s] = .5 * ((a[0][1] - a[0][0]) * le[0] + (a[1][1] - a[1][0]) * le[1]);
        G[5][s] = .5 * ((b[0][1] - b[0][0]) * le[0] + (b[1][1] - b[1][0]) * le[1]);
    }
}
// Jm[s][d] = d x_s / d xi_d (the transpose convention of algsys/SumFactorization.hpp:716-725) and x at xi on the pencil
__device__ __forceinline__ void hexPointOnPencil(const double G[6][3], double xi, double Jm[3][3], double xyz[3])
{
#pragma unroll
    for (int s = 0; s < 3; ++s)
    {
        xyz[s]   = G[0][s] + xi * G[1][s];
        Jm[s][0] = G[1][s];
        Jm[s][1] = G[2][s] + xi * G[3][s];
        Jm[s][2] = G[4][s] + xi * G[5][s];
    }
}
// inverse + determinant of a 3x3 (cofactors), Ji = Jm^{-1}: Ji[d][s] = d xi_d / d x_s
__device__ __forceinline__ double inverse3(const double M[3][3], double Mi[3][3])
{
    const double c00 = M[1][1] * M[2][2] - M[1][2] * M[2][1];
    const double c01 = M[1][2] * M[2][0] - M[1][0] * M[2][2];
    const double c02 = M[1][0] * M[2][1] - M[1][1] * M[2][0];
    const double det = M[0][0] * c00 + M[0][1] * c01 + M[0][2] * c02;
    // 1/det: hardware reciprocal seed + two Newton steps (quadratic convergence from ~2^-26 to < 1 ulp-ish), ~6
    // instructions instead of the ~20 of the IEEE-exact division expansion
    double id = __builtin_amdgcn_rcp(det);
    id        = id * (2. - det * id);
    id        = id * (2. - det * id);
    Mi[0][0]         = c00 * id;
    Mi[1][0]         = c01 * id;
    Mi[2][0]         = c02 * id;
    Mi[0][1]         = (M[0][2] * M[2][1] - M[0][1] * M[2][2]) * id;
    Mi[1][1]         = (M[0][0] * M[2][2] - M[0][2] * M[2][0]) * id;
    Mi[2][1]         = (M[0][1] * M[2][0] - M[0][0] * M[2][1]) * id;
    Mi[0][2]         = (M[0][1] * M[1][2] - M[0][2] * M[1][1]) * id;
    Mi[1][2]         = (M[0][2] * M[1][0] - M[0][0] * M[1][2]) * id;
    Mi[2][2]         = (M[0][0] * M[1][1] - M[0][1] * M[1][0]) * id;
    return det;
}

// The quadrature-point stage (evalAtHexQPs, algsys/SumFactorization.hpp:707-753) for one point.
// v[op], dv[d][op]: values / REFERENCE derivatives of the operands (op = u + U*r) and, after them, the F fields.
// On return r0[op], rd[d][op] hold A0^T t and D_d^T t.  RHS_MODE: t = wgt * (f - B x) (rhs with Dirichlet lifting).
template < typename K, int R, bool RHS_MODE, int RT, int C0, bool ENERGY >
__device__ __forceinline__ void qpStageAt(const K& kern, const double (*Ji)[3], double det, const double* xyz, double w_ref, double time,
                                          const double* v, const double (*dv)[K::params.n_unknowns * R + K::params.n_fields], double* r0,
                                          double (*rd)[K::params.n_unknowns * R], double* energy, bool ref_z0);
// ENERGY: *energy += sum_e wgt * (B x)_e^2, this point's share of x^T A x.
template < typename K, int R, bool RHS_MODE, int RT = R, int C0 = 0, bool ENERGY = false >
__device__ __forceinline__ void qpStage(const K&      kern,
                                        const double (*G)[3],
                                        double        xi,
                                        double        w_ref,
                                        double        time,
                                        const double* v,
                                        const double (*dv)[K::params.n_unknowns * R + K::params.n_fields],
                                        double*       r0,
                                        double (*rd)[K::params.n_unknowns * R],
                                        double*       energy = nullptr,
                                        bool          ref_z0 = false)
{
    double Jm[3][3], Ji[3][3], xyz[3];
    hexPointOnPencil(G, xi, Jm, xyz);
    const double det = inverse3(Jm, Ji);
    qpStageAt< K, R, RHS_MODE, RT, C0, ENERGY >(kern, Ji, det, xyz, w_ref, time, v, dv, r0, rd, energy, ref_z0);
}
// the same with the geometry of the point given: Ji = (dx/dxi)^{-1} (Ji[d][s] = d xi_d / d x_s), det, position
template < typename K, int R, bool RHS_MODE, int RT, int C0, bool ENERGY >
__device__ __forceinline__ void qpStageAt(const K&      kern,
                                          const double (*Ji)[3],
                                          double        det,
                                          const double* xyz,
                                          double        w_ref,
                                          double        time,
                                          const double* v,
                                          const double (*dv)[K::params.n_unknowns * R + K::params.n_fields],
                                          double*       r0,
                                          double (*rd)[K::params.n_unknowns * R],
                                          double*       energy,
                                          bool          ref_z0)
{
    constexpr KernelParams params = K::params;
    constexpr int          U = params.n_unknowns, E = params.n_equations, F = params.n_fields, OPS = U * R;
    using Iface = KernelInterface< KernelParams{params.dimension, E, U, F, RT} >; // rhs is E x RT, columns C0..C0+R-1 used
    const double wgt = w_ref * det;

    typename Iface::DomainInput in;
#pragma unroll
    for (int f = 0; f < F; ++f)
    {
        in.field_vals[f] = v[OPS + f];
#pragma unroll
        for (int s = 0; s < 3; ++s) // algsys/SumFactorization.hpp:596-612
            in.field_ders[s][f] = Ji[0][s] * dv[0][OPS + f] + Ji[1][s] * dv[1][OPS + f] + Ji[2][s] * dv[2][OPS + f];
    }
    // The true point by default.  ref_z0 (l3k_ctx_set_reference_z0): the apply hands the kernel Point{x, y, 0.} as the
    // reference's evalAtHexQPs does (algsys/SumFactorization.hpp:732); diag / rhs follow the reference's
    // local-element path, which passes the true point (algsys/AssembleLocalSystem.hpp:229-230).  A kernel that does not read
    // point.space.z() compiles to the same code either way
    in.point = SpaceTimePoint{Point3{{xyz[0], xyz[1], (!RHS_MODE && ref_z0) ? 0. : xyz[2]}}, time};
    typename Iface::Result res{};
    kern(in, res);

    // D_d = sum_s A_{s+1} Ji[d][s]   (:736-738)
    double Dm[3][E][U];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int u = 0; u < U; ++u)
                Dm[d][e][u] = res.operators[1](e, u) * Ji[d][0] + res.operators[2](e, u) * Ji[d][1] +
                              res.operators[3](e, u) * Ji[d][2];
#pragma unroll
    for (int r = 0; r < R; ++r)
    {
        double t[E];
#pragma unroll
        for (int e = 0; e < E; ++e)
        {
            double acc = 0.;
#pragma unroll
            for (int u = 0; u < U; ++u)
                acc += res.operators[0](e, u) * v[r * U + u] + Dm[0][e][u] * dv[0][r * U + u] +
                       Dm[1][e][u] * dv[1][r * U + u] + Dm[2][e][u] * dv[2][r * U + u];
            t[e] = RHS_MODE ? wgt * (res.rhs(e, C0 + r) - acc) : wgt * acc;
            if constexpr (ENERGY)
                *energy += acc * t[e];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
#pragma unroll
            for (int e = 0; e < E; ++e)
            {
                a0 += res.operators[0](e, u) * t[e];
                a1 += Dm[0][e][u] * t[e];
                a2 += Dm[1][e][u] * t[e];
                a3 += Dm[2][e][u] * t[e];
            }
            r0[r * U + u]    = a0;
            rd[0][r * U + u] = a1;
            rd[1][r * U + u] = a2;
            rd[2][r * U + u] = a3;
        }
    }
}

template < int P, int NQ >
constexpr int applyThreads()
{
    constexpr int nqp = NQ * NQ * NQ, n3 = (P + 1) * (P + 1) * (P + 1);
    constexpr int m   = cmax(nqp, n3);
    int           nt  = ((m + 63) / 64) * 64;
    return nt > 512 ? 512 : nt;
}
template < typename K, int P, int NQ, int R >
constexpr size_t applyLdsBytes()
{
    constexpr int M = cmax(P + 1, NQ);
    return sizeof(double) * (5 * size_t(K::params.n_unknowns * R + K::params.n_fields) * M * M * M + 24);
}

// RHS_MODE == false: y += alpha * A x        (matrix-free apply)
// RHS_MODE == true : rhs += B^T W (f - B g)   with g = Dirichlet values on Dirichlet dofs, 0 elsewhere
//                    (precomputeDiagRhsImpl's rhs, algsys/EvaluateLocalOperator.hpp:187,195-207, in sum-factorised
//                    form; nothing is skipped in the scatter, as scatterInit :377-390)
// GS (global scratch): the five buffers do not fit the LDS -- the workgroup keeps them in its slice of a.scratch (global memory,
// served by the L2 / MALL) and walks the elements persistently.  Same code otherwise: __syncthreads() orders a workgroup's global
// accesses too (its waves share one CU and its L1).
template < typename K, int P, int NQ, int R, bool RHS_MODE, int RT = R, int C0 = 0, bool GS = false >
__global__ __launch_bounds__((applyThreads< P, NQ >())) void sumfactApplyKernel(const ElemArgs a, const K kern)
{
    constexpr KernelParams params = K::params;
    constexpr int          U = params.n_unknowns, F = params.n_fields, OPS = U * R, NF = OPS + F;
    constexpr int          N1 = P + 1, NN = N1 * N1 * N1, NQP = NQ * NQ * NQ, M = cmax(N1, NQ), M3 = M * M * M;
    constexpr int          NT = applyThreads< P, NQ >();
    constexpr TableLayout  TL{N1, NQ};

    extern __shared__ double lds[];
    double* const            B0 = GS ? a.scratch + size_t(blockIdx.x) * (applyLdsBytes< K, P, NQ, R >() / sizeof(double)) : lds;
    double* const            B1 = B0 + NF * M3;
    double* const            B2 = B1 + NF * M3;
    double* const            B3 = B2 + NF * M3;
    double* const            B4 = B3 + NF * M3;
    double* const            vs = B4 + NF * M3; // [8][3]

    const int            tid  = threadIdx.x;
    const double* const  tabI = a.tables + TL.offI();
    const double* const  tabC = a.tables + TL.offC();
    int64_t              eb   = blockIdx.x; // position in the launch's element range (GS: persistent workgroups)
    do
    {
    const int64_t        e    = a.elem_begin + eb;
    const uint32_t*      en   = a.elem_nodes + e * NN;

    if (tid < 24)
        vs[tid] = a.elem_verts[e * 24 + tid];

    // ---- gather (gatherSumFact, algsys/MatrixFreeSystem.hpp:421-467): node layout [op][iz][iy][ix]
    // lanes run over (node, unknown) pairs with the unknown fastest: the U dofs of a node are contiguous in x, and so
    // are the nodes of a face / of the element interior, so a wave reads long contiguous runs.
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
                val = (L3K_DBG(a) & 2) ? double(t) * 1e-3 : (dir ? 0. : (dof < a.n_owned_dofs ? a.x[dof + a.ldx * r] : a.xg[(dof - a.n_owned_dofs) + a.ldxg * r]));
            B0[(r * U + u) * M3 + i] = val;
        }
    }
    if constexpr (F > 0)
        for (int t = tid; t < NN * F; t += NT) // FieldAccess::fill, post/FieldAccess.hpp:21-30
        {
            const int f = t / NN, i = t - f * NN;
            B0[(OPS + f) * M3 + i] = a.fields[en[i] + f * a.ldf];
        }
    __syncthreads();

    // ---- interpolation to the Gauss points: x, y, z sweeps
    if (!(L3K_DBG(a) & 8))
    {
    sweep< 0, N1, NQ, false, false, N1, N1, N1, NF, NT >(B0, B1, M3, tabI, tid); // -> (NQ, N1, N1)
    __syncthreads();
    sweep< 1, N1, NQ, false, false, NQ, N1, N1, NF, NT >(B1, B0, M3, tabI, tid); // -> (NQ, NQ, N1)
    __syncthreads();
    sweep< 2, N1, NQ, false, false, NQ, NQ, N1, NF, NT >(B0, B1, M3, tabI, tid); // -> (NQ, NQ, NQ) values in B1
    __syncthreads();
    // ---- collocation derivatives on the Gauss grid
    sweep< 0, NQ, NQ, false, false, NQ, NQ, NQ, NF, NT >(B1, B2, M3, tabC, tid);
    sweep< 1, NQ, NQ, false, false, NQ, NQ, NQ, NF, NT >(B1, B3, M3, tabC, tid);
    sweep< 2, NQ, NQ, false, false, NQ, NQ, NQ, NF, NT >(B1, B4, M3, tabC, tid);
    __syncthreads();
    }

    // ---- quadrature points: one per thread
    if (!(L3K_DBG(a) & 4))
    for (int q = tid; q < NQP; q += NT)
    {
        const int qx = q % NQ, qy = (q / NQ) % NQ, qz = q / (NQ * NQ);
        double    v[NF], dv[3][NF], r0[OPS], rd[3][OPS];
#pragma unroll
        for (int o = 0; o < NF; ++o)
        {
            v[o]     = B1[o * M3 + q];
            dv[0][o] = B2[o * M3 + q];
            dv[1][o] = B3[o * M3 + q];
            dv[2][o] = B4[o * M3 + q];
        }
        const double* qw = a.tables + TL.offW();
        const double* qp = a.tables + TL.offX();
        double G[6][3];
        hexPencilGeom(vs, qp[qy], qp[qz], G);
        qpStage< K, R, RHS_MODE, RT, C0 >(kern, G, qp[qx], qw[qx] * qw[qy] * qw[qz], a.time, v, dv, r0, rd, nullptr, a.ref_z0 != 0);
#pragma unroll
        for (int o = 0; o < OPS; ++o)
        {
            B1[o * M3 + q] = r0[o];
            B2[o * M3 + q] = rd[0][o];
            B3[o * M3 + q] = rd[1][o];
            B4[o * M3 + q] = rd[2][o];
        }
    }
    __syncthreads();

    // ---- transposed collocation derivatives accumulate into the value array
    if (!(L3K_DBG(a) & 8))
    {
    sweep< 0, NQ, NQ, true, true, NQ, NQ, NQ, OPS, NT >(B2, B1, M3, tabC, tid);
    __syncthreads();
    sweep< 1, NQ, NQ, true, true, NQ, NQ, NQ, OPS, NT >(B3, B1, M3, tabC, tid);
    __syncthreads();
    sweep< 2, NQ, NQ, true, true, NQ, NQ, NQ, OPS, NT >(B4, B1, M3, tabC, tid);
    __syncthreads();
    // ---- transposed interpolation back to the nodes: z, y, x
    sweep< 2, NQ, N1, true, false, NQ, NQ, NQ, OPS, NT >(B1, B0, M3, tabI, tid); // -> (NQ, NQ, N1)
    __syncthreads();
    sweep< 1, NQ, N1, true, false, NQ, NQ, N1, OPS, NT >(B0, B1, M3, tabI, tid); // -> (NQ, N1, N1)
    __syncthreads();
    sweep< 0, NQ, N1, true, false, NQ, N1, N1, OPS, NT >(B1, B0, M3, tabI, tid); // -> (N1, N1, N1)
    __syncthreads();
    }

    // ---- scatter-add (scatterSumFact, algsys/MatrixFreeSystem.hpp:494-537; scatterInit :377-390 in RHS mode)
    for (int t = tid; t < NN * U; t += NT)
    {
        const int     i    = t / U;
        const int     u    = t - i * U;
        const int64_t node = en[i];
        const int64_t dof  = node * a.dofs_per_node + a.field_inds[u];
        const bool    dir  = !RHS_MODE && a.dirichlet != nullptr && a.dirichlet[dof] != 0;
        const bool    excl = !RHS_MODE && a.fuse_beta && node >= a.exclusive_node_begin && node < a.exclusive_node_end;
#pragma unroll
        for (int r = 0; r < R; ++r)
        {
            const double val = (RHS_MODE ? 1. : a.alpha) * B0[(r * U + u) * M3 + i];
            double* dst = dof < a.n_owned_dofs ? a.y + dof + a.ldy * r : a.yg + (dof - a.n_owned_dofs) + a.ldyg * r;
            if constexpr (RHS_MODE)
                if (a.local_out) // F_e of assembleLocalSystem: column-major [Nd][RT] per element, written once
                {
                    a.F[((e - a.elem_begin) + a.elem_begin_out) * int64_t(NN * U) * RT + (i * U + u) + int64_t(NN * U) * (C0 + r)] = val;
                    continue;
                }
            if (L3K_DBG(a) & 1)
            {
                if (val == 1.2345e300)
                    *dst = val;
            }
            else if (excl) // node of this element only: write alpha*A*x + beta*y (see l3k_mf_scale)
                *dst = (dir ? 0. : val) + (a.beta == 0. ? 0. : a.beta * *dst);
            else if (!dir)
            {
                if (L3K_DBG(a) & 16)
                    *dst = val;
                else
                    unsafeAtomicAdd(dst, val);
            }
        }
    }
    if constexpr (GS)
        __syncthreads(); // (the buffers are rewritten by the next element)
    } while (GS && (eb += gridDim.x) < a.elem_count);
}

inline constexpr size_t lds_limit_bytes = 160 * 1024; // LDS per CU on gfx950; one workgroup may use all of it

// the route of a launch through the generic kernel as text (l3k_mf_route)
template < typename K, int P, int NQ, int R >
int describeSumfactApply(const ElemArgs& a, char* buf, size_t n)
{
    constexpr bool   by_columns = applyLdsBytes< K, P, NQ, R >() > lds_limit_bytes && R > 1;
    constexpr size_t ws         = applyLdsBytes< K, P, NQ, by_columns ? 1 : R >();
    std::snprintf(buf, n, "sumfactApplyKernel<p=%d,nq=%d,U=%d,F=%d,R=%d>: generic kernel, %s, %zu B %s%s%s",
                  P, NQ, K::params.n_unknowns, K::params.n_fields, by_columns ? 1 : R,
                  ws > lds_limit_bytes ? "persistent workgroups on GLOBAL scratch (the element's buffers exceed the LDS)" : "one element per workgroup",
                  ws, ws > lds_limit_bytes ? "of global scratch per workgroup" : "LDS",
                  by_columns ? ", column by column (the R-column working set exceeds the LDS)" : "", a.dense ? "" : ", non-dense dof layout");
    return 0;
}

template < typename K, int P, int NQ, int R, bool RHS_MODE, int RT = R, int C0 = 0 >
int launchSumfactApply(const ElemArgs& a, const void* kparam_blob, hipStream_t stream);

// column C of an RT-column operand with the single-column kernel, then the remaining columns
template < typename K, int P, int NQ, bool RHS_MODE, int RT, int C >
int launchColumns(const ElemArgs& a, const void* kparam_blob, hipStream_t stream)
{
    if constexpr (C < RT)
    {
        ElemArgs c       = a;
        c.x              = a.x ? a.x + a.ldx * C : nullptr;
        c.xg             = a.xg ? a.xg + a.ldxg * C : nullptr;
        c.y              = a.y ? a.y + a.ldy * C : nullptr;
        c.yg             = a.yg ? a.yg + a.ldyg * C : nullptr;
        c.dirichlet_vals = a.dirichlet_vals ? a.dirichlet_vals + a.ldg * C : nullptr;
        if (int rc = launchSumfactApply< K, P, NQ, 1, RHS_MODE, RT, C >(c, kparam_blob, stream))
            return rc;
        return launchColumns< K, P, NQ, RHS_MODE, RT, C + 1 >(a, kparam_blob, stream);
    }
    else
        return 0;
}

template < typename K, int P, int NQ, int R, bool RHS_MODE, int RT, int C0 >
int launchSumfactApply(const ElemArgs& a, const void* kparam_blob, hipStream_t stream)
{
    if (a.elem_count <= 0)
        return 0;
    if constexpr (applyLdsBytes< K, P, NQ, R >() > lds_limit_bytes && R > 1)
    {
        // the R-column working set does not fit the LDS: run column by column with the single-column kernel, as the
        // reference does when fewer columns than n_rhs are passed (algsys/MatrixFreeSystem.hpp:1124-1138); the user
        // kernel still sees an E x R rhs and column c of it is used
        return launchColumns< K, P, NQ, RHS_MODE, R, 0 >(a, kparam_blob, stream);
    }
    else if constexpr (applyLdsBytes< K, P, NQ, R >() > lds_limit_bytes)
    {
        // one column and still too large for the LDS: the global-scratch variant, persistent workgroups (two per CU: no LDS, the
        // registers admit them), each on its own slice of the context's scratch arena
        K kern{};
        if (kparam_blob)
            __builtin_memcpy(&kern, kparam_blob, sizeof(K));
        constexpr size_t ws = applyLdsBytes< K, P, NQ, R >();
        const int64_t    max_wgs = 2 * int64_t(deviceComputeUnits());
        const unsigned   grid = static_cast< unsigned >(a.elem_count < max_wgs ? a.elem_count : max_wgs);
        if (!a.scratch_alloc)
        {
            setError("this shape needs %zu bytes per element in global scratch (its buffers exceed the LDS) and the caller gave no arena", ws);
            return -3;
        }
        ElemArgs ag = a;
        ag.scratch  = a.scratch_alloc(a.scratch_owner, ws * grid);
        if (!ag.scratch)
        {
            setError("could not allocate %zu bytes of global scratch for the element kernel", ws * grid);
            return -3;
        }
        hipLaunchKernelGGL((sumfactApplyKernel< K, P, NQ, R, RHS_MODE, RT, C0, true >), dim3(grid), dim3(applyThreads< P, NQ >()), 0, stream, ag, kern);
        const hipError_t err = hipGetLastError();
        if (err != hipSuccess)
        {
            setError("sumfactApplyKernel (global scratch) launch failed: %s", hipGetErrorString(err));
            return -3;
        }
        return 0;
    }
    else
    {
        K kern{};
        if (kparam_blob)
            __builtin_memcpy(&kern, kparam_blob, sizeof(K));
        constexpr size_t lds      = applyLdsBytes< K, P, NQ, R >();
        auto             kernel   = sumfactApplyKernel< K, P, NQ, R, RHS_MODE, RT, C0 >;
        static bool      attr_set = false;
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
        hipLaunchKernelGGL(kernel, dim3(static_cast< unsigned >(a.elem_count)), dim3(applyThreads< P, NQ >()), lds, stream,
                           a, kern);
        const hipError_t err = hipGetLastError();
        if (err != hipSuccess)
        {
            setError("sumfactApplyKernel launch failed: %s", hipGetErrorString(err));
            return -3;
        }
        return 0;
    }
}
} // namespace l3k::dev
#endif
