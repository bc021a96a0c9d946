// diag.hpp -- diagonal of the element operators, diag(K_e)[b*U+u] = sum_q w detJ sum_e (B_q[e, b*U+u])^2, scattered
// into the global diagonal (precomputeDiagRhsImpl's `diagonal` term, algsys/EvaluateLocalOperator.hpp:185-186, and
// scatterInit, algsys/MatrixFreeSystem.hpp:377-390).  The reference loops over all (q, b) pairs (O(nq^3 n^3) per
// element); here the square is expanded,
//     (c0 phi + sum_d c_d dphi_d)^2 = sum_{k<=l} m_kl c_k c_l psi_k psi_l,   c_0 = A0[e][u], c_d = (sum_s A_s Ji[d][s])[e][u],
// and every psi_k psi_l is a tensor product of the 1-D tables I*I, I*D, D*D, so each of the 10 symmetric coefficient
// arrays G_kl(q,u) = w detJ sum_e c_k c_l is contracted with 3 transposed sweeps: O(10 * 3 * nq^3 n) per element.
#ifndef L3K_DEVICE_DIAG_HPP
#define L3K_DEVICE_DIAG_HPP

#include "sumfact_apply.hpp"
#include "sumfact_fast.hpp"

namespace l3k::dev
{
template < typename K, int P, int NQ >
constexpr size_t diagLdsBytes()
{
    constexpr int M = cmax(P + 1, NQ);
    return sizeof(double) * (size_t(4 * K::params.n_unknowns + 5 * K::params.n_fields + 13) * M * M * M + 24); // (+ 13: Ji, w detJ, point)
}

// GS: working set in the workgroup's slice of a.scratch instead of the LDS, persistent workgroups (sumfact_apply.hpp)
template < typename K, int P, int NQ, bool GS = false >
__global__ __launch_bounds__((applyThreads< P, NQ >())) void diagKernel(const ElemArgs a, const K kern)
{
    constexpr KernelParams params = K::params;
    constexpr int          U = params.n_unknowns, E = params.n_equations, F = params.n_fields;
    constexpr int          N1 = P + 1, NN = N1 * N1 * N1, NQP = NQ * NQ * NQ, M = cmax(N1, NQ), M3 = M * M * M;
    constexpr int          NT = applyThreads< P, NQ >();
    constexpr TableLayout  TL{N1, NQ};
    using Iface = KernelInterface< KernelParams{params.dimension, E, U, F, 1} >;

    extern __shared__ double lds[];
    double* const            Gb  = GS ? a.scratch + size_t(blockIdx.x) * (diagLdsBytes< K, P, NQ >() / sizeof(double)) : lds; // [U][M3] coefficient array of the current (k,l)
    double* const            T1  = Gb + U * M3;      // sweep temporaries
    double* const            T2  = T1 + U * M3;
    double* const            acc = T2 + U * M3;      // [U][M3] element diagonal at the nodes
    double* const            Fv  = acc + U * M3;     // fields at the QPs: values, then 3 reference derivatives, + 1 temp
    double* const            geo = Fv + 5 * F * M3;  // [13][M3]: J^-1 (9), w detJ, the point (3) at the quadrature points, formed once per element
    double* const            vs  = geo + 13 * M3;    // [8][3]

    const int       tid = threadIdx.x;
    const double*   tab[3] = {a.tables + TL.offII(), a.tables + TL.offID(), a.tables + TL.offDD()};
    int64_t         eb = blockIdx.x;
    do
    {
    const int64_t   e   = a.elem_begin + eb;
    const uint32_t* en  = a.elem_nodes + e * NN;

    if (tid < 24)
        vs[tid] = a.elem_verts[e * 24 + tid];
    for (int i = tid; i < U * M3; i += NT)
        acc[i] = 0.;
    if constexpr (F > 0)
    {
        double* const Ft = Fv + 4 * F * M3;
        for (int t = tid; t < NN * F; t += NT)
        {
            const int f = t / NN, i = t - f * NN;
            Fv[f * M3 + i] = a.fields[en[i] + f * a.ldf];
        }
        __syncthreads();
        const double* tabI = a.tables + TL.offI();
        const double* tabC = a.tables + TL.offC();
        sweep< 0, N1, NQ, false, false, N1, N1, N1, F, NT >(Fv, Ft, M3, tabI, tid);
        __syncthreads();
        sweep< 1, N1, NQ, false, false, NQ, N1, N1, F, NT >(Ft, Fv, M3, tabI, tid);
        __syncthreads();
        sweep< 2, N1, NQ, false, false, NQ, NQ, N1, F, NT >(Fv, Ft, M3, tabI, tid);
        __syncthreads();
        for (int i = tid; i < F * M3; i += NT)
            Fv[i] = Ft[i];
        __syncthreads();
        sweep< 0, NQ, NQ, false, false, NQ, NQ, NQ, F, NT >(Fv, Fv + 1 * F * M3, M3, tabC, tid);
        sweep< 1, NQ, NQ, false, false, NQ, NQ, NQ, F, NT >(Fv, Fv + 2 * F * M3, M3, tabC, tid);
        sweep< 2, NQ, NQ, false, false, NQ, NQ, NQ, F, NT >(Fv, Fv + 3 * F * M3, M3, tabC, tid);
    }
    __syncthreads();
    // the geometry of the element's quadrature points, once (the ten pair steps below read it back: 13 LDS loads instead of ~150
    // instructions with a division per pair and point)
    for (int q = tid; q < NQP; q += NT)
    {
        const int     qx = q % NQ, qy = (q / NQ) % NQ, qz = q / (NQ * NQ);
        const double* qw = a.tables + TL.offW();
        const double* qp = a.tables + TL.offX();
        double        G[6][3], Jm[3][3], Ji[3][3], xyz[3];
        hexPencilGeom(vs, qp[qy], qp[qz], G);
        hexPointOnPencil(G, qp[qx], Jm, xyz);
        const double wgt = qw[qx] * qw[qy] * qw[qz] * inverse3(Jm, Ji);
#pragma unroll
        for (int i = 0; i < 9; ++i)
            geo[i * M3 + q] = Ji[i / 3][i % 3];
        geo[9 * M3 + q] = wgt;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            geo[(10 + i) * M3 + q] = xyz[i];
    }
    __syncthreads();

    auto pairStep = [&]< int KK, int LL >() {
        // ---- G_kl at the quadrature points
        for (int q = tid; q < NQP; q += NT)
        {
            double Ji[3][3], xyz[3];
#pragma unroll
            for (int i = 0; i < 9; ++i)
                Ji[i / 3][i % 3] = geo[i * M3 + q];
            const double wgt = geo[9 * M3 + q];
#pragma unroll
            for (int i = 0; i < 3; ++i)
                xyz[i] = geo[(10 + i) * M3 + q];
            typename Iface::DomainInput in;
#pragma unroll
            for (int f = 0; f < F; ++f)
            {
                in.field_vals[f] = Fv[f * M3 + q];
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    in.field_ders[s][f] = Ji[0][s] * Fv[(1 * F + f) * M3 + q] + Ji[1][s] * Fv[(2 * F + f) * M3 + q] +
                                          Ji[2][s] * Fv[(3 * F + f) * M3 + q];
            }
            in.point = SpaceTimePoint{Point3{{xyz[0], xyz[1], xyz[2]}}, a.time};
            typename Iface::Result res{};
            kern(in, res);
            auto coef = [&](int k, int e_, int u) {
                return k == 0 ? res.operators[0](e_, u)
                              : res.operators[1](e_, u) * Ji[k - 1][0] + res.operators[2](e_, u) * Ji[k - 1][1] +
                                    res.operators[3](e_, u) * Ji[k - 1][2];
            };
#pragma unroll
            for (int u = 0; u < U; ++u)
            {
                double g = 0.;
#pragma unroll
                for (int e_ = 0; e_ < E; ++e_)
                    g += coef(KK, e_, u) * coef(LL, e_, u);
                Gb[u * M3 + q] = (KK == LL ? 1. : 2.) * wgt * g;
            }
        }
        __syncthreads();
        // ---- psi_k psi_l along one axis: I*I (neither differentiates that axis), I*D (one does), D*D (both)
        constexpr auto sel = [](int axis) { return (KK == axis + 1 ? 1 : 0) + (LL == axis + 1 ? 1 : 0); };
        sweep< 2, NQ, N1, true, false, NQ, NQ, NQ, U, NT >(Gb, T1, M3, tab[sel(2)], tid); // -> (NQ, NQ, N1)
        __syncthreads();
        sweep< 1, NQ, N1, true, false, NQ, NQ, N1, U, NT >(T1, T2, M3, tab[sel(1)], tid); // -> (NQ, N1, N1)
        __syncthreads();
        sweep< 0, NQ, N1, true, true, NQ, N1, N1, U, NT >(T2, acc, M3, tab[sel(0)], tid); // += (N1, N1, N1)
        __syncthreads();
    };
    pairStep.template operator()< 0, 0 >();
    pairStep.template operator()< 0, 1 >();
    pairStep.template operator()< 0, 2 >();
    pairStep.template operator()< 0, 3 >();
    pairStep.template operator()< 1, 1 >();
    pairStep.template operator()< 1, 2 >();
    pairStep.template operator()< 1, 3 >();
    pairStep.template operator()< 2, 2 >();
    pairStep.template operator()< 2, 3 >();
    pairStep.template operator()< 3, 3 >();

    // ---- scatterInit: add everywhere (Dirichlet rows are overwritten by the finalize pass, MatrixFreeSystem.hpp:911-915)
    for (int t = tid; t < NN * U; t += NT)
    {
        const int     i   = t / U;
        const int     u   = t - i * U;
        const int64_t dof = int64_t(en[i]) * a.dofs_per_node + a.field_inds[u];
        double*       dst = dof < a.n_owned_dofs ? a.diag + dof : a.diag_g + (dof - a.n_owned_dofs);
        unsafeAtomicAdd(dst, acc[u * M3 + i]);
    }
    if constexpr (GS)
        __syncthreads();
    } while (GS && (eb += gridDim.x) < a.elem_count);
}

// diag + rhs of one element range: the rhs part runs the sum-factorised apply in RHS mode (B^T W (f - B g_D))
template < typename K, int P, int NQ, int R >
int launchDiagRhs(const ElemArgs& a, const void* kparam_blob, hipStream_t stream)
{
    if (a.elem_count <= 0)
        return 0;
    // the right-hand side: the single-wave kernel's RHS variant where it applies (one column, dense dof layout, not a small
    // launch), else the generic kernel in RHS mode
    int rhs_rc = 1;
    if constexpr (R == 1 && FastCfg< K, P, NQ >::feasible)
        rhs_rc = launchSumfactFastRhs< K, P, NQ >(a, kparam_blob, stream);
    if (rhs_rc < 0)
        return rhs_rc;
    if (rhs_rc == 1)
        if (int rc = launchSumfactApply< K, P, NQ, R, true >(a, kparam_blob, stream))
            return rc;
    if (!a.diag)
        return 0;
    K kern{};
    if (kparam_blob)
        __builtin_memcpy(&kern, kparam_blob, sizeof(K));
    constexpr size_t lds = diagLdsBytes< K, P, NQ >();
    if constexpr (lds > lds_limit_bytes)
    {
        const int64_t  max_wgs = 2 * int64_t(deviceComputeUnits());
        const unsigned grid    = static_cast< unsigned >(a.elem_count < max_wgs ? a.elem_count : max_wgs);
        ElemArgs       ag      = a;
        ag.scratch             = a.scratch_alloc ? a.scratch_alloc(a.scratch_owner, lds * grid) : nullptr;
        if (!ag.scratch)
        {
            setError("could not obtain %zu bytes of global scratch for the diagonal kernel", lds * grid);
            return -3;
        }
        hipLaunchKernelGGL((diagKernel< K, P, NQ, true >), dim3(grid), dim3(applyThreads< P, NQ >()), 0, stream, ag, kern);
        const hipError_t err = hipGetLastError();
        if (err != hipSuccess)
        {
            setError("diagKernel (global scratch) launch failed: %s", hipGetErrorString(err));
            return -3;
        }
        return 0;
    }
    else
    {
    auto        kernel   = diagKernel< K, P, NQ >;
    static bool attr_set = false;
    if (!attr_set)
    {
        if (hipFuncSetAttribute(reinterpret_cast< const void* >(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)) !=
            hipSuccess)
        {
            setError("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu) failed", lds);
            return -3;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(kernel, dim3(static_cast< unsigned >(a.elem_count)), dim3(applyThreads< P, NQ >()), lds, stream, a, kern);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess)
    {
        setError("diagKernel launch failed: %s", hipGetErrorString(err));
        return -3;
    }
    return 0;
    }
}
} // namespace l3k::dev
#endif
