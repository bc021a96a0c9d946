// sumfact_fast.hpp -- register-resident, software-pipelined sum-factorised apply for hex elements (single column).
//
// Same mathematics as sumfact_apply.hpp (collocation-derivative form of evalLocalOperatorSumFact,
// algsys/SumFactorization.hpp:882-917, fused with gatherSumFact / scatterSumFact, algsys/MatrixFreeSystem.hpp:421-537)
// re-organised for the CDNA4 execution model, where this kernel is FP64-VALU bound (about 250 fp64 flop per dof,
// SURVEY.md D9) with LDS store bandwidth (~85 B/clk/CU) as the second limiter:
//
//  * one "team" of M*M threads (M = max(n, nq)) per element, EB elements per workgroup so that the 64-wide waves are
//    ~96 % full (p = 6: 5 elements = 245 of 256 lanes); every thread owns ONE 1-D pencil of ALL fields in registers,
//    so each sweep is register-only FMAs with the even-odd decomposition (the reference's own flop cut,
//    algsys/SumFactorization.hpp:88-343: 37 instead of 49 instructions per 7-point pencil), and the LDS only carries
//    the pencil re-orientations (two buffers per element);
//  * the global gather goes straight to registers one element ahead (node ids two ahead) and stays in flight behind the
//    whole compute phase of the current element; one persistent workgroup per CU walks the element batches;
//  * the scatter uses plain stores for nodes touched by exactly one element (the element-internal nodes of the
//    reference's numbering, mesh/LocalMeshView.hpp:425-458) and f64 atomics only for shared nodes.
#ifndef L3K_DEVICE_SUMFACT_FAST_HPP
#define L3K_DEVICE_SUMFACT_FAST_HPP

#include "sumfact_apply.hpp"

namespace l3k::dev
{
// out[q] = sum_b in[b] * W[b][q] for a table with W[NIN-1-b][NOUT-1-q] = s*W[b][q] (s = ANTI ? -1 : +1), from the
// even-odd tables We | Wo (host/tables.cpp:evenOddTables).  ACC: out += result.
template < int NIN, int NOUT, bool ANTI, bool ACC >
__device__ __forceinline__ void sweepEO(const double (&in)[NIN], double (&out)[NOUT], const double* __restrict__ eo)
{
    constexpr int HI = NIN / 2, HO = NOUT / 2, RI = (NIN + 1) / 2, RO = (NOUT + 1) / 2;
    const double* We = eo;
    const double* Wo = eo + RI * RO;
    double        e[RI], o[HI > 0 ? HI : 1];
#pragma unroll
    for (int r = 0; r < HI; ++r)
    {
        e[r] = in[r] + in[NIN - 1 - r];
        o[r] = in[r] - in[NIN - 1 - r];
    }
    if constexpr (NIN % 2)
        e[HI] = in[HI];
#pragma unroll
    for (int q = 0; q < HO; ++q)
    {
        double A = 0., B = 0.;
#pragma unroll
        for (int r = 0; r < RI; ++r)
            A += e[r] * We[r * RO + q];
#pragma unroll
        for (int r = 0; r < HI; ++r)
            B += o[r] * Wo[r * RO + q];
        const double lo = A + B, hi = ANTI ? B - A : A - B;
        out[q]            = ACC ? out[q] + lo : lo;
        out[NOUT - 1 - q] = ACC ? out[NOUT - 1 - q] + hi : hi;
    }
    if constexpr (NOUT % 2)
    {
        double m = 0.;
        if constexpr (ANTI)
        {
#pragma unroll
            for (int r = 0; r < HI; ++r)
                m += o[r] * Wo[r * RO + HO];
        }
        else
        {
#pragma unroll
            for (int r = 0; r < RI; ++r)
                m += e[r] * We[r * RO + HO];
        }
        out[HO] = ACC ? out[HO] + m : m;
    }
}

// the 1-D tables the fast kernel needs, passed BY VALUE in the kernel arguments: kernarg memory is read with scalar
// loads (s_load), so every coefficient is an SGPR operand of the FMAs and costs no vector-memory or LDS traffic
template < int N1, int NQ >
struct FastTables
{
    static constexpr int HN = (N1 + 1) / 2, HQ = (NQ + 1) / 2;
    double               eoI[2 * HN * HQ], eoC[2 * HQ * HQ], eoIt[2 * HQ * HN], eoCt[2 * HQ * HQ], qw[NQ], qx[NQ];
};

// workgroup barrier that orders LDS traffic only: s_waitcnt lgkmcnt(0) + s_barrier.  __syncthreads() would also wait
// for vmcnt(0) and drain the next element's global loads that are deliberately left in flight.
__device__ __forceinline__ void ldsBarrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// A zero the compiler cannot see through, in an SGPR.  Table pointers are offset by a fresh one in every stage so that
// the coefficient loads (s_load from the kernel-argument segment) are re-issued next to their use instead of being
// hoisted out of the element loop all at once, which would need ~280 SGPRs and spill them to VGPR lanes.
__device__ __forceinline__ int opaqueZero()
{
    int z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
    return z;
}

template < typename K, int P, int NQ >
struct FastCfg
{
    static constexpr int N1 = P + 1, M = cmax(N1, NQ), TEAM = M * M, M3 = M * M * M;
    static constexpr int U = K::params.n_unknowns, F = K::params.n_fields, NF = U + F;
    // elements per workgroup: as many teams as fit 256 threads (p=6: 5 -> 245 lanes; p=4: 10 -> 250 lanes)
    static constexpr int    EB       = 256 / TEAM > 0 ? 256 / TEAM : 1;
    static constexpr int    NT       = ((EB * TEAM + 63) / 64) * 64;
    static constexpr size_t lds      = sizeof(double) * (size_t(EB) * (2 * NF * M3 + 24));
    static constexpr bool   feasible = lds <= 160 * 1024 && NT <= 1024 && NF * cmax(N1, NQ) <= 64;
};

template < typename K, int P, int NQ >
__global__ __launch_bounds__((FastCfg< K, P, NQ >::NT)) void sumfactFastKernel(const ElemArgs a, const K kern, int64_t n_batches,
                                                                                  const FastTables< P + 1, NQ > tab)
{
    using Cfg = FastCfg< K, P, NQ >;
    constexpr int         N1 = Cfg::N1, M3 = Cfg::M3, TEAM = Cfg::TEAM, EB = Cfg::EB;
    constexpr int         U = Cfg::U, F = Cfg::F, NF = Cfg::NF, NN = N1 * N1 * N1;

    extern __shared__ double lds[];
    const int                tid  = threadIdx.x;
    const int                team = tid / TEAM, l = tid - team * TEAM;
    const bool               live = team < EB;
    double* const            bufA = lds + size_t(team < EB ? team : 0) * (2 * NF * M3 + 24);
    double* const            bufB = bufA + NF * M3;
    double* const            vs   = bufB + NF * M3; // [8][3]

    const double* const eoI  = tab.eoI;
    const double* const eoC  = tab.eoC;
    const double* const eoIt = tab.eoIt;
    const double* const eoCt = tab.eoCt;
    const double* const qw   = tab.qw;
    const double* const qp   = tab.qx;

    // pencil coordinates of this thread in the different stage grids (a fastest)
    const int  i1 = l % N1, j1 = l / N1;   // (N1 x N1) grid: load / z-sweep / final z-sweep + scatter
    const bool on_nn = live && l < N1 * N1;
    const int  iq = l % N1, kq = l / N1;   // (N1 x NQ) grid: y-sweeps at node-x
    const bool on_nq = live && l < N1 * NQ;
    const int  qa = l % NQ, qb = l / NQ;   // (NQ x NQ) grid
    const bool on_qq = live && l < NQ * NQ;

    // ---- software pipeline state: node ids two batches ahead, x values one batch ahead
    uint32_t ids_cur[N1], ids_nxt[N1];
    double   xn[N1][U > 0 ? U : 1];
    double   fn[N1][F > 0 ? F : 1];
    uint32_t dm_nxt[N1];
    auto     elemOf = [&](int64_t batch) { return a.elem_begin + batch * EB + team; };
    auto     valid  = [&](int64_t batch) { return batch < n_batches && on_nn && (batch * EB + team) < a.elem_count; };
    auto     loadIds = [&](int64_t batch, uint32_t (&ids)[N1]) {
        if (valid(batch))
        {
            const uint32_t* en = a.elem_nodes + elemOf(batch) * NN + i1 + N1 * j1;
#pragma unroll
            for (int k = 0; k < N1; ++k)
                ids[k] = en[k * N1 * N1];
        }
    };
    auto loadX = [&](int64_t batch, const uint32_t (&ids)[N1]) {
        if (!valid(batch))
            return;
        const bool flagged = a.dirichlet != nullptr && a.elem_flags != nullptr && a.elem_flags[elemOf(batch)] != 0;
#pragma unroll
        for (int k = 0; k < N1; ++k)
        {
            const int64_t base = int64_t(ids[k]) * a.dofs_per_node;
            uint32_t      dm   = 0;
#pragma unroll
            for (int u = 0; u < U; ++u)
            {
                const int64_t dof = base + a.field_inds[u];
                xn[k][u] = (a.dbg & 2) ? double(dof) * 1e-9 : (dof < a.n_owned_dofs ? a.x[dof] : a.xg[dof - a.n_owned_dofs]);
                if (flagged)
                    dm |= uint32_t(a.dirichlet[dof] != 0) << u;
            }
            dm_nxt[k] = dm;
#pragma unroll
            for (int f = 0; f < F; ++f)
                fn[k][f] = a.fields[ids[k] + f * a.ldf];
        }
    };

    int64_t batch = blockIdx.x;
    loadIds(batch, ids_cur);
    loadIds(batch + gridDim.x, ids_nxt);
    loadX(batch, ids_cur);

    for (; batch < n_batches; batch += gridDim.x)
    {
        const bool act = (batch * EB + team) < a.elem_count; // this team has an element in this batch
        // ---- take over the prefetched data (gatherSumFact: Dirichlet dofs read as 0, MatrixFreeSystem.hpp:441-466)
        double   u0[N1][NF];
        uint32_t ids_sc[N1], dm_sc[N1];
        if (on_nn && act)
        {
#pragma unroll
            for (int k = 0; k < N1; ++k)
            {
                ids_sc[k] = ids_cur[k];
                dm_sc[k]  = dm_nxt[k];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    u0[k][u] = (dm_nxt[k] >> u) & 1u ? 0. : xn[k][u];
#pragma unroll
                for (int f = 0; f < F; ++f)
                    u0[k][U + f] = fn[k][f];
            }
        }
        if (live && act)
            for (int t = l; t < 24; t += TEAM)
                vs[t] = a.elem_verts[elemOf(batch) * 24 + t];
        // rotate the id pipeline and launch the next loads: they stay in flight behind this batch's compute
#pragma unroll
        for (int k = 0; k < N1; ++k)
            ids_cur[k] = ids_nxt[k];
        loadX(batch + gridDim.x, ids_cur);
        loadIds(batch + 2 * int64_t(gridDim.x), ids_nxt);

        // ---- S1: z interpolation in registers; T1: write [op][qz][j][i] into bufA
        if (on_nn && act)
        {
            const double* tI = eoI + opaqueZero();
#pragma unroll
            for (int o = 0; o < NF; ++o)
            {
                double in[N1], out[NQ];
#pragma unroll
                for (int k = 0; k < N1; ++k)
                    in[k] = u0[k][o];
                sweepEO< N1, NQ, false, false >(in, out, tI);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    bufA[o * M3 + (q * N1 + j1) * N1 + i1] = out[q];
            }
        }
        ldsBarrier();
        // ---- S2: y interpolation, thread (i, qz): bufA -> bufB [op][qz][qy][i]
        if (on_nq && act)
        {
            const double* tI = eoI + opaqueZero();
#pragma unroll
            for (int o = 0; o < NF; ++o)
            {
                double in[N1], out[NQ];
#pragma unroll
                for (int j = 0; j < N1; ++j)
                    in[j] = bufA[o * M3 + (kq * N1 + j) * N1 + iq];
                sweepEO< N1, NQ, false, false >(in, out, tI);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    bufB[o * M3 + (kq * NQ + q) * N1 + iq] = out[q];
            }
        }
        ldsBarrier();
        // ---- S3/S4: x interpolation + xi-derivative, thread (qy, qz) = (qa, qb); values to bufA [op][qz][qy][qx]
        double v[NQ][NF], dxi[NQ][NF];
        if (on_qq && act)
        {
            const double* tI = eoI + opaqueZero();
#pragma unroll
            for (int o = 0; o < NF; ++o)
            {
                double in[N1], out[NQ];
#pragma unroll
                for (int i = 0; i < N1; ++i)
                    in[i] = bufB[o * M3 + (qb * NQ + qa) * N1 + i];
                sweepEO< N1, NQ, false, false >(in, out, tI);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                {
                    v[q][o] = out[q];
                    bufA[o * M3 + (qb * NQ + qa) * NQ + q] = out[q];
                }
            }
            const double* tC = eoC + opaqueZero();
#pragma unroll
            for (int o = 0; o < NF; ++o)
            {
                double in[NQ], der[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    in[q] = v[q][o];
                sweepEO< NQ, NQ, true, false >(in, der, tC);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    dxi[q][o] = der[q];
            }
        }
        ldsBarrier();
        // ---- S5: eta-derivative, thread (qx, qz) = (qa, qb): y-pencils of bufA -> bufB
        if (on_qq && act)
        {
            const double* tC = eoC + opaqueZero();
#pragma unroll
            for (int o = 0; o < NF; ++o)
            {
                double in[NQ], out[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    in[q] = bufA[o * M3 + (qb * NQ + q) * NQ + qa];
                sweepEO< NQ, NQ, true, false >(in, out, tC);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    bufB[o * M3 + (qb * NQ + q) * NQ + qa] = out[q];
            }
        }
        ldsBarrier();
        // ---- S6: zeta-derivative in place in bufA, thread (qx, qy) = (qa, qb)
        if (on_qq && act)
        {
            const double* tC = eoC + opaqueZero();
#pragma unroll
            for (int o = 0; o < NF; ++o)
            {
                double in[NQ], out[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    in[q] = bufA[o * M3 + (q * NQ + qb) * NQ + qa];
                sweepEO< NQ, NQ, true, false >(in, out, tC);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    bufA[o * M3 + (q * NQ + qb) * NQ + qa] = out[q];
            }
        }
        ldsBarrier();
        // ---- quadrature points of the x-pencil (qy, qz) = (qa, qb): evalAtHexQPs, SumFactorization.hpp:707-753
        if (on_qq && act)
        {
            const double wyz = qw[qa] * qw[qb];
            double       G[6][3];
            hexPencilGeom(vs, qp[qa], qp[qb], G);
#pragma unroll
            for (int q = 0; q < NQ; ++q)
            {
                double vv[NF], dv[3][NF], r0[U], rd[3][U];
#pragma unroll
                for (int o = 0; o < NF; ++o)
                {
                    vv[o]    = v[q][o];
                    dv[0][o] = dxi[q][o];
                    dv[1][o] = bufB[o * M3 + (qb * NQ + qa) * NQ + q];
                    dv[2][o] = bufA[o * M3 + (qb * NQ + qa) * NQ + q];
                }
                qpStage< K, 1, false >(kern, G, qp[q], qw[q] * wyz, a.time, vv, dv, r0, rd);
#pragma unroll
                for (int o = 0; o < U; ++o)
                {
                    v[q][o]   = r0[o];
                    dxi[q][o] = rd[0][o];
                    bufB[o * M3 + (qb * NQ + qa) * NQ + q] = rd[1][o];
                    bufA[o * M3 + (qb * NQ + qa) * NQ + q] = rd[2][o];
                }
            }
        }
        ldsBarrier();
        // ---- S8: C^T along eta in place in bufB (thread (qx,qz)); S9: C^T along zeta in place in bufA (thread (qx,qy))
        if (on_qq && act)
        {
            const double* tCt = eoCt + opaqueZero();
#pragma unroll
            for (int o = 0; o < U; ++o)
            {
                double in[NQ], out[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    in[q] = bufB[o * M3 + (qb * NQ + q) * NQ + qa];
                sweepEO< NQ, NQ, true, false >(in, out, tCt);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    bufB[o * M3 + (qb * NQ + q) * NQ + qa] = out[q];
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    in[q] = bufA[o * M3 + (q * NQ + qb) * NQ + qa];
                sweepEO< NQ, NQ, true, false >(in, out, tCt);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    bufA[o * M3 + (q * NQ + qb) * NQ + qa] = out[q];
            }
        }
        ldsBarrier();
        // ---- x-pencil (qy,qz): w = r0 + C^T r1 + g2 + g3, then I^T along x -> h[ix]
        double h[N1][U];
        if (on_qq && act)
        {
            const double* tCt = eoCt + opaqueZero();
#pragma unroll
            for (int o = 0; o < U; ++o)
            {
                double r1[NQ], w[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                {
                    r1[q] = dxi[q][o];
                    w[q]  = v[q][o] + bufB[o * M3 + (qb * NQ + qa) * NQ + q] + bufA[o * M3 + (qb * NQ + qa) * NQ + q];
                }
                sweepEO< NQ, NQ, true, true >(r1, w, tCt);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    v[q][o] = w[q];
            }
            const double* tIt = eoIt + opaqueZero();
#pragma unroll
            for (int o = 0; o < U; ++o)
            {
                double w[NQ], out[N1];
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    w[q] = v[q][o];
                sweepEO< NQ, N1, false, false >(w, out, tIt);
#pragma unroll
                for (int i = 0; i < N1; ++i)
                    h[i][o] = out[i];
            }
        }
        if constexpr (N1 != NQ)
            ldsBarrier(); // rows of different length: finish all reads of bufB before re-laying it out
        if (on_qq && act)
        {
#pragma unroll
            for (int o = 0; o < U; ++o)
#pragma unroll
                for (int i = 0; i < N1; ++i)
                    bufB[o * M3 + (qb * NQ + qa) * N1 + i] = h[i][o];
        }
        ldsBarrier();
        // ---- I^T along y, thread (ix, qz) = (iq, kq): bufB [op][qz][qy][ix] -> bufA [op][qz][iy][ix]
        if (on_nq && act)
        {
            const double* tIt = eoIt + opaqueZero();
#pragma unroll
            for (int o = 0; o < U; ++o)
            {
                double in[NQ], out[N1];
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    in[q] = bufB[o * M3 + (kq * NQ + q) * N1 + iq];
                sweepEO< NQ, N1, false, false >(in, out, tIt);
#pragma unroll
                for (int j = 0; j < N1; ++j)
                    bufA[o * M3 + (kq * N1 + j) * N1 + iq] = out[j];
            }
        }
        ldsBarrier();
        // ---- I^T along z in registers, thread (ix, iy) = (i1, j1), then scatter (scatterSumFact, :494-537)
        if (on_nn && act)
        {
            double ye[N1][U];
            const double* tIt = eoIt + opaqueZero();
#pragma unroll
            for (int o = 0; o < U; ++o)
            {
                double in[NQ], out[N1];
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    in[q] = bufA[o * M3 + (q * N1 + j1) * N1 + i1];
                sweepEO< NQ, N1, false, false >(in, out, tIt);
#pragma unroll
                for (int k = 0; k < N1; ++k)
                    ye[k][o] = out[k];
            }
#pragma unroll
            for (int k = 0; k < N1; ++k)
            {
                const int64_t node      = ids_sc[k];
                const int64_t base      = node * a.dofs_per_node;
                const bool    exclusive = node >= a.exclusive_node_begin && node < a.exclusive_node_end;
#pragma unroll
                for (int o = 0; o < U; ++o)
                {
                    if ((dm_sc[k] >> o) & 1u)
                        continue;
                    const int64_t dof = base + a.field_inds[o];
                    const double  val = a.alpha * ye[k][o];
                    double*       dst = dof < a.n_owned_dofs ? a.y + dof : a.yg + (dof - a.n_owned_dofs);
                    if (a.dbg & 1)
                    {
                        if (val == 1.2345e300)
                            *dst = val;
                    }
                    else if (exclusive || (a.dbg & 16))
                        *dst += val; // touched by this element only: no atomic needed
                    else
                        unsafeAtomicAdd(dst, val);
                }
            }
        }
        ldsBarrier(); // bufA is rewritten by the next batch's T1
    }
}

template < typename K, int P, int NQ >
int launchSumfactFast(const ElemArgs& a, const void* kparam_blob, hipStream_t stream)
{
    using Cfg = FastCfg< K, P, NQ >;
    if (a.elem_count <= 0)
        return 0;
    K kern{};
    if (kparam_blob)
        __builtin_memcpy(&kern, kparam_blob, sizeof(K));
    auto        kernel   = sumfactFastKernel< K, P, NQ >;
    static int  n_cus    = 0;
    static bool attr_set = false;
    if (!attr_set)
    {
        if (hipFuncSetAttribute(reinterpret_cast< const void* >(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                int(Cfg::lds)) != hipSuccess)
        {
            setError("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu) failed", Cfg::lds);
            return -3;
        }
        int dev = 0;
        (void)hipGetDevice(&dev);
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess)
        {
            setError("hipGetDeviceProperties failed");
            return -3;
        }
        n_cus    = prop.multiProcessorCount;
        attr_set = true;
    }
    const int64_t n_batches      = (a.elem_count + Cfg::EB - 1) / Cfg::EB;
    const int     blocks_per_cu  = int((160 * 1024) / Cfg::lds) > 0 ? int((160 * 1024) / Cfg::lds) : 1;
    const int64_t max_blocks     = int64_t(n_cus) * blocks_per_cu;
    const unsigned grid          = static_cast< unsigned >(n_batches < max_blocks ? n_batches : max_blocks);
    constexpr TableLayout        TL{P + 1, NQ};
    FastTables< P + 1, NQ >      tab;
    const double*                th = a.tables_host;
    __builtin_memcpy(tab.eoI, th + TL.offEoI(), sizeof tab.eoI);
    __builtin_memcpy(tab.eoC, th + TL.offEoC(), sizeof tab.eoC);
    __builtin_memcpy(tab.eoIt, th + TL.offEoIt(), sizeof tab.eoIt);
    __builtin_memcpy(tab.eoCt, th + TL.offEoCt(), sizeof tab.eoCt);
    __builtin_memcpy(tab.qw, th + TL.offW(), sizeof tab.qw);
    __builtin_memcpy(tab.qx, th + TL.offX(), sizeof tab.qx);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(Cfg::NT), Cfg::lds, stream, a, kern, n_batches, tab);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess)
    {
        setError("sumfactFastKernel launch failed: %s", hipGetErrorString(err));
        return -3;
    }
    return 0;
}
} // namespace l3k::dev
#endif
