// sumfact_plane.hpp -- the sum-factorised apply with the quadrature stage STREAMED PLANE BY PLANE (round 3 experiment for
// occupancy: VERDICT r2 item 1).  Same mathematics and the same gather / scatter as sumfact_fast.hpp; what differs is what is
// alive at the quadrature-point stage (evalAtHexQPs, algsys/SumFactorization.hpp:678-756):
//
//   sumfact_fast.hpp : v, d/dxi of all fields as x-pencils in registers (112 VGPRs at U = 4, order 6) + d/deta, d/dzeta of
//                      all points in LDS (2 x 10.7 KB)                                  -> 7 waves per CU, 252 VGPRs
//   here             : v of all points in LDS (10.7 KB); per x-plane qx = k: d/deta, d/dzeta of the plane's 49 points formed
//                      just in time by 2 n U "pencil lanes" (one 1-D sweep each, scalar coefficients) into a 3 KB plane buffer,
//                      d/dxi of the plane by a dot product over the x-pencil re-read from LDS, the transposed x-sweep
//                      accumulated into a register x-pencil w                           -> 14-16 KB of LDS, w = 56 VGPRs
//
// Price: the x-direction loses the even-odd saving (2 x 28 instead of 2 x 19.4 instructions per point), four LDS round trips
// per plane instead of one per stage.  One element per wave, n == nq, no external fields, ghost rows behind the owned rows.
#ifndef L3K_DEVICE_SUMFACT_PLANE_HPP
#define L3K_DEVICE_SUMFACT_PLANE_HPP

#include "sumfact_fast.hpp"

#ifndef L3K_PLANE_MIN_WAVES
#define L3K_PLANE_MIN_WAVES 2
#endif

namespace l3k::dev
{
template < int N1, int NQ >
struct PlaneTables
{
    static constexpr int HN = (N1 + 1) / 2, HQ = (NQ + 1) / 2;
    double               eoI[2 * HN * HQ], eoC[2 * HQ * HQ], eoIt[2 * HQ * HN], eoCt[2 * HQ * HQ];
    double               Ct[NQ * NQ]; // Ct[k][m] = C[m][k]: row k = the coefficients of plane k (d/dxi there, and its transpose)
    double               qw[NQ], qx[NQ];
};

template < typename K, int P, int NQ >
struct PlaneCfg
{
    static constexpr int    N1 = P + 1, M = N1, MM = M * M, NN = N1 * N1 * N1;
    static constexpr int    U = K::params.n_unknowns, F = K::params.n_fields, NG = U / 2;
    static constexpr int    PS = MM, OS = PS * M;
    static constexpr int    A_D    = 2 * NG * OS;  // the element's array (doubles): interpolation stages, v, result staging
    static constexpr int    P_D    = 2 * U * MM;   // plane buffer: d/deta | d/dzeta (then r_eta | r_zeta, then their transposed sweeps)
    static constexpr int    IDS_D  = (NN + 1) / 2; // node ids in slot order (uint32)
    static constexpr int    SLOT_B = 16 * N1 * N1;
    static constexpr size_t lds    = sizeof(double) * size_t(A_D + P_D + IDS_D) + SLOT_B;
    static constexpr int    SG     = 64 / U * U;
    static constexpr int    NSH    = NN - (N1 - 2) * (N1 - 2) * (N1 - 2);
    static constexpr bool   feasible = N1 == NQ && F == 0 && U % 2 == 0 && U >= 2 && MM > 32 && MM <= 64 && 2 * M * U <= 64 && N1 <= 8 && lds <= 64 * 1024;
    static constexpr int    waves_by_lds = int((160 * 1024) / (lds > 0 ? lds : 1));
};

template < typename K, int P, int NQ >
__global__ __launch_bounds__(64, L3K_PLANE_MIN_WAVES) void sumfactPlaneKernel(const ElemArgs a, const K kern, int64_t n_batches, int xcd_chunk,
                                                                              const PlaneTables< P + 1, NQ > tab)
{
    using Cfg = PlaneCfg< K, P, NQ >;
    constexpr int M = Cfg::M, MM = Cfg::MM, PS = Cfg::PS, OS = Cfg::OS, U = Cfg::U, NG = Cfg::NG, NN = Cfg::NN;
    constexpr int HQ = (NQ + 1) / 2;
    auto          at = [](int c, int b, int a_) { return b * PS + a_ * M + c; }; // (x, y, z index) -> 16-byte unit, as sumfact_fast.hpp

    extern __shared__ double lds[];
    const int                lane = threadIdx.x;
    const int                l    = lane;
    const bool               worker = l < MM;
    constexpr int            SG     = Cfg::SG;
    double2* const           bufA   = reinterpret_cast< double2* >(lds);
    double* const            Ad     = lds;
    double* const            Pb     = lds + Cfg::A_D;
    uint32_t* const          idsL   = reinterpret_cast< uint32_t* >(lds + Cfg::A_D + Cfg::P_D);
    uint4* const             slotRows = reinterpret_cast< uint4* >(lds + Cfg::A_D + Cfg::P_D + Cfg::IDS_D);

    const double* const eoI  = tab.eoI;
    const double* const eoC  = tab.eoC;
    const double* const eoIt = tab.eoIt;
    const double* const eoCt = tab.eoCt;
    const double* const Ctp  = tab.Ct;
    const double* const qw   = tab.qw;
    const double* const qp   = tab.qx;

    const int qa = l % M, qb = l / M; // this lane's pencil in every grid (n == nq): (first, second) running index
    auto      ldg = [&](const double2* buf, int g, int idx) { return buf[g * OS + idx]; };
    auto      stg = [&](double2* buf, int g, int idx, double x0, double x1) { buf[g * OS + idx] = make_double2(x0, x1); };

    // ---- pencil role: lane t < 2 M U owns one in-plane pencil of one field: direction (0: along y, 1: along z), the other
    // in-plane index, field.  Its v values sit in the element's array at pbase + j * pstride (+ 2 k for plane k, in doubles); its
    // results go to the plane buffer, direction-major, one M x M array per field: [dir][field][j + M * pidx] -- the y
    // derivatives indexed by qy + M qz (= the point lane), the z derivatives by qz + M qy: both orders are conflict-free for
    // the pencil lanes' 8-byte stores and for the point lanes' loads
    const bool pl_ok  = lane < 2 * M * U;
    const int  pdir   = lane >= M * U ? 1 : 0;
    const int  ptt    = lane - pdir * M * U;
    const int  pidx   = ptt % M, pf = ptt / M;
    const int  pbase  = ((pf >> 1) * OS + (pdir ? pidx * PS : pidx * M)) * 2 + (pf & 1);
    const int  pstride = (pdir ? M : PS) * 2;
    const int  ppbase = (pdir * U + pf) * MM + M * pidx;
    const int  lt     = qb + M * qa; // this point lane in the z-derivative arrays

    // ---- persistent waves, dynamic batches by XCD chunk: as sumfact_fast.hpp
    const int  nb      = static_cast< int >(n_batches);
    const int  by_xcd  = xcd_chunk > 0;
    const int  stride  = by_xcd ? int(gridDim.x) >> 3 : int(gridDim.x);
    const int  first   = by_xcd ? int(blockIdx.x & 7) * xcd_chunk : 0;
    const int  last    = by_xcd ? (first + xcd_chunk < nb ? first + xcd_chunk : nb) : nb;
    const bool dyn     = a.work_counters != nullptr;
    const bool sharded = by_xcd || (gridDim.x & 7u) == 0;
    int        victim = sharded ? int(blockIdx.x & 7u) : 0, switches = 0;
    auto       vBase  = [&](int v) { return by_xcd ? v * xcd_chunk : v; };
    auto       vLimit = [&](int v) { return by_xcd ? (v * xcd_chunk + xcd_chunk < nb ? v * xcd_chunk + xcd_chunk : nb) : nb; };
    const int  dyn_step = by_xcd || !sharded ? 1 : 8;
    auto       drawTicket = [&]() -> uint32_t {
        uint32_t t = 0;
        if (dyn && lane == 0)
            t = __hip_atomic_fetch_add(a.work_counters + 32 * victim, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return t;
    };
    auto ticketBatch = [&](uint32_t t) {
        int b = vBase(victim) + int(__builtin_amdgcn_readfirstlane(t)) * dyn_step;
        while (b >= vLimit(victim))
        {
            if (!by_xcd || ++switches > 7)
                return nb;
            victim = (victim + 1) & 7;
            b      = vBase(victim) + int(__builtin_amdgcn_readfirstlane(drawTicket())) * dyn_step;
        }
        return b;
    };
    const int lim   = dyn ? nb : last;
    int       batch = dyn ? ticketBatch(drawTicket()) : first + (by_xcd ? int(blockIdx.x >> 3) : int(blockIdx.x));

    uint32_t   ids_cur[M], ids_nxt[M];
    const bool have_flags = a.elem_flags != nullptr;
    auto       loadIds    = [&](int bt, uint32_t (&ids)[M], uint32_t& flag) {
        const int lane_o = opaqueCopy(lane);
        uint32_t  fs     = 0;
        if (have_flags && bt < lim)
            fs = a.elem_flags[a.elem_begin + bt];
        flag = (fs & 1u) ? 3u : 0u;
        if (bt < lim && worker)
        {
            const uint32_t* en = a.elem_nodes + (a.elem_begin + int64_t(bt)) * NN + lane_o;
#pragma unroll
            for (int k = 0; k < M; ++k)
                ids[k] = en[k * MM];
        }
    };
    const double eta_l = qp[qa < NQ ? qa : 0], zeta_l = qp[qb < NQ ? qb : 0];
    const double wyz_l = qw[qa < NQ ? qa : 0] * qw[qb < NQ ? qb : 0];

    if (lane < MM)
        slotRows[lane] = reinterpret_cast< const uint4* >(a.slot_tab)[lane];
    stageFence();
    uint32_t flag_cur, flag_nxt = 0;
    loadIds(batch, ids_cur, flag_cur);

    while (batch < lim)
    {
        const uint32_t ticket = drawTicket();
        int            batch_next = 0;
        if (worker)
        {
            // ---- gather (gatherSumFact: Dirichlet dofs read as 0, MatrixFreeSystem.hpp:441-466) + S1: z interpolation
            double     u0[M][U];
            const bool flagged = (flag_cur & 1u) != 0;
#pragma unroll
            for (int k = 0; k < M; ++k)
            {
                const int64_t node = ids_cur[k];
                const double* p    = a.x + node * U;
#pragma unroll
                for (int hh = 0; hh < U / 2; ++hh)
                {
                    const double2 t = *reinterpret_cast< const double2* >(p + 2 * hh);
                    u0[k][2 * hh]     = t.x;
                    u0[k][2 * hh + 1] = t.y;
                }
                if (flagged)
                {
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        u0[k][u] = a.dirichlet[node * U + u] != 0 ? 0. : u0[k][u];
                }
            }
            // node ids to LDS in slot order right away (the scatter reads them from there): no id register lives through the stages
            {
                const uint4    srow  = slotRows[l];
                const uint32_t sw[4] = {srow.x, srow.y, srow.z, srow.w};
#pragma unroll
                for (int k = 0; k < M; ++k)
                    idsL[(k & 1) ? sw[k >> 1] >> 16 : sw[k >> 1] & 0xffffu] = ids_cur[k];
            }
            {
                double tI[2 * HQ * HQ];
                loadTable(tI, eoI + opaqueZero());
#pragma unroll
                for (int g = 0; g < NG; ++g)
                {
                    double in0[M], in1[M], o0[M], o1[M];
#pragma unroll
                    for (int k = 0; k < M; ++k)
                    {
                        in0[k] = u0[k][2 * g];
                        in1[k] = u0[k][2 * g + 1];
                    }
                    sweepEO< M, M, false, false >(in0, o0, tI);
                    sweepEO< M, M, false, false >(in1, o1, tI);
#pragma unroll
                    for (int q = 0; q < M; ++q)
                        stg(bufA, g, at(qa, qb, q), o0[q], o1[q]);
                }
            }
            stageFence();
            // ---- S2: y interpolation in place, lane (i, qz)
            {
                double tI[2 * HQ * HQ];
                loadTable(tI, eoI + opaqueZero());
                double in0[NG][M], in1[NG][M];
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int j = 0; j < M; ++j)
                    {
                        const double2 t = ldg(bufA, g, at(qa, j, qb));
                        in0[g][j]       = t.x;
                        in1[g][j]       = t.y;
                    }
#pragma unroll
                for (int g = 0; g < NG; ++g)
                {
                    double o0[M], o1[M];
                    sweepEO< M, M, false, false >(in0[g], o0, tI);
                    sweepEO< M, M, false, false >(in1[g], o1, tI);
#pragma unroll
                    for (int q = 0; q < M; ++q)
                        stg(bufA, g, at(qa, q, qb), o0[q], o1[q]);
                }
            }
            stageFence();
            // ---- S3: x interpolation in place, lane (qy, qz): the array now holds v at the quadrature points
            {
                double tI[2 * HQ * HQ];
                loadTable(tI, eoI + opaqueZero());
                double in0[NG][M], in1[NG][M];
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int i = 0; i < M; ++i)
                    {
                        const double2 t = ldg(bufA, g, at(i, qa, qb));
                        in0[g][i]       = t.x;
                        in1[g][i]       = t.y;
                    }
#pragma unroll
                for (int g = 0; g < NG; ++g)
                {
                    double o0[M], o1[M];
                    sweepEO< M, M, false, false >(in0[g], o0, tI);
                    sweepEO< M, M, false, false >(in1[g], o1, tI);
#pragma unroll
                    for (int q = 0; q < M; ++q)
                        stg(bufA, g, at(q, qa, qb), o0[q], o1[q]);
                }
            }
        }
        stageFence();
        // ---- geometry of this lane's x-pencil (qy, qz), vertices as wave-uniform operands (sumfact_fast.hpp)
        double G[6][3];
        double w[M][U];
        double vtx[24];
        {
            double vr = 0.;
            if (l < 24)
                vr = a.elem_verts[(a.elem_begin + int64_t(batch)) * 24 + opaqueCopy(l)];
#pragma unroll
            for (int t = 0; t < 24; ++t)
            {
                const int lo = __builtin_amdgcn_readlane(__double2loint(vr), t), hi = __builtin_amdgcn_readlane(__double2hiint(vr), t);
                vtx[t]       = __hiloint2double(hi, lo);
            }
        }
        hexPencilGeom(vtx, eta_l, zeta_l, G);
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int f = 0; f < U; ++f)
                w[m][f] = 0.;
        const double wyz = opaqueCopy(wyz_l) * a.alpha;
        // ---- the quadrature stage plane by plane in x
#pragma unroll
        for (int k = 0; k < M; ++k)
        {
            // (a) in-plane derivatives of plane k by the pencil lanes: d/deta (y-pencils), d/dzeta (z-pencils)
            if (pl_ok)
            {
                double tC[2 * HQ * HQ];
                loadTable(tC, eoC + opaqueZero());
                double in[M], out[M];
#pragma unroll
                for (int j = 0; j < M; ++j)
                    in[j] = Ad[pbase + j * pstride + 2 * k];
                sweepEO< M, M, true, false >(in, out, tC);
#pragma unroll
                for (int j = 0; j < M; ++j)
                    Pb[ppbase + j] = out[j];
            }
            stageFence();
            // (b) the plane's points: d/dxi by a dot product over the x-pencil, evalAtHexQPs, transposed x-sweep into w
            if (worker)
            {
                double ck[M];
                loadTable(ck, Ctp + k * M + opaqueZero());
                double vv[U], dv[3][U], r0[U], rd[3][U];
#pragma unroll
                for (int g = 0; g < NG; ++g)
                {
                    double2 t[M];
#pragma unroll
                    for (int m = 0; m < M; ++m)
                        t[m] = ldg(bufA, g, at(m, qa, qb));
                    vv[2 * g]     = t[k].x;
                    vv[2 * g + 1] = t[k].y;
                    double d0 = ck[0] * t[0].x, d1 = ck[0] * t[0].y;
#pragma unroll
                    for (int m = 1; m < M; ++m)
                    {
                        d0 += ck[m] * t[m].x;
                        d1 += ck[m] * t[m].y;
                    }
                    dv[0][2 * g]     = d0;
                    dv[0][2 * g + 1] = d1;
                }
#pragma unroll
                for (int f = 0; f < U; ++f)
                {
                    dv[1][f] = Pb[f * MM + l];
                    dv[2][f] = Pb[(U + f) * MM + lt];
                }
                qpStage< K, 1, false, 1, 0, false >(kern, G, qp[k], qw[k] * wyz, a.time, vv, dv, r0, rd);
#pragma unroll
                for (int f = 0; f < U; ++f)
                {
#pragma unroll
                    for (int m = 0; m < M; ++m)
                        w[m][f] += ck[m] * rd[0][f];
                    w[k][f] += r0[f];
                    Pb[f * MM + l]        = rd[1][f];
                    Pb[(U + f) * MM + lt] = rd[2][f];
                }
            }
            stageFence();
            // (c) transposed in-plane sweeps, in place in the plane buffer
            if (pl_ok)
            {
                double tCt[2 * HQ * HQ];
                loadTable(tCt, eoCt + opaqueZero());
                double in[M], out[M];
#pragma unroll
                for (int j = 0; j < M; ++j)
                    in[j] = Pb[ppbase + j];
                sweepEO< M, M, true, false >(in, out, tCt);
#pragma unroll
                for (int j = 0; j < M; ++j)
                    Pb[ppbase + j] = out[j];
            }
            stageFence();
            // (d) their sum joins the x-pencil
            if (worker)
            {
#pragma unroll
                for (int f = 0; f < U; ++f)
                    w[k][f] += Pb[f * MM + l] + Pb[(U + f) * MM + lt];
            }
            stageFence();
        }
        if (worker)
        {
            // ---- I^T along x from the register pencil w -> array (c = ix, b = qy, a = qz)
            {
                double tIt[2 * HQ * HQ];
                loadTable(tIt, eoIt + opaqueZero());
#pragma unroll
                for (int g = 0; g < NG; ++g)
                {
                    double w0[M], w1[M], o0[M], o1[M];
#pragma unroll
                    for (int q = 0; q < M; ++q)
                    {
                        w0[q] = w[q][2 * g];
                        w1[q] = w[q][2 * g + 1];
                    }
                    sweepEO< M, M, false, false >(w0, o0, tIt);
                    sweepEO< M, M, false, false >(w1, o1, tIt);
#pragma unroll
                    for (int i = 0; i < M; ++i)
                        stg(bufA, g, at(i, qa, qb), o0[i], o1[i]);
                }
            }
            stageFence();
            // ---- I^T along y in place, lane (ix, qz)
            {
                double tIt[2 * HQ * HQ];
                loadTable(tIt, eoIt + opaqueZero());
                double in0[NG][M], in1[NG][M];
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int q = 0; q < M; ++q)
                    {
                        const double2 t = ldg(bufA, g, at(qa, q, qb));
                        in0[g][q]       = t.x;
                        in1[g][q]       = t.y;
                    }
#pragma unroll
                for (int g = 0; g < NG; ++g)
                {
                    double o0[M], o1[M];
                    sweepEO< M, M, false, false >(in0[g], o0, tIt);
                    sweepEO< M, M, false, false >(in1[g], o1, tIt);
#pragma unroll
                    for (int j = 0; j < M; ++j)
                        stg(bufA, g, at(qa, j, qb), o0[j], o1[j]);
                }
            }
            stageFence();
            // ---- I^T along z, lane (ix, iy); the result is staged IN PLACE as [slot][unknown]: every lane has read its pencils of
            // all groups before the first store (one wave, LDS in order)
            {
                double tIt[2 * HQ * HQ];
                loadTable(tIt, eoIt + opaqueZero());
                const uint4    srow  = slotRows[l];
                const uint32_t sw[4] = {srow.x, srow.y, srow.z, srow.w};
                double         in0[NG][M], in1[NG][M];
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int q = 0; q < M; ++q)
                    {
                        const double2 t = ldg(bufA, g, at(qa, qb, q));
                        in0[g][q]       = t.x;
                        in1[g][q]       = t.y;
                    }
                stageFence();
#pragma unroll
                for (int g = 0; g < NG; ++g)
                {
                    double o0[M], o1[M];
                    sweepEO< M, M, false, false >(in0[g], o0, tIt);
                    sweepEO< M, M, false, false >(in1[g], o1, tIt);
#pragma unroll
                    for (int k = 0; k < M; ++k)
                    {
                        const uint32_t slot = (k & 1) ? sw[k >> 1] >> 16 : sw[k >> 1] & 0xffffu;
                        *reinterpret_cast< double2* >(Ad + slot * U + 2 * g) = make_double2(o0[k], o1[k]);
                    }
                }
            }
        }
        stageFence();
        batch_next = dyn ? ticketBatch(ticket) : batch + stride;
        loadIds(batch_next, ids_nxt, flag_nxt);
        // ---- scatter in slot order by all 64 lanes (as sumfact_fast.hpp: dense atomic requests on the shell slots, 16-byte
        // stores of alpha A x + beta y on the exclusive slots)
        {
            const int             lane_s  = opaqueCopy(lane);
            const int             sl      = lane_s;
            const bool            flagged = (flag_cur & 2u) != 0;
            static_assert(SG % U == 0);
            const int             sl_node = sl / U, sl_o = sl % U, sl_node2 = sl / (U / 2), sl_o2 = 2 * (sl % (U / 2));
            const uint32_t* const ids1 = idsL + sl_node;
            const double* const   sb1  = Ad + sl;
            const uint32_t* const ids2 = idsL + sl_node2;
            const double2* const  sb2  = reinterpret_cast< const double2* >(Ad) + sl;
            auto shellRound = [&]< int NSH_, bool FLAGGED >(int r) {
                if ((r + 1) * SG > NSH_ * U && r * SG + sl >= NSH_ * U)
                    return;
                const int64_t node = ids1[r * (SG / U)];
                const int64_t dof  = node * U + sl_o;
                const double  val  = sb1[r * SG];
                if constexpr (FLAGGED)
                    if (a.dirichlet[dof] != 0)
                        return;
                unsafeAtomicAdd(a.y + dof, val);
            };
            auto exclRound = [&]< int NSH_, bool FLAGGED >(int r) {
                if (NSH_ * (U / 2) + r * SG + sl >= NN * (U / 2))
                    return;
                const int64_t node = ids2[NSH_ + r * (SG / (U / 2))];
                const int64_t dof  = node * U + sl_o2;
                const double2 val  = sb2[NSH_ * (U / 2) + r * SG];
                double*       dst  = a.y + dof;
                double2       out  = val;
                if constexpr (FLAGGED)
                {
                    out.x = a.dirichlet[dof] != 0 ? 0. : val.x;
                    out.y = a.dirichlet[dof + 1] != 0 ? 0. : val.y;
                }
                if (a.beta != 0.)
                {
                    const double2 old = *reinterpret_cast< const double2* >(dst);
                    out.x += a.beta * old.x;
                    out.y += a.beta * old.y;
                }
                *reinterpret_cast< double2* >(dst) = out;
            };
            constexpr int RS = (Cfg::NSH * U + SG - 1) / SG, RX = ((NN - Cfg::NSH) * (U / 2) + SG - 1) / SG;
            if (a.fuse_beta && !flagged)
            {
                constexpr int  NSHU = Cfg::NSH * U, NXH = (NN - Cfg::NSH) * (U / 2);
                uint32_t       nid[RS], nid2[RX > 0 ? RX : 1];
                double         val[RS];
                double2        val2[RX > 0 ? RX : 1];
                constexpr bool part1 = RS * SG > NSHU, part2 = RX * SG > NXH;
                const bool     in1 = !part1 || (RS - 1) * SG + sl < NSHU, in2 = !part2 || (RX - 1) * SG + sl < NXH;
#pragma unroll
                for (int r = 0; r < RS; ++r)
                {
                    nid[r] = ids1[r * (SG / U)];
                    val[r] = sb1[r * SG];
                }
#pragma unroll
                for (int r = 0; r < RX; ++r)
                    if (r + 1 < RX || in2)
                    {
                        nid2[r] = ids2[Cfg::NSH + r * (SG / (U / 2))];
                        val2[r] = sb2[Cfg::NSH * (U / 2) + r * SG];
                    }
#pragma unroll
                for (int r = 0; r < RS; ++r)
                    if (r + 1 < RS || in1)
                        unsafeAtomicAdd(a.y + int64_t(nid[r]) * U + sl_o, val[r]);
#pragma unroll
                for (int r = 0; r < RX; ++r)
                    if (r + 1 < RX || in2)
                    {
                        double* dst = a.y + int64_t(nid2[r]) * U + sl_o2;
                        double2 out = val2[r];
                        if (a.beta != 0.)
                        {
                            const double2 old = *reinterpret_cast< const double2* >(dst);
                            out.x += a.beta * old.x;
                            out.y += a.beta * old.y;
                        }
                        *reinterpret_cast< double2* >(dst) = out;
                    }
            }
            else if (a.fuse_beta)
            {
#pragma unroll 1
                for (int r = 0; r < RS; ++r)
                    shellRound.template operator()< Cfg::NSH, true >(r);
#pragma unroll 1
                for (int r = 0; r < RX; ++r)
                    exclRound.template operator()< Cfg::NSH, true >(r);
            }
            else if (!flagged)
            {
#pragma unroll 1
                for (int r = 0; r < (NN * U + SG - 1) / SG; ++r)
                    shellRound.template operator()< NN, false >(r);
            }
            else
            {
#pragma unroll 1
                for (int r = 0; r < (NN * U + SG - 1) / SG; ++r)
                    shellRound.template operator()< NN, true >(r);
            }
        }
        stageFence(); // the array is rewritten by the next element
#pragma unroll
        for (int k = 0; k < M; ++k)
            ids_cur[k] = ids_nxt[k];
        flag_cur = flag_nxt;
        batch    = batch_next;
    }
}

// Launch: the same contract as launchSumfactFast for the plain single-column apply with ghost rows behind the owned rows
template < typename K, int P, int NQ >
int launchSumfactPlane(const ElemArgs& a, const void* kparam_blob, hipStream_t stream)
{
    using Cfg = PlaneCfg< K, P, NQ >;
    static_assert(Cfg::feasible);
    K kern{};
    if (kparam_blob)
        __builtin_memcpy(&kern, kparam_blob, sizeof(K));
    auto              kernel = sumfactPlaneKernel< K, P, NQ >;
    static bool       ready[64] = {};
    static int        n_cus_of[64] = {};
    static std::mutex mtx;
    int               dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64)
    {
        setError("device index %d not supported", dev);
        return -3;
    }
    {
        std::lock_guard< std::mutex > lock{mtx};
        if (!ready[dev])
        {
            hipDeviceProp_t prop;
            if (hipFuncSetAttribute(reinterpret_cast< const void* >(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(Cfg::lds)) != hipSuccess ||
                hipGetDeviceProperties(&prop, dev) != hipSuccess)
            {
                setError("sumfactPlaneKernel: launch set-up failed");
                return -3;
            }
            n_cus_of[dev] = prop.multiProcessorCount;
            ready[dev]    = true;
        }
    }
    int waves_cu = Cfg::waves_by_lds < 4 * L3K_PLANE_MIN_WAVES ? Cfg::waves_by_lds : 4 * L3K_PLANE_MIN_WAVES;
    if (const char* e = std::getenv("L3K_FAST_WAVES_PER_CU"))
        waves_cu = std::atoi(e) > 0 ? std::atoi(e) : waves_cu;
    const int64_t  n_batches  = a.elem_count;
    const int64_t  max_blocks = int64_t(n_cus_of[dev]) * waves_cu;
    const unsigned grid       = static_cast< unsigned >(n_batches < max_blocks ? n_batches : max_blocks);
    constexpr TableLayout    TL{P + 1, NQ};
    PlaneTables< P + 1, NQ > tab;
    const double*            th = a.tables_host;
    __builtin_memcpy(tab.eoI, th + TL.offEoI(), sizeof tab.eoI);
    __builtin_memcpy(tab.eoC, th + TL.offEoC(), sizeof tab.eoC);
    __builtin_memcpy(tab.eoIt, th + TL.offEoIt(), sizeof tab.eoIt);
    __builtin_memcpy(tab.eoCt, th + TL.offEoCt(), sizeof tab.eoCt);
    __builtin_memcpy(tab.qw, th + TL.offW(), sizeof tab.qw);
    __builtin_memcpy(tab.qx, th + TL.offX(), sizeof tab.qx);
    for (int k = 0; k < NQ; ++k)
        for (int m = 0; m < NQ; ++m)
            tab.Ct[k * NQ + m] = th[TL.offC() + m * NQ + k];
    const int xcd_chunk = (grid % 8 == 0 && n_batches >= int64_t(grid) && n_batches < (int64_t(1) << 30) && std::getenv("L3K_FAST_NO_XCD") == nullptr)
                              ? int((n_batches + 7) / 8)
                              : 0;
    if (a.work_counters && hipMemsetAsync(a.work_counters, 0, 8 * 128, stream) != hipSuccess)
    {
        setError("hipMemsetAsync(work counters) failed");
        return -3;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), Cfg::lds, stream, a, kern, n_batches, xcd_chunk, tab);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess)
    {
        setError("sumfactPlaneKernel launch failed: %s", hipGetErrorString(err));
        return -3;
    }
    return 0;
}
} // namespace l3k::dev
#endif
