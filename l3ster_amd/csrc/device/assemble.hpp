// assemble.hpp -- LocalAssembly on the device: K_e = sum_q w detJ B_q^T B_q for a batch of elements
// (assembleLocalSystem + LocalSystemManager, algsys/AssembleLocalSystem.hpp:77-216,234-256).
//
// The reference accumulates batched symmetric rank-k updates (Eigen selfadjointView::rankUpdate, :192-208).  Here the
// same sum is one GEMM per element, K_e = (W Z)^T Z with Z[(q,e), (b,u)] = B_q[e, b*U+u] of size (nq^3 E) x (n^3 U)
// (2401 x 1372 at p = 6, 9.0 GFLOP, 4.5 using symmetry), which is genuinely matmul-shaped: it runs on the FP64 matrix
// cores (v_mfma_f64_16x16x4_f64).  Two kernels:
//   1. assembleCoeffKernel: one workgroup per element evaluates, per quadrature point, the user kernel and the geometry
//      and stores c_k[q][e][u] (k = 0: A0, k = 1..3: sum_s A_s Ji[k-1][s]) and w detJ -- 113 doubles per point;
//   2. assembleGemmKernel: one workgroup per 128x128 tile of the lower triangle of K_e and per element; Z tiles are
//      generated on the fly in LDS from c_k and the 1-D tables (Z = c0 phi + c1 dphi/dxi + c2 dphi/deta + c3 dphi/dzeta,
//      phi and its reference derivatives are products of I / D entries), 4 waves x (4x4) MFMA blocks; the tile and its
//      mirror image are written (getSystem symmetrises from the lower triangle, :176-182), or only a checksum.
// F_e = sum_q w detJ B_q^T f_q comes from the sum-factorised RHS-mode kernel with element-local output.
#ifndef L3K_DEVICE_ASSEMBLE_HPP
#define L3K_DEVICE_ASSEMBLE_HPP

#include "sumfact_apply.hpp"

#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

namespace l3k::dev
{
using mfma_d4 = __attribute__((ext_vector_type(4))) double;

template < typename K >
constexpr int coeffStride()
{
    return 4 * K::params.n_equations * K::params.n_unknowns + 1; // c_k[e][u][k] then w*detJ
}

template < typename K, int P, int NQ >
constexpr size_t coeffLdsBytes()
{
    constexpr int M = cmax(P + 1, NQ);
    return sizeof(double) * (size_t(5 * K::params.n_fields) * M * M * M + 24);
}
// GS: the field buffers in the workgroup's slice of a.scratch instead of the LDS, persistent workgroups (sumfact_apply.hpp)
template < typename K, int P, int NQ, bool GS = false >
__global__ __launch_bounds__((applyThreads< P, NQ >())) void assembleCoeffKernel(const ElemArgs a, const K kern, double* __restrict__ cbuf)
{
    constexpr KernelParams params = K::params;
    constexpr int          U = params.n_unknowns, E = params.n_equations, F = params.n_fields;
    constexpr int          N1 = P + 1, NN = N1 * N1 * N1, NQP = NQ * NQ * NQ, M = cmax(N1, NQ), M3 = M * M * M;
    constexpr int          NT = applyThreads< P, NQ >(), CS = coeffStride< K >();
    constexpr TableLayout  TL{N1, NQ};
    using Iface = KernelInterface< KernelParams{params.dimension, E, U, F, 1} >;

    extern __shared__ double lds[];
    double* const            Fv = GS ? a.scratch + size_t(blockIdx.x) * (coeffLdsBytes< K, P, NQ >() / sizeof(double)) : lds; // fields at the QPs: values, 3 reference derivatives, 1 temp
    double* const            vs = Fv + 5 * F * M3;  // [8][3]
    const int                tid = threadIdx.x;
    int64_t                  el  = blockIdx.x;      // element within the batch
    do
    {
    const int64_t            e   = a.elem_begin + el;
    const uint32_t*          en  = a.elem_nodes + e * NN;
    if (tid < 24)
        vs[tid] = a.elem_verts[e * 24 + tid];
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
    for (int q = tid; q < NQP; q += NT)
    {
        const int     qx = q % NQ, qy = (q / NQ) % NQ, qz = q / (NQ * NQ);
        const double* qw = a.tables + TL.offW();
        const double* qp = a.tables + TL.offX();
        double        G[6][3], Jm[3][3], Ji[3][3], xyz[3];
        hexPencilGeom(vs, qp[qy], qp[qz], G);
        hexPointOnPencil(G, qp[qx], Jm, xyz);
        const double det = inverse3(Jm, Ji);
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
        // records of one element as [entry][q] (structure of arrays): neighbouring threads = neighbouring quadrature points
        // write, and the assembly kernels read, neighbouring addresses
        // K_tiled == 2 (the x-major tiled layout, api_assembled.hip): the records of the element with the roles of x and z exchanged --
        // point (qx, qy, qz) at position (qz, qy, qx), the xi and the zeta derivative exchanged; the 1-D tables are the same in the
        // three directions, so the assembly kernel then forms the same K_e with x and z exchanged in its loops and in its layout
        const bool xz = a.K_tiled == 2;
        double*    c  = cbuf + el * NQP * CS + (xz ? qz + NQ * (qy + NQ * qx) : q);
#pragma unroll
        for (int e_ = 0; e_ < E; ++e_)
#pragma unroll
            for (int u = 0; u < U; ++u)
            {
                c[((e_ * U + u) * 4 + 0) * NQP] = res.operators[0](e_, u);
#pragma unroll
                for (int d = 0; d < 3; ++d)
                    c[((e_ * U + u) * 4 + 1 + (xz ? 2 - d : d)) * NQP] = res.operators[1](e_, u) * Ji[d][0] + res.operators[2](e_, u) * Ji[d][1] +
                                                                         res.operators[3](e_, u) * Ji[d][2];
            }
        c[(CS - 1) * NQP] = qw[qx] * qw[qy] * qw[qz] * det;
        if (!(det > 0.)) // reference: "Encountered degenerate element ( |J| <= 0 )" (AssembleLocalSystem.hpp:249)
            a.workspace[int64_t(a.elem_count) * NQP * CS] = 1.;
    }
    if constexpr (GS)
        __syncthreads();
    } while (GS && (el += gridDim.x) < a.elem_count);
}

template < int P, int NQ, int U, int E >
struct GemmCfg
{
    static constexpr int N1 = P + 1, NN = N1 * N1 * N1, ND = NN * U, NQP = NQ * NQ * NQ, KD = NQP * E;
    static constexpr int BT  = 128;                        // tile edge
    static constexpr int NTL = (ND + BT - 1) / BT;         // tiles per edge
    static constexpr int NLT = NTL * (NTL + 1) / 2;        // lower-triangular tiles
    static constexpr int QC  = 4;                          // quadrature points per K chunk
    static constexpr int KC  = QC * E;                     // K chunk: a multiple of the MFMA k-step 4 for every E
    static constexpr int LDS_ROW = BT + 16;                // +16 doubles: rows k and k+1 land in disjoint bank halves
    static constexpr int CSP = ((4 * E * U + 1) + 1) & ~1; // coefficient record padded to an even number of doubles
    static constexpr size_t lds = sizeof(double) * (2 * KC * LDS_ROW + QC * CSP + 2 * N1 * NQ);
};

#ifndef L3K_GEMM_MIN_BLOCKS
#define L3K_GEMM_MIN_BLOCKS 2
#endif
// two workgroups per CU (LDS: 69 KB each): one generates its Z chunk while the other runs its MFMAs; needs <= 256 registers
template < typename K, int P, int NQ >
__global__ __launch_bounds__(256, L3K_GEMM_MIN_BLOCKS) void assembleGemmKernel(const ElemArgs a, const double* __restrict__ cbuf, int64_t elem0)
{
    constexpr KernelParams params = K::params;
    constexpr int          U = params.n_unknowns, E = params.n_equations;
    using C = GemmCfg< P, NQ, U, E >;
    constexpr int         N1 = C::N1, ND = C::ND, NQP = C::NQP, KC = C::KC, QC = C::QC, BT = C::BT, LR = C::LDS_ROW, CSP = C::CSP;
    constexpr int         CS = coeffStride< K >();
    constexpr TableLayout TL{N1, NQ};

    extern __shared__ double lds[];
    double* const            ZA = lds;                 // [KC][LR]: w*detJ * Z for the tile's rows (A operand)
    double* const            ZB = ZA + KC * LR;        // [KC][LR]: Z for the tile's columns (B operand)
    double* const            cs = ZB + KC * LR;        // [QC][CSP]: coefficient records of the chunk's quadrature points
    double* const            tI = cs + QC * CSP;       // [N1][NQ]
    double* const            tD = tI + N1 * NQ;

    const int     tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int64_t el  = blockIdx.y; // element within the batch
    // lower-triangular tile index -> (ti, tj), tj <= ti
    int ti = 0, rem = blockIdx.x;
    while (rem > ti)
    {
        rem -= ti + 1;
        ++ti;
    }
    const int tj = rem;

    for (int i = tid; i < 2 * N1 * NQ; i += 256)
        tI[i] = a.tables[(i < N1 * NQ ? TL.offI() : TL.offD() - N1 * NQ) + i];

    // this thread generates the Z entries of one fixed column of each panel for QC/2 quadrature points of every chunk
    const int  col  = tid & (BT - 1), half = tid >> 7;
    const int  ga = ti * BT + col, gb = tj * BT + col;
    const bool va = ga < ND, vb = gb < ND;
    const int  ba = va ? ga / U : 0, ua = va ? ga % U : 0, bb = vb ? gb / U : 0, ub = vb ? gb % U : 0;
    const int  bax = ba % N1, bay = (ba / N1) % N1, baz = ba / (N1 * N1);
    const int  bbx = bb % N1, bby = (bb / N1) % N1, bbz = bb / (N1 * N1);

    mfma_d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[i][j] = mfma_d4{0., 0., 0., 0.};

    const double* cel = cbuf + el * NQP * CS;
    for (int q0 = 0; q0 < NQP; q0 += QC)
    {
        // ---- stage the chunk's coefficient records (contiguous in the workspace) in LDS
        for (int i = tid; i < QC * CS; i += 256)
        {
            const int r = i / QC, qi = i - r * QC; // (the workspace holds [entry][q]: consecutive threads read consecutive q)
            cs[qi * CSP + r] = q0 + qi < NQP ? cel[int64_t(r) * NQP + q0 + qi] : 0.;
        }
        __syncthreads();
        // ---- generate the two Z chunks: the basis products of (column, q) once, then 4 FMAs per equation
#pragma unroll
        for (int s = 0; s < QC / 2; ++s)
        {
            const int qi = half * (QC / 2) + s, q = q0 + qi;
            if (q < NQP)
            {
                const int     qx = q % NQ, qy = (q / NQ) % NQ, qz = q / (NQ * NQ);
                const double* cq = cs + qi * CSP;
                const double  w  = cq[CS - 1];
                double        pa[4], pb[4];
                {
                    const double ix = tI[bax * NQ + qx], iy = tI[bay * NQ + qy], iz = tI[baz * NQ + qz];
                    const double dx = tD[bax * NQ + qx], dy = tD[bay * NQ + qy], dz = tD[baz * NQ + qz];
                    const double yz = iy * iz * w, xw = ix * w;
                    pa[0] = ix * yz, pa[1] = dx * yz, pa[2] = xw * dy * iz, pa[3] = xw * iy * dz;
                }
                {
                    const double ix = tI[bbx * NQ + qx], iy = tI[bby * NQ + qy], iz = tI[bbz * NQ + qz];
                    const double dx = tD[bbx * NQ + qx], dy = tD[bby * NQ + qy], dz = tD[bbz * NQ + qz];
                    const double yz = iy * iz;
                    pb[0] = ix * yz, pb[1] = dx * yz, pb[2] = ix * dy * iz, pb[3] = ix * iy * dz;
                }
#pragma unroll
                for (int e_ = 0; e_ < E; ++e_)
                {
                    const double* ca = cq + (e_ * U + ua) * 4;
                    const double* cb = cq + (e_ * U + ub) * 4;
                    const double  za = ca[0] * pa[0] + ca[1] * pa[1] + ca[2] * pa[2] + ca[3] * pa[3];
                    const double  zb = cb[0] * pb[0] + cb[1] * pb[1] + cb[2] * pb[2] + cb[3] * pb[3];
                    ZA[(qi * E + e_) * LR + col] = va ? za : 0.;
                    ZB[(qi * E + e_) * LR + col] = vb ? zb : 0.;
                }
            }
            else
            {
#pragma unroll
                for (int e_ = 0; e_ < E; ++e_)
                {
                    ZA[(qi * E + e_) * LR + col] = 0.;
                    ZB[(qi * E + e_) * LR + col] = 0.;
                }
            }
        }
        __syncthreads();
        // ---- KC/4 k-steps of 16 MFMAs: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15]
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks)
        {
            const int krow = ks * 4 + (lane >> 4);
            double    af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
            {
                af[i] = ZA[krow * LR + wm * 64 + i * 16 + (lane & 15)];
                bf[i] = ZB[krow * LR + wn * 64 + i * 16 + (lane & 15)];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        // the next chunk's staging writes `cs` only; Z is rewritten after the barrier that follows the staging
    }

    // ---- epilogue: C/D layout of the f64 MFMA: row = (lane>>4) + 4*reg, col = lane&15
    double* Kel = a.K ? a.K + (elem0 + el) * int64_t(ND) * ND : nullptr;
    double  csum = 0.;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                const int    gi = ti * BT + wm * 64 + i * 16 + (lane >> 4) + 4 * r;
                const int    gj = tj * BT + wn * 64 + j * 16 + (lane & 15);
                const double v  = acc[i][j][r];
                // lower triangle only (diagonal tiles are computed in full, their upper half is dropped), then mirrored:
                // the result is bitwise symmetric like the reference's selfadjointView copy (:176-182)
                if (gi < ND && gj <= gi)
                {
                    if (Kel)
                    {
                        Kel[int64_t(gi) * ND + gj] = v;
                        if (gi != gj)
                            Kel[int64_t(gj) * ND + gi] = v;
                    }
                    csum += v * (1 + ((gi * 31 + gj * 17) % 7));
                    if (gi != gj)
                        csum += v * (1 + ((gj * 31 + gi * 17) % 7));
                }
            }
    if (a.checksum)
    {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            csum += __shfl_down(csum, off);
        if (lane == 0)
            unsafeAtomicAdd(a.checksum + elem0 + el, csum);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Sum-factorised LocalAssembly.  The dense product above executes 2 * (nq^3 E) * (n^3 U)^2 / 2 flops per element
// (4.5 GFLOP at order 6) on a part whose FP64 matrix rate equals its FP64 vector rate (and shares the pipe with it:
// profiles/r02_fp64_vector_matrix_coexecution.log), so the matrix cores buy nothing and the flop count is everything.
// With the tensor-product basis
//     K[(b,u),(b',u')] = sum_{k,k'} sum_q G_kk'^{uu'}(q) psi_k(b,q) psi_k'(b',q),   G_kk'^{uu'}(q) = w detJ sum_e c_k[e][u] c_k'[e][u'],
// psi_0 = Ix Iy Iz, psi_1 = Dx Iy Iz, psi_2 = Ix Dy Iz, psi_3 = Ix Iy Dz, the sum over q = (qx, qy, qz) factorises
// direction by direction on index PAIRS (b1, b1') with the product tables P[t][(b1,b1')][q1] = T_s[b1][q1] T_s'[b1'][q1]
// (t = s + 2 s', T_0 = I, T_1 = D):
//     stage 1 (qx):  A[ty,tz][qy,qz]        = sum_{(k,k') of type (ty,tz)} sum_qx P[tx][(bx,bx')][qx] G_kk'[qx,qy,qz]     (9 groups)
//     stage 2 (qy):  B[tz][(by,by')][qz]    = sum_{ty} sum_qy P[ty][(by,by')][qy] A[ty,tz][qy,qz]                          (4 groups)
//     stage 3 (qz):  M[(by,by')][(bz,bz')]  = sum_{tz} sum_qz B[tz][(by,by')][qz] P[tz][(bz,bz')][qz]
// for every (bx, bx') and every pair of unknowns: ~1e8 flop per element at order 6 instead of 4.5e9 (O(n^7) instead of
// O(n^9) per pair of unknowns), the same K_e to rounding (another summation order).  One workgroup per (element, u' <= u);
// only the blocks u' <= u are formed (diagonal blocks: one half) and mirrored, so K_e is bitwise symmetric as the
// reference's (algsys/AssembleLocalSystem.hpp:176-182).
// Variants of the one kernel template (DESIGN.md 4.4 has the measurements behind each):
//   * BLOCKS = 0: every pair u' <= u in one launch, seven iterations b_x', diagonal blocks: the z-major half -- the stored
//     row-major matrices (the four workgroups that fill a 64-byte line of K_e pass through the same b_x' together);
//   * BLOCKS = 1 / 2: the diagonal / the off-diagonal blocks as kernels of their own (a register allocation each) -- the
//     streaming mode; diagonal blocks by x-major halves in n / 2 + 1 merged iterations;
//   * TILED: all U x U blocks into the tiled layout (coalesced stores);
//   * orders >= 4 of the streaming and tiled kernels: wave-uniform operands by DPP row broadcast (the slot's A entries in
//     stage 2, the 1-D tables in stage 3) instead of LDS reads and scalar loads; the rows of a slot then sit in whole
//     16-lane units and stages 2 and 3 run on whole waves.
template < int P, int NQ >
struct SfAsmCfg
{
    static constexpr int N1 = P + 1, N2 = N1 * N1, NQP = NQ * NQ * NQ;
    static constexpr int PAIRS = N1;                      // one iteration = all bx for one bx'
    static constexpr int ROWS  = PAIRS * N2;              // one thread per row (bx, by, by')
    static constexpr int K3    = 4 * NQ;                  // stage-3 contraction length: (tz, qz)
    static constexpr int AROW  = NQ;                      // (no padding: at order 6 the workgroup then needs 79.6 KB -> two per CU)
    static constexpr int threads = ((ROWS + 63) / 64) * 64;
    // G | P | A
    static constexpr size_t lds = sizeof(double) * (size_t(16) * NQP + size_t(4) * N2 * NQ + size_t(PAIRS) * 9 * NQ * AROW);
    static constexpr bool feasible = lds <= 160 * 1024 && threads <= 1024;
    // DPP2 (the streaming and the tiled-store kernels at orders >= 4): the rows of a slot sit in whole 16-lane DPP rows (a slot's
    // A entries are operands of stage 2 by row broadcast): UNITS * 16 lanes per slot instead of N2 (order 6: 64 for 49)
    static constexpr bool dpp2(bool tiled, int blocks) { return (tiled || blocks != 0) && P >= 4; }
    static constexpr int  UNITS = (N2 + 15) / 16; // 16-lane units per slot
    static constexpr int  rowThreadsFor(bool tiled, int blocks) { return dpp2(tiled, blocks) ? ((PAIRS * UNITS * 16 + 63) / 64) * 64 : threads; }
    // PRODUCER (DPP kernels at orders 5 and 6; order 4: slower, order 7: a ninth wave would halve the registers): one more wave forms A of the NEXT iteration (lane = (qy, qz), every entry of G read
    // once for all slots, the 1-D tables as scalar operands) while the row waves run stages 2 and 3 of this one on the other copy
    // of A -- stage 1 is the LDS-bound phase (G is read once per slot by the cooperative form), stages 2 and 3 the FP64-bound one,
    // and at order 6 the eighth wave sits on the SIMD that had one: 2, 2, 2, 2
    static constexpr size_t A_BYTES = sizeof(double) * size_t(PAIRS) * 9 * NQ * AROW;
    static constexpr bool   producer(bool tiled, int blocks)
    {
#ifdef L3K_ASM_NO_PRODUCER
        return false;
#else
        return dpp2(tiled, blocks) && P >= 5 && NQ * NQ <= 64 && rowThreadsFor(tiled, blocks) + 64 <= 512 && lds + A_BYTES <= 160 * 1024;
#endif
    }
    static constexpr int    threadsFor(bool tiled, int blocks) { return rowThreadsFor(tiled, blocks) + (producer(tiled, blocks) ? 64 : 0); }
    static constexpr size_t ldsFor(bool tiled, int blocks) { return lds + (producer(tiled, blocks) ? A_BYTES : 0); }
};

// (The scalar-operand form, kept for orders < 4 and the one-launch kernel:)
// Stage 1 is cooperative (A[bx][group][qz][qy] for the iteration's bx', through LDS); stages 2 and 3 are fused per row:
// the thread of row (bx, by, by') forms its 4 nq values B[tz][qz] in registers from its own rows of the y product tables (28
// doubles re-read from LDS per iteration) and the A arrays (LDS reads shared by the 49 threads of a bx), then forms the n^2
// entries (bz, bz') of its row in two steps (the z product table is a product of the two 1-D tables: stage 3 below), with
// the 1-D tables read by SCALAR loads (wave-uniform SGPR-pair operands: no LDS operand traffic in that stage).
// out[q] (+)= sum_b in[b] W[b][q] from the even-odd tables We | Wo of W (host/tables.cpp:evenOddTables; the same scheme as
// sweepEO of device/sumfact_fast.hpp), the tables read through the constant address space: scalar loads, SGPR operands
template < int NIN, int NOUT, bool ANTI, bool ACC >
__device__ __forceinline__ void sweepEOScalar(const double (&in)[NIN], double (&out)[NOUT],
                                              const __attribute__((address_space(4))) double* eo)
{
    constexpr int HI = NIN / 2, HO = NOUT / 2, RI = (NIN + 1) / 2, RO = (NOUT + 1) / 2;
    const __attribute__((address_space(4))) double* const We = eo;
    const __attribute__((address_space(4))) double* const Wo = eo + RI * RO;
    double                                                e[RI], o[HI > 0 ? HI : 1];
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
        double A = 0.;
#pragma unroll
        for (int r = 0; r < RI; ++r)
            A += e[r] * We[r * RO + q];
        double lo = A;
#pragma unroll
        for (int r = 0; r < HI; ++r)
            lo += o[r] * Wo[r * RO + q];
        const double hi = ANTI ? lo - 2. * A : 2. * A - lo;
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
// ---- wave-uniform table operands without scalar loads: a table lives in the lanes of every 16-lane DPP row (lane j of a row holds
// entries 16 c + j of chunk c), and an FP64 FMA takes entry T as "row_newbcast:(T % 16)" of chunk T / 16 -- v_fmac_f64_dpp at the
// rate of the plain instruction (tools/dpp_f64_probe.hip, profiles/r03_dpp_f64_probe.log).  Conditions, checked there: the source
// lane must be active (EXEC), so the code around these runs with whole rows of lanes; a VALU write of the table register needs two
// wait states before the DPP read (the tables are loaded once, by memory instructions; tools/check_dpp_hazards.py scans the ISA).
template < int T, int NCH >
__device__ __forceinline__ void fmaTab(double& acc, const double (&tab)[NCH], double v)
{
    static_assert(T >= 0 && T < 16 * NCH);
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(tab[T / 16]), "v"(v), "n"(T % 16));
}
template < int... I, typename F >
__device__ __forceinline__ void staticForImpl(std::integer_sequence< int, I... >, F&& f)
{
    (f(std::integral_constant< int, I >{}), ...);
}
template < int N, typename F >
__device__ __forceinline__ void staticFor(F&& f)
{
    staticForImpl(std::make_integer_sequence< int, N >{}, static_cast< F&& >(f));
}
// sweepEOScalar with the even-odd tables We | Wo in DPP rows.  The first term of every sum is a plain product with a scalar operand
// (there is no v_mul_f64 with DPP, and a zeroed accumulator costs a move per sum): first[q] = We[0][q], q = 0 .. RO-1, and
// first[RO] = Wo[0][HO] -- RO + 1 values per table, loaded once per kernel
template < int NIN, int NOUT, bool ANTI, bool ACC, int NCH >
__device__ __forceinline__ void sweepEODpp(const double (&in)[NIN], double (&out)[NOUT], const double (&tab)[NCH],
                                           const double (&first)[(NOUT + 1) / 2 + 1])
{
    constexpr int HI = NIN / 2, HO = NOUT / 2, RI = (NIN + 1) / 2, RO = (NOUT + 1) / 2, WO = RI * RO;
    double        e[RI], o[HI > 0 ? HI : 1];
#pragma unroll
    for (int r = 0; r < HI; ++r)
    {
        e[r] = in[r] + in[NIN - 1 - r];
        o[r] = in[r] - in[NIN - 1 - r];
    }
    if constexpr (NIN % 2)
        e[HI] = in[HI];
    // (the FMAs are volatile asm and stay in program order: term by term across the sums, not sum by sum -- consecutive
    // instructions are independent)
    double A[HO > 0 ? HO : 1], lo[HO > 0 ? HO : 1];
#pragma unroll
    for (int q = 0; q < HO; ++q)
        A[q] = e[0] * first[q];
    staticFor< RI - 1 >([&](auto rc) {
        staticFor< HO >([&](auto qc) { fmaTab< (decltype(rc)::value + 1) * RO + decltype(qc)::value >(A[decltype(qc)::value], tab, e[decltype(rc)::value + 1]); });
    });
#pragma unroll
    for (int q = 0; q < HO; ++q)
        lo[q] = A[q];
    staticFor< HI >([&](auto rc) {
        staticFor< HO >([&](auto qc) { fmaTab< WO + decltype(rc)::value * RO + decltype(qc)::value >(lo[decltype(qc)::value], tab, o[decltype(rc)::value]); });
    });
#pragma unroll
    for (int q = 0; q < HO; ++q)
    {
        const double hi   = ANTI ? lo[q] - 2. * A[q] : 2. * A[q] - lo[q];
        out[q]            = ACC ? out[q] + lo[q] : lo[q];
        out[NOUT - 1 - q] = ACC ? out[NOUT - 1 - q] + hi : hi;
    }
    if constexpr (NOUT % 2)
    {
        double m;
        if constexpr (ANTI)
        {
            m = o[0] * first[RO];
            staticFor< HI - 1 >([&](auto rc) { fmaTab< WO + (decltype(rc)::value + 1) * RO + HO >(m, tab, o[decltype(rc)::value + 1]); });
        }
        else
        {
            m = e[0] * first[HO];
            staticFor< RI - 1 >([&](auto rc) { fmaTab< (decltype(rc)::value + 1) * RO + HO >(m, tab, e[decltype(rc)::value + 1]); });
        }
        out[HO] = ACC ? out[HO] + m : m;
    }
}
__device__ __forceinline__ int opaqueOffset(int x)
{
    int y;
    asm volatile("v_mov_b32 %0, %1" : "=v"(y) : "v"(x));
    return y;
}
// the 9 nq dot products of stage 2 in the order of their results B[tz][qz]: (tz, qz, ty) with the group g of (ty, tz)
struct Stage2Item
{
    int tz, qz, ty, g;
};
template < int NQ >
struct Stage2Items
{
    Stage2Item v[9 * NQ];
    constexpr Stage2Items() : v{}
    {
        constexpr int gtab[4][4] = {{0, 4, 5, 6}, {1, -1, 7, -1}, {2, 8, -1, -1}, {3, -1, -1, -1}}; // [ty][tz]
        int           n          = 0;
        for (int tz = 0; tz < 4; ++tz)
            for (int qz = 0; qz < NQ; ++qz)
                for (int ty = 0; ty < 4; ++ty)
                    if (gtab[ty][tz] >= 0)
                        v[n++] = Stage2Item{tz, qz, ty, gtab[ty][tz]};
    }
};
// TILED: all U x U blocks (no symmetry: more FP64 instructions for a layout in which the stores of a wave fill contiguous memory and
// a row reader finds runs of n^2 doubles), stored as [u][u'][bx'][bz][bx][by][by'][bz'] -- row node b = (bx, by, bz), column node
// b' = (bx', by', bz').  In the row-major layout of the reference every 64-byte line of K_e collects its 8 entries from four
// workgroups and two iterations: measured write traffic 3.9 x the matrix (profiles/r03_tcc_assembly_stored.txt).
// BLOCKS: 0 every pair u' <= u of unknowns (TILED: every pair), 1 the diagonal blocks u' == u only, 2 the off-diagonal ones only --
// as kernels of their own the two kinds of iteration below get a register allocation each (both in one kernel: 302 k matrices/s at
// order 6; separately 1.04 + 2.22 us per element of which 0.5 counted twice)
// TMODE: 0 row-major / streaming, 1 the tiled layout (TILED), 2 the x-major tiled layout, lower triangle only (TILED with the
// diagonal-block kernel's merged iterations for ALL U x U blocks: the pairs bx' <= bx in n / 2 + 1 iterations of n slots, the
// slots with bx' == bx with their rows by' <= by -- the half "column line (bx', by') not behind the row line (bx, by)" that the
// mirroring transposition of api_assembled.hip reads; bx here is the kernel's first index, the element's z after the exchange)
template < typename K, int P, int NQ, int TMODE = 0, int BLOCKS = 0 >
__global__ __launch_bounds__((SfAsmCfg< P, NQ >::threadsFor(TMODE != 0, BLOCKS))) void assembleSumfactKernel(const ElemArgs a, const double* __restrict__ cbuf,
                                                                                       int64_t elem0, int xcd_group)
{
    constexpr bool TILED = TMODE != 0;
    constexpr KernelParams params = K::params;
    constexpr int          U = params.n_unknowns, E = params.n_equations;
    using C = SfAsmCfg< P, NQ >;
    constexpr int         N1 = C::N1, N2 = C::N2, NQP = C::NQP, ROWS = C::ROWS, AROW = C::AROW;
    constexpr int         NT = C::threadsFor(TILED, BLOCKS), ND = N1 * N2 * U;
    constexpr int         CS = coeffStride< K >();
    constexpr TableLayout TL{N1, NQ};
    // the 16 terms (k, k') by group: type of a direction = s + 2 s' with s = (k == d + 1), s' = (k' == d + 1); groups (ty, tz):
    // 0 (II,II)  1 (DI,II)  2 (ID,II)  3 (DD,II)  4 (II,DI)  5 (II,ID)  6 (II,DD)  7 (DI,ID)  8 (ID,DI)

    extern __shared__ double lds[];
    double* const            G  = lds;              // [16][NQP]
    double* const            Pt = G + 16 * NQP;     // [4][N2][NQ]
    double* const            A  = Pt + 4 * N2 * NQ; // [PAIRS][9][NQ (qz)][AROW (qy, padded)]

    const int     tid = threadIdx.x;
    // Workgroup -> (element, unknown pair).  Workgroups are dealt round-robin to the 8 XCDs: with xcd_group the U (U + 1) / 2
    // workgroups of ONE element have the same index modulo 8, i.e. sit on one XCD and share its L2 -- in stored mode every
    // 64-byte line of K_e receives its 8 entries from 4 pair-workgroups (the unknown u' is the fastest index of a row), so
    // they should at least meet in one L2.  Otherwise blockIdx = pair + NP * element.
    static_assert(!TILED || BLOCKS == 0);
    // (a single unknown has no off-diagonal block: that kernel is instantiated but never launched)
    constexpr int NP = TILED ? U * U : (BLOCKS == 1 ? U : (BLOCKS == 2 ? cmax(U * (U - 1) / 2, 1) : U * (U + 1) / 2));
    int64_t       el;
    int           rem;
    if (xcd_group)
    {
        const int64_t chunk = blockIdx.x / (8 * NP);
        const int     w     = int(blockIdx.x - chunk * (8 * NP));
        el                  = chunk * 8 + (w & 7);
        rem                 = w >> 3;
        if (el >= a.elem_count)
            return;
    }
    else
    {
        el  = blockIdx.x / NP;
        rem = int(blockIdx.x - el * NP);
    }
    int u = 0; // unknown pair of this workgroup: u' <= u (TILED: every pair)
    if constexpr (TILED)
    {
        u   = rem / U;
        rem = rem - u * U;
    }
    else if constexpr (BLOCKS == 1)
        u = rem;
    else if constexpr (BLOCKS == 2)
    {
        u = 1;
        while (rem >= u)
        {
            rem -= u;
            ++u;
        }
    }
    else
        while (rem > u)
        {
            rem -= u + 1;
            ++u;
        }
    const int up = rem;

    // ---- product tables and G
    for (int i = tid; i < 4 * N2 * NQ; i += NT)
    {
        const int t = i / (N2 * NQ), r = i - t * (N2 * NQ), bb = r / NQ, q = r - bb * NQ;
        const int b1 = bb % N1, b1p = bb / N1;
        const double* T0 = a.tables + ((t & 1) ? TL.offD() : TL.offI());
        const double* T1 = a.tables + ((t & 2) ? TL.offD() : TL.offI());
        Pt[i]            = T0[b1 * NQ + q] * T1[b1p * NQ + q];
    }
    const double* cel = cbuf + el * NQP * CS;
    for (int q = tid; q < NQP; q += NT)
    {
        const double* cq = cel + q; // [entry][q]
        const double  w  = cq[(CS - 1) * NQP];
        double        cu[E][4], cp[E][4];
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                cu[e][k] = cq[((e * U + u) * 4 + k) * NQP];
                cp[e][k] = cq[((e * U + up) * 4 + k) * NQP];
            }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int kp = 0; kp < 4; ++kp)
            {
                double g = 0.;
#pragma unroll
                for (int e = 0; e < E; ++e)
                    g += cu[e][k] * cp[e][kp];
                G[(k * 4 + kp) * NQP + q] = w * g;
            }
    }
    __syncthreads();

    // An iteration works on N1 SLOTS, each an index pair (bx, bx'), and on rows (slot, by, by').
    // Off-diagonal blocks (and TILED, and all blocks of the one-launch kernel): iteration j holds the pairs (bx = slot, bx' = j);
    // this thread's row is fixed.
    // The diagonal-block kernel (BLOCKS == 1): diagonal blocks (u' == u) are symmetric in (b, b'): only the half "bx' < bx, or bx' == bx and (by' < by, or by' == by and
    // bz' <= bz)" is formed and mirrored -- x-major, because then whole (bx, bx') pairs drop out: bx' = j needs the N1 - j pairs
    // bx >= j, so bx' = j and bx' = N1 - j share one iteration of N1 slots (slot s < N1 - j: (j + s, j), else (s, N1 - j)), and a
    // slot with bx == bx' has rows by' <= by only: N1 / 2 + 1 iterations instead of N1 (order 6: 4 instead of 7, of 322 and
    // 3 x 301 rows), with a per-iteration assignment of the threads to rows.
    constexpr int NTRI    = N1 * (N1 + 1) / 2; // rows (by' <= by) of a slot with bx == bx'
    // (TILED: by' runs fastest over the threads, so that consecutive threads write consecutive column nodes of one row node)
    const int  row = tid < ROWS ? tid : 0;

    double  csum = 0.;
    double* Kel  = a.K ? a.K + (elem0 + el) * int64_t(ND) * ND : nullptr;

    // DPPT: the 1-D tables of stage 3 (I, D and the even-odd tables of I^T, D^T) as DPP-row operands instead of scalar loads
    constexpr bool DPPT  = C::dpp2(TILED, BLOCKS); // (the one-launch kernel for the stored row-major matrices keeps the scalar tables)
    constexpr bool PRODUCER    = C::producer(TILED, BLOCKS);
    constexpr int  ROW_THREADS = C::rowThreadsFor(TILED, BLOCKS), A_D = N1 * 9 * NQ * AROW;
    constexpr bool DPP2  = DPPT; // stage 2 with the slot's A entries as DPP-row operands (needs the unit layout of the rows)
    constexpr int  NCH_T = (N1 * NQ + 15) / 16, EO_N = ((NQ + 1) / 2 + NQ / 2) * ((N1 + 1) / 2), NCH_E = (EO_N + 15) / 16;
    [[maybe_unused]] double tabD[NCH_T], tabEI[NCH_E], tabED[NCH_E];
    [[maybe_unused]] double firstEI[(N1 + 1) / 2 + 1], firstED[(N1 + 1) / 2 + 1]; // (scalar: We[0][.] and Wo[0][middle] of the two tables)
    if constexpr (DPPT)
    {
        const int j = tid & 15;
#pragma unroll
        for (int c = 0; c < NCH_T; ++c)
        {
            const int t = 16 * c + j < N1 * NQ ? 16 * c + j : N1 * NQ - 1;
            tabD[c]     = a.tables[TL.offD() + t];
            // (opaque: a chunk whose lanes all hold the clamped last entry is wave-uniform -- the compiler kept it in scalar
            // registers and moved it into the vector register right in front of the DPP read: the hazard of the helpers' comment)
            asm volatile("" : "+v"(tabD[c]));
        }
#pragma unroll
        for (int c = 0; c < NCH_E; ++c)
        {
            const int t = 16 * c + j < EO_N ? 16 * c + j : EO_N - 1;
            tabEI[c]    = a.tables[TL.offEoIt() + t];
            tabED[c]    = a.tables[TL.offEoDt() + t];
            asm volatile("" : "+v"(tabEI[c]));
            asm volatile("" : "+v"(tabED[c]));
        }
        constexpr int RI_ = (NQ + 1) / 2, RO_ = (N1 + 1) / 2;
        const __attribute__((address_space(4))) double* const sI =
            reinterpret_cast< const __attribute__((address_space(4))) double* >(reinterpret_cast< uintptr_t >(a.tables + TL.offEoIt()));
        const __attribute__((address_space(4))) double* const sD =
            reinterpret_cast< const __attribute__((address_space(4))) double* >(reinterpret_cast< uintptr_t >(a.tables + TL.offEoDt()));
#pragma unroll
        for (int q = 0; q < RO_; ++q)
        {
            firstEI[q] = sI[q];
            firstED[q] = sD[q];
        }
        firstEI[RO_] = sI[RI_ * RO_ + N1 / 2];
        firstED[RO_] = sD[RI_ * RO_ + N1 / 2];
    }
    // (DIAG: the diagonal-block kernel, BLOCKS == 1)
    auto iterations = [&]< bool DIAG >() {
    for (int iter = 0; iter < (DIAG ? N1 / 2 + 1 : N1); ++iter)
    {
        // the second list of a diagonal block's iteration: slots s >= N1 - iter hold (s, N1 - iter)
        const bool two_lists = DIAG && iter >= 1 && iter < N1 - iter;
        const int  split     = DIAG ? N1 - iter : N1;                    // slots below: first list
        const int  n_slots   = DIAG && !two_lists ? N1 - iter : N1;      // (the middle bx' of an even N1 stands alone)
        auto       slotPairAt = [&](int it_, int s_, int& bx_, int& bxp_) { // (the pairs of iteration it_)
            if (!DIAG)
            {
                bx_  = s_;
                bxp_ = it_;
            }
            else if (s_ < N1 - it_)
            {
                bx_  = it_ + s_;
                bxp_ = it_;
            }
            else
            {
                bx_  = s_;
                bxp_ = N1 - it_;
            }
        };
        auto slotPair = [&](int s_, int& bx_, int& bxp_) { slotPairAt(iter, s_, bx_, bxp_); };
        // ---- stage 1 on the producer wave, for iteration it_ into Aout: lane (qy, qz) forms A of ALL slots.  With
        // h[term] = G[term][qx] T_s'[bx'][qx] and, per group, hI = the sum of its h with s = I, hD = ... with s = D (only the groups
        // 0, 2, 5 have one): A[slot][g] = sum_qx I[bx][qx] hI[g] + D[bx][qx] hD[g].  A diagonal block's iteration has two bx'
        // (the two lists): two sets of sums
        [[maybe_unused]] auto produce = [&](int it_, double* Aout) {
            const int qyz = tid - ROW_THREADS;
            if (qyz >= NQ * NQ)
                return;
            const bool two_  = DIAG && it_ >= 1 && it_ < N1 - it_;
            const int  nsl_  = DIAG && !two_ ? N1 - it_ : N1, spl_ = DIAG ? N1 - it_ : N1;
            const __attribute__((address_space(4))) double* const tI =
                reinterpret_cast< const __attribute__((address_space(4))) double* >(reinterpret_cast< uintptr_t >(a.tables + TL.offI()));
            const __attribute__((address_space(4))) double* const tD =
                reinterpret_cast< const __attribute__((address_space(4))) double* >(reinterpret_cast< uintptr_t >(a.tables + TL.offD()));
            double acc[N1][9];
#pragma unroll
            for (int sl = 0; sl < N1; ++sl)
#pragma unroll
                for (int g = 0; g < 9; ++g)
                    acc[sl][g] = 0.;
            const double* gq   = G + qyz * NQ;
            const int     bxpA = it_, bxpB = two_ ? N1 - it_ : it_;
#pragma unroll
            for (int qx = 0; qx < NQ; ++qx)
            {
                const double ipA = tI[bxpA * NQ + qx], dpA = tD[bxpA * NQ + qx], ipB = tI[bxpB * NQ + qx], dpB = tD[bxpB * NQ + qx];
                double       hIA[9], hDA[9], hIB[9], hDB[9];
#pragma unroll
                for (int g = 0; g < 9; ++g)
                    hIA[g] = hDA[g] = hIB[g] = hDB[g] = 0.;
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int kp = 0; kp < 4; ++kp)
                    {
                        constexpr int gtab[4][4] = {{0, 4, 5, 6}, {1, -1, 7, -1}, {2, 8, -1, -1}, {3, -1, -1, -1}}; // [ty][tz]
                        const int     ty = (k == 2) + 2 * (kp == 2), tz = (k == 3) + 2 * (kp == 3);
                        const int     g  = gtab[ty][tz];
                        const double  gv = gq[(k * 4 + kp) * NQP + qx];
                        const double  hA = gv * (kp == 1 ? dpA : ipA);
                        (k == 1 ? hDA[g] : hIA[g]) += hA;
                        if (two_)
                        {
                            const double hB = gv * (kp == 1 ? dpB : ipB);
                            (k == 1 ? hDB[g] : hIB[g]) += hB;
                        }
                    }
#pragma unroll
                for (int sl = 0; sl < N1; ++sl)
                    if (sl < nsl_)
                    {
                        int bx_, bxp_;
                        slotPairAt(it_, sl, bx_, bxp_);
                        const double ib = tI[bx_ * NQ + qx], db = tD[bx_ * NQ + qx];
                        const bool   lb = two_ && sl >= spl_; // second list
#pragma unroll
                        for (int g = 0; g < 9; ++g)
                        {
                            acc[sl][g] += ib * (lb ? hIB[g] : hIA[g]);
                            if (g == 0 || g == 2 || g == 5)
                                acc[sl][g] += db * (lb ? hDB[g] : hDA[g]);
                        }
                    }
            }
#pragma unroll
            for (int sl = 0; sl < N1; ++sl)
                if (sl < nsl_)
#pragma unroll
                    for (int g = 0; g < 9; ++g)
                        Aout[(sl * NQ + qyz % NQ) * 9 * NQ + g * NQ + qyz / NQ] = acc[sl][g]; // (the DPP kernels' layout [slot][qy][g][qz])
        };
        const bool          is_producer = PRODUCER && tid >= ROW_THREADS;
        const double* const Acur        = PRODUCER ? A + (iter & 1) * A_D : A;
        // this thread's row of the iteration: (slot pp, by, by'), its pair (bx, bx') and its rows of the y product tables
        bool has_row;
        int  pp, by, byp;
        [[maybe_unused]] int n_units_iter = 0; // DPP2: 16-lane units with rows in this iteration
        [[maybe_unused]] int row_t        = row; // TILED: this thread's row (slot, by, by') in the tiled layout's order
        if constexpr (DPP2)
        {
            // 16-lane units: a slot with bx == bx' (DIAG) has the NTRI rows by' <= by in (NTRI + 15) / 16 units, every other slot its
            // N2 rows in UNITS units; order: the triangle of slot 0, the triangle of slot `split` (second list), the other slots
            constexpr int UT = (NTRI + 15) / 16, UF = C::UNITS;
            const int     unit = tid >> 4, j = tid & 15;
            const int     n_tri_slots = !DIAG ? 0 : (two_lists ? 2 : 1);
            n_units_iter              = n_tri_slots * UT + (n_slots - n_tri_slots) * UF;
            if (unit < n_tri_slots * UT)
            {
                pp    = unit < UT ? 0 : split;
                int t = (unit < UT ? unit : unit - UT) * 16 + j;
                has_row = t < NTRI;
                t       = has_row ? t : 0;
                by      = 0;
                while (t > by) // triangle index -> (by, by' <= by)
                {
                    t -= by + 1;
                    ++by;
                }
                byp = t;
            }
            else
            {
                const int uu = unit - n_tri_slots * UT, f = uu / UF, r = (uu - f * UF) * 16 + j; // f-th slot among the full ones
                pp           = !DIAG ? f : f + 1 + (two_lists && f + 1 >= split ? 1 : 0);
                has_row      = pp < n_slots && r < N2;
                pp           = pp < n_slots ? pp : 0;
                const int bbt = r < N2 ? r : 0;
                by            = TILED ? bbt / N1 : bbt % N1; // (TILED: by' runs fastest over the lanes, cf. the static rows)
                byp           = TILED ? bbt % N1 : bbt / N1;
                row_t         = pp * N2 + bbt;
            }
        }
        else if (!DIAG)
        {
            has_row       = tid < ROWS;
            pp            = row / N2;
            const int bbt = row - pp * N2;
            by            = TILED ? bbt / N1 : bbt % N1;
            byp           = TILED ? bbt % N1 : bbt / N1;
        }
        else
        {
            // rows in the order: the triangle of slot 0, the triangle of slot `split` (second list), the other slots in full
            const int n_tri = two_lists ? 2 * NTRI : NTRI;
            int       t     = tid;
            if (t < n_tri)
            {
                pp = t < NTRI ? 0 : split;
                t  = t < NTRI ? t : t - NTRI;
                by = 0;
                while (t > by) // triangle index -> (by, by' <= by)
                {
                    t -= by + 1;
                    ++by;
                }
                byp     = t;
                has_row = true;
            }
            else
            {
                t -= n_tri;
                const int f = t / N2, bbt = t - f * N2; // f-th slot among those with bx != bx'
                pp          = f + 1 + (two_lists && f + 1 >= split ? 1 : 0);
                by          = bbt % N1;
                byp         = bbt / N1;
                has_row     = pp < n_slots;
                pp          = has_row ? pp : 0;
            }
        }
        int bx_row, bxp;
        slotPair(pp, bx_row, bxp);
        if constexpr (TILED) // (position of the row in the tiled layout: by its pair's bx, not by its slot)
            row_t = bx_row * N2 + by * N1 + byp;
        const int bb = by + N1 * byp; // pair index of the y product tables
        // ---- stage 1: A[bx][g][qz][qy] for the pairs (bx, bx' = bxp): one thread per (bx, qy, qz) forms all nine groups from its
        // 16 x nq entries of G and the 4 x nq entries of the x product table of its pair -- one straight-line body of 16
        // independent 7-term sums with compile-time term lists (the earlier form dealt (group, bx, qy, qz) items to the
        // threads: a switch per item, the product-table row re-read per term, three to five LDS reads in flight)
        if (!PRODUCER || iter == 0)
        {
        for (int it = tid; it < n_slots * NQ * NQ; it += NT)
        {
            const int bx = it / (NQ * NQ), qyz = it - bx * (NQ * NQ); // (bx: the slot)
            int       bx1, bx1p;
            slotPair(bx, bx1, bx1p);
            const int pair = bx1p * N1 + bx1; // pair index (bx, bx') = bx + N1 bx' of the x product table
            double    pxv[4][NQ];
#pragma unroll
            for (int tx = 0; tx < 4; ++tx)
#pragma unroll
                for (int qx = 0; qx < NQ; ++qx)
                    pxv[tx][qx] = Pt[(tx * N2 + pair) * NQ + qx];
            double accg[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int kp = 0; kp < 4; ++kp)
                {
                    // group (ty, tz) of the term (k, k'): type of a direction = s + 2 s' with s = (k == d + 1), s' = (k' == d + 1)
                    constexpr int gtab[4][4] = {{0, 4, 5, 6}, {1, -1, 7, -1}, {2, 8, -1, -1}, {3, -1, -1, -1}}; // [ty][tz]
                    const int     ty = (k == 2) + 2 * (kp == 2), tz = (k == 3) + 2 * (kp == 3), tx = (k == 1) + 2 * (kp == 1);
                    const int     g  = gtab[ty][tz];
                    const double* gq = G + (k * 4 + kp) * NQP + qyz * NQ;
                    double        t  = 0.;
#pragma unroll
                    for (int qx = 0; qx < NQ; ++qx)
                        t += pxv[tx][qx] * gq[qx];
                    accg[g] += t;
                }
#pragma unroll
            for (int g = 0; g < 9; ++g)
                if constexpr (DPP2) // [slot][qy][g][qz]: 16 consecutive entries belong to 16 different sums of stage 2
                    A[(bx * NQ + qyz % NQ) * 9 * NQ + g * NQ + qyz / NQ] = accg[g];
                else
                    A[((bx * 9 + g) * NQ + qyz / NQ) * AROW + qyz % NQ] = accg[g];
        }
        __syncthreads();
        }
        // (A of iteration 0 by the cooperative stage 1 above, all waves: the producer alone would take as long as a whole iteration)
        if constexpr (PRODUCER)
            if (is_producer && iter + 1 < (DIAG ? N1 / 2 + 1 : N1))
                produce(iter + 1, A + ((iter + 1) & 1) * A_D);
        // (DPPT: whole waves run stages 2 and 3 -- a DPP operand comes from a lane of the row that must be active; lanes without a
        // row work on row 0 and are masked where results leave the registers)
        const int  n_rows_iter = !DIAG ? ROWS : (two_lists ? 2 * NTRI : NTRI) + (n_slots - (two_lists ? 2 : 1)) * N2;
        const bool wave_rows   = DPP2 ? (tid >> 6) * 4 < n_units_iter && !is_producer : (tid & ~63) < n_rows_iter;
        if (DPPT ? wave_rows : has_row)
        {
            // ---- stage 2 in registers: B[tz][qz] = sum_{ty} sum_qy P[ty][(by,by')][qy] A[(ty,tz)][qz][qy]
            // (this row's 4 nq entries of the y product table are re-read from LDS in every iteration: held across stage 3
            // they cost 2 * 4 nq registers where stage 3 has none to spare)
            double py[4][NQ];
            {
                const double* pt = Pt + opaqueOffset(bb * NQ);
#pragma unroll
                for (int ty = 0; ty < 4; ++ty)
#pragma unroll
                    for (int qy = 0; qy < NQ; ++qy)
                        py[ty][qy] = pt[ty * N2 * NQ + qy];
            }
            double B[4][NQ];
            if constexpr (DPP2)
            {
                // the slot's 9 nq^2 entries of A (stored [qy][g][qz] by stage 1: consecutive FMAs go to different sums) in chunks of 16, one entry per lane of a DPP row (64 bytes of LDS per row and chunk
                // instead of 8 bytes per lane and FMA); chunk c + 1 is requested before the FMAs of chunk c
                constexpr int    NA = 9 * NQ * NQ, NCA = (NA + 15) / 16;
                const double* const Apos = Acur + pp * 9 * NQ * AROW + (tid & 15);
                static_assert(AROW == NQ);
#pragma unroll
                for (int tz = 0; tz < 4; ++tz)
#pragma unroll
                    for (int qz = 0; qz < NQ; ++qz)
                        B[tz][qz] = 0.;
                double ach[2][1];
                ach[0][0] = Apos[0];
                staticFor< NCA >([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    if constexpr (c + 1 < NCA)
                    {
                        // (the last chunk is short: its lanes beyond the array read the array's last entries again)
                        constexpr int last = NA - 16 * (c + 1); // entries in chunk c + 1
                        ach[(c + 1) & 1][0] = last >= 16 ? Apos[16 * (c + 1)] : Apos[(tid & 15) < last ? 16 * (c + 1) : NA - 16];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    staticFor< (NA - 16 * c < 16 ? NA - 16 * c : 16) >([&](auto kc) {
                        constexpr int f = 16 * c + decltype(kc)::value, qy = f / (9 * NQ), g = (f / NQ) % 9, qz = f % NQ; // [qy][g][qz]
                        // group -> (ty, tz): 0 (II,II) 1 (DI,II) 2 (ID,II) 3 (DD,II) 4 (II,DI) 5 (II,ID) 6 (II,DD) 7 (DI,ID) 8 (ID,DI)
                        constexpr int tyg[9] = {0, 1, 2, 3, 0, 0, 0, 1, 2}, tzg[9] = {0, 0, 0, 0, 1, 2, 3, 2, 1};
                        fmaTab< decltype(kc)::value >(B[tzg[g]][qz], ach[c & 1], py[tyg[g]][qy]);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            else if constexpr (!TILED)
            {
#pragma unroll
            for (int tz = 0; tz < 4; ++tz)
#pragma unroll
                for (int qz = 0; qz < NQ; ++qz)
                {
                    double acc = 0.;
#pragma unroll
                    for (int ty = 0; ty < 4; ++ty)
                    {
                        constexpr int gtab[4][4] = {{0, 4, 5, 6}, {1, -1, 7, -1}, {2, 8, -1, -1}, {3, -1, -1, -1}}; // [ty][tz]
                        const int     g = gtab[ty][tz];
                        if (g >= 0)
                        {
                            const double* aq = A + ((pp * 9 + g) * NQ + qz) * AROW;
#pragma unroll
                            for (int qy = 0; qy < NQ; ++qy)
                                acc += py[ty][qy] * aq[qy];
                        }
                    }
                    B[tz][qz] = acc;
                }
            }
            else
            {
                // TILED at orders < 4 (orders >= 4 take the DPP form above; measured at order 6 before that existed: 204 registers,
                // the stores of stage 3 in flight): the 9 nq dot products (group, qz) of nq terms as a software
                // pipeline -- the nq entries of A of item i + 1 are requested before the FMAs of item i, and the scheduler is told to
                // keep it that way.  Left to itself the compiler forms one dependent chain per B entry and requests each pair of
                // operands two instructions before their FMAs.  (+7 % for the tiled store, 142 -> 152 k matrices/s at order 6; the
                // streaming kernels at 256 registers lose 1.5 % with it: profiles/r03_assembly_stage2.log)
                constexpr int            NIT2 = 9 * NQ;
                constexpr Stage2Items< NQ > items{};
                const double* const      Apos = A + pp * 9 * NQ * AROW;
                double                   abuf[2][NQ];
                auto                     request = [&](double (&dst)[NQ], const Stage2Item it) {
                    const double* aq = Apos + (it.g * NQ + it.qz) * AROW;
#pragma unroll
                    for (int qy = 0; qy < NQ; ++qy)
                        dst[qy] = aq[qy];
                };
                request(abuf[0], items.v[0]);
#pragma unroll
                for (int i = 0; i < NIT2; ++i)
                {
                    const Stage2Item it = items.v[i];
                    if (i + 1 < NIT2)
                        request(abuf[(i + 1) & 1], items.v[i + 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    // (two partial sums: the chain of dependent FMAs is half as long)
                    double acc0 = i > 0 && items.v[i > 0 ? i - 1 : 0].tz == it.tz && items.v[i > 0 ? i - 1 : 0].qz == it.qz ? B[it.tz][it.qz] : 0.;
                    double acc1 = 0.;
#pragma unroll
                    for (int qy = 0; qy < NQ; ++qy)
                        if (qy % 2 == 0)
                            acc0 += py[it.ty][qy] * abuf[i & 1][qy];
                        else
                            acc1 += py[it.ty][qy] * abuf[i & 1][qy];
                    B[it.tz][it.qz] = acc0 + acc1;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- stage 3: M[row][(bz,bz')] = sum_{tz,qz} B[tz][qz] Pz[(bz,bz')][(tz,qz)], the table through scalar loads
            const int rowb = bx_row + N1 * by, rowbp = bxp + N1 * byp;
            // Output bookkeeping hoisted out of the entry loops: which half of a diagonal block an entry belongs to is decided by
            // the row (rdlt) unless bx' == bx and by' == by, and the checksum weight 1 + (31 gi + 17 gj) mod 7 = 1 + 3 (gi + gj) mod 7 does not depend
            // on (bz, bz') when N2 U is a multiple of 7 (order 6: 196): three partial sums per row instead of ~15 integer
            // instructions per entry.  All loops are unrolled: (bz, bz') are compile-time and the scalar loads are issued ahead
            // of their use.  (History at order 6: per-entry bookkeeping + rolled columns 162 k matrices/s -> unrolled, hoisted
            // 193 k -> factorised 218 k.)
            constexpr bool wgt_const = (N2 * U) % 7 == 0;
            // Otherwise the weight depends on (bz, bz') through bz + bz' only: 1 + (c0 + step (bz + bz')) mod 7 with c0 of the row and
            // step = 3 N2 U mod 7 -- 2 n - 1 weights per row (an add and a conditional subtract each) instead of ~15 integer
            // instructions per entry (orders 4, 5, 7: the bookkeeping cost more than the entries)
            [[maybe_unused]] double wt[2 * N1 - 1];
            if constexpr (!TILED && !wgt_const)
            {
                constexpr int step = (3 * N2 * U) % 7;
                const int     c0   = (3 * ((rowb + rowbp) * U + u + up)) % 7;
#pragma unroll
                for (int sb = 0; sb < 2 * N1 - 1; ++sb)
                {
                    const int v = c0 + (step * sb) % 7; // < 14
                    wt[sb]      = double(1 + (v >= 7 ? v - 7 : v));
                }
            }
            // diagonal blocks: > 0 the whole row belongs to the formed half, == 0 (bx' == bx, by' == by) its columns bz' <= bz do
            const int      rdlt      = bx_row != bxp ? 1 : by - byp;
            // BLOCKS == 0 (one launch over all pairs: the stored row-major matrices) keeps the seven iterations and the z-major
            // half b' <= b of its diagonal blocks -- there the four workgroups that fill a 64-byte line of K_e should pass
            // through the same b_x' at about the same time (profiles/r03_assembly_diagonal_blocks.log)
            const bool     zhalf     = !DIAG && BLOCKS == 0 && !TILED && u == up;
            const int      dlt       = rowb - rowbp; // sign of b - b' on the bz == bz' columns
            double         s_lo = 0., s_eq = 0., s_up = 0.; // sums over the columns with bz > bz', bz == bz', bz < bz'
            // Stage 3 is sum-factorised once more: the z product table is itself a product, Pz[(bz,bz')][(s + 2 s', qz)] =
            // T_s[bz][qz] T_s'[bz'][qz] (T_0 = I, T_1 = D), so M[bz][bz'] = sum_{s',qz} W[bz][s'][qz] T_s'[bz'][qz] with
            // W[bz][s'][qz] = sum_s B[s + 2 s'][qz] T_s[bz][qz]: 4 nq n + 2 nq n^2 = 882 FMAs per row instead of 4 nq n^2 = 1 372,
            // and every scalar operand of the second step (the 1-D tables through scalar loads) serves a block of ZB values of bz
            // instead of one entry -- the first form waited for its scalar loads (56 scalar registers per column)
            constexpr int ZB = (TILED && P >= 7) || (!TILED && P == 6) ? 1 : 2; // (the tiled store at order 7: one, 39 -> 46 k stored matrices/s; orders 5, 6: no difference; the streaming kernels at order 6: one -- the off-diagonal kernel spills 1 instead of 10 registers, 534 -> 539 k matrices/s)  values of bz per block: W of a block is 2 ZB nq doubles of registers (4: the same rate, more spills;
                                  // with the DPP tables 1 / 2 / 4 at order 6: 458 / 450 / 446 k, but 1 loses 12 % at order 4)
            const __attribute__((address_space(4))) double* const tIz =
                reinterpret_cast< const __attribute__((address_space(4))) double* >(reinterpret_cast< uintptr_t >(a.tables + TL.offI()));
            const __attribute__((address_space(4))) double* const tDz =
                reinterpret_cast< const __attribute__((address_space(4))) double* >(reinterpret_cast< uintptr_t >(a.tables + TL.offD()));
            const __attribute__((address_space(4))) double* const eoItz =
                reinterpret_cast< const __attribute__((address_space(4))) double* >(reinterpret_cast< uintptr_t >(a.tables + TL.offEoIt()));
            const __attribute__((address_space(4))) double* const eoDtz =
                reinterpret_cast< const __attribute__((address_space(4))) double* >(reinterpret_cast< uintptr_t >(a.tables + TL.offEoDt()));
#pragma unroll
            for (int b0 = 0; b0 < N1; b0 += ZB)
            {
                double W[ZB][2][NQ];
#pragma unroll
                for (int bi = 0; bi < ZB; ++bi)
                    if (b0 + bi < N1)
                    {
#pragma unroll
                        for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                            for (int qz = 0; qz < NQ; ++qz)
                                if constexpr (!DPPT)
                                    W[bi][sp][qz] = B[2 * sp][qz] * tIz[(b0 + bi) * NQ + qz] + B[2 * sp + 1][qz] * tDz[(b0 + bi) * NQ + qz];
                    }
                if constexpr (DPPT)
                    staticFor< ZB >([&](auto bic) {
                        constexpr int bi = decltype(bic)::value;
                        // (b0 is the unrolled loop's variable, not a constant expression: one copy of the block per value)
                        staticFor< (N1 + ZB - 1) / ZB >([&](auto blk) {
                            constexpr int bz = decltype(blk)::value * ZB + bi;
                            if constexpr (bz < N1)
                                if (b0 == decltype(blk)::value * ZB)
                                    staticFor< 2 * NQ >([&](auto ic) {
                                        constexpr int sp = decltype(ic)::value / NQ, qz = decltype(ic)::value % NQ;
                                        double        w  = B[2 * sp][qz] * tIz[bz * NQ + qz]; // (scalar operand: no product with DPP)
                                        fmaTab< bz * NQ + qz >(w, tabD, B[2 * sp + 1][qz]);
                                        W[bi][sp][qz] = w;
                                    });
                        });
                    });
                // second step with the even-odd decomposition of the two 1-D tables (I^T symmetric, D^T antisymmetric): 2 x 34
                // instead of 2 nq n = 98 instructions per b_z, half as many scalar operands
                double Mz[ZB][N1];
#pragma unroll
                for (int bi = 0; bi < ZB; ++bi)
                    if (b0 + bi < N1)
                    {
                        if constexpr (DPPT)
                        {
                            sweepEODpp< NQ, N1, false, false >(W[bi][0], Mz[bi], tabEI, firstEI);
                            sweepEODpp< NQ, N1, true, true >(W[bi][1], Mz[bi], tabED, firstED);
                        }
                        else
                        {
                            sweepEOScalar< NQ, N1, false, false >(W[bi][0], Mz[bi], eoItz);
                            sweepEOScalar< NQ, N1, true, true >(W[bi][1], Mz[bi], eoDtz);
                        }
                    }
#pragma unroll
                for (int bzp = 0; bzp < N1; ++bzp)
#pragma unroll
                    for (int bi = 0; bi < ZB; ++bi)
                        if (b0 + bi < N1 && !(zhalf && bzp > b0 + bi)) // (z-major half: b_z' > b_z is the mirror image)
                        {
                            const int    bz = b0 + bi;
                            const double m  = Mz[bi][bzp];
                            const int  b = rowb + N2 * bz, bp = rowbp + N2 * bzp;
                            // diagonal blocks: one half only (mirrored)
                            const bool skip = DIAG ? rdlt == 0 && bzp > bz : zhalf && (bz != bzp ? bzp > bz : dlt < 0);
                            const int  gi = b * U + u, gj = bp * U + up;
                            if constexpr (TILED)
                            {
                                constexpr int64_t NNc = int64_t(N1) * N2;
                                if (has_row && !(DIAG && rdlt == 0 && bzp > bz)) // (TMODE 2: bx' == bx, by' == by: the columns bz' <= bz)
                                    Kel[(u * U + up) * NNc * NNc + ((int64_t(bxp) * N1 + bz) * ROWS + row_t) * N1 + bzp] = m;
                            }
                            else if (Kel && !skip && has_row)
                            {
                                Kel[int64_t(gi) * ND + gj] = m;
                                if (gi != gj)
                                    Kel[int64_t(gj) * ND + gi] = m;
                            }
                            if constexpr (TILED)
                                ;
                            else if constexpr (wgt_const)
                            {
                                s_lo += bz > bzp ? m : 0.;
                                s_eq += bz == bzp ? m : 0.;
                                s_up += bz < bzp ? m : 0.;
                            }
                            else
                            {
                                s_lo += bz > bzp ? wt[bz + bzp] * m : 0.;
                                s_eq += bz == bzp ? wt[bz + bzp] * m : 0.;
                                s_up += bz < bzp ? wt[bz + bzp] * m : 0.;
                            }
                        }
            }
            if constexpr (!TILED)
            {
                // weight of this (row, column-x) pair (already in the sums if it varies with bz + bz'); entries of off-diagonal blocks
                // stand for themselves and their mirror image, in diagonal blocks those of the formed half do, b' == b counts once
                const double wgt = wgt_const ? double(1 + (3 * ((rowb + rowbp) * U + u + up)) % 7) : 1.;
                const double f_eq = dlt > 0 ? 2. : (dlt == 0 ? 1. : 0.); // (z-major half)
                csum += (has_row ? wgt : 0.) * (DIAG ? (rdlt > 0 ? 2. * (s_lo + s_eq + s_up) : 2. * s_lo + s_eq)
                                                     : (zhalf ? 2. * s_lo + f_eq * s_eq : 2. * (s_lo + s_eq + s_up)));
            }
        }
        __syncthreads(); // A is rewritten by the next iteration
    }
    };
    iterations.template operator()< BLOCKS == 1 || TMODE == 2 >();
    if (a.checksum)
    {
        // fixed-order reduction over the workgroup, one atomic per workgroup
        double* red = A; // (free after the last barrier)
        red[tid]    = csum;
        __syncthreads();
        constexpr int P2 = NT <= 64 ? 64 : (NT <= 128 ? 128 : (NT <= 256 ? 256 : (NT <= 512 ? 512 : 1024))); // (NT = 384 at order 6)
        for (int w = P2 / 2; w > 0; w >>= 1)
        {
            if (tid < w && tid + w < NT)
                red[tid] += red[tid + w];
            __syncthreads();
        }
        if (tid == 0)
            unsafeAtomicAdd(a.checksum + elem0 + el, red[0]);
    }
}

// K: batch [a.elem_begin, a.elem_begin + a.elem_count); output slot elem0 + i for the i-th element of the batch
template < typename K, int P, int NQ >
int launchAssemble(const ElemArgs& a, const void* kparam_blob, hipStream_t stream)
{
    constexpr int U = K::params.n_unknowns, E = K::params.n_equations;
    using C = GemmCfg< P, NQ, U, E >;
    if (a.elem_count <= 0)
        return 0;
    K kern{};
    if (kparam_blob)
        __builtin_memcpy(&kern, kparam_blob, sizeof(K));
    constexpr size_t ldc    = coeffLdsBytes< K, P, NQ >();
    constexpr bool   coef_gs = ldc > lds_limit_bytes; // the field buffers of the coefficient kernel exceed the LDS: global scratch
    auto             kc  = assembleCoeffKernel< K, P, NQ, coef_gs >;
    auto             kg  = assembleGemmKernel< K, P, NQ >;
    // the dynamic-LDS attributes of the three kernels are set once PER DEVICE, under a lock (several contexts of one
    // process may sit on different GPUs)
    using S = SfAsmCfg< P, NQ >;
    {
        static bool       attr_set[64] = {};
        static std::mutex attr_mutex;
        int               dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64)
        {
            setError("device index %d not supported", dev);
            return -3;
        }
        std::lock_guard< std::mutex > lock{attr_mutex};
        if (!attr_set[dev])
        {
            bool ok = (coef_gs || hipFuncSetAttribute(reinterpret_cast< const void* >(kc), hipFuncAttributeMaxDynamicSharedMemorySize, int(ldc)) == hipSuccess) &&
                      hipFuncSetAttribute(reinterpret_cast< const void* >(kg), hipFuncAttributeMaxDynamicSharedMemorySize, int(C::lds)) == hipSuccess;
            if constexpr (S::feasible)
                ok = ok && hipFuncSetAttribute(reinterpret_cast< const void* >(assembleSumfactKernel< K, P, NQ >),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, int(S::ldsFor(false, 0))) == hipSuccess &&
                     hipFuncSetAttribute(reinterpret_cast< const void* >(assembleSumfactKernel< K, P, NQ, 1 >),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, int(S::ldsFor(true, 0))) == hipSuccess &&
                     hipFuncSetAttribute(reinterpret_cast< const void* >(assembleSumfactKernel< K, P, NQ, 2 >),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, int(S::ldsFor(true, 0))) == hipSuccess &&
                     hipFuncSetAttribute(reinterpret_cast< const void* >(assembleSumfactKernel< K, P, NQ, 0, 1 >),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, int(S::ldsFor(false, 1))) == hipSuccess &&
                     hipFuncSetAttribute(reinterpret_cast< const void* >(assembleSumfactKernel< K, P, NQ, 0, 2 >),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, int(S::ldsFor(false, 2))) == hipSuccess;
            if (!ok)
            {
                setError("hipFuncSetAttribute failed for the assembly kernels");
                return -3;
            }
            attr_set[dev] = true;
        }
    }
    double* cbuf = a.workspace; // coeffStride * nq^3 doubles per element, + 1 flag
    if constexpr (coef_gs)
    {
        const int64_t  max_wgs = 2 * int64_t(deviceComputeUnits());
        const unsigned grid    = static_cast< unsigned >(a.elem_count < max_wgs ? a.elem_count : max_wgs);
        ElemArgs       ag      = a;
        ag.scratch             = a.scratch_alloc ? a.scratch_alloc(a.scratch_owner, ldc * grid) : nullptr;
        if (!ag.scratch)
        {
            setError("could not obtain %zu bytes of global scratch for the assembly coefficient kernel", ldc * grid);
            return -3;
        }
        hipLaunchKernelGGL(kc, dim3(grid), dim3(applyThreads< P, NQ >()), 0, stream, ag, kern, cbuf);
    }
    else
        hipLaunchKernelGGL(kc, dim3(static_cast< unsigned >(a.elem_count)), dim3(applyThreads< P, NQ >()), ldc, stream, a, kern, cbuf);
    // the sum-factorised kernel unless it does not fit or l3k_tuning::assemble_dense asks for the dense MFMA product (cross-check)
    const l3k_tuning& tune  = tuneOf(a);
    const bool        dense = !S::feasible || (tune.assemble_dense && !a.K_tiled);
    if (a.K_tiled && (!S::feasible || !a.K))
    {
        setError("the tiled layout of the element matrices needs the sum-factorised assembly kernel (this shape has none)");
        return -1;
    }
    if constexpr (S::feasible)
        if (!dense)
        {
            // (stored modes: the partial 64-byte lines of K_e meet in one L2; streaming mode: the element's coefficient records are
            // fetched into one L2 instead of up to eight: +2.3 %, 516 -> 528 k matrices/s at order 6)
            constexpr int xcd_group = 1;
            auto      launch    = [&](auto ks, int NP, int threads, size_t lds_bytes) {
                const int64_t n_blocks = xcd_group ? ((a.elem_count + 7) / 8) * 8 * NP : a.elem_count * NP;
                if (n_blocks > int64_t(0x7fffffff))
                {
                    setError("assembly batch too large: %lld workgroups", (long long)n_blocks);
                    return -1;
                }
                if (n_blocks > 0)
                    hipLaunchKernelGGL(ks, dim3(static_cast< unsigned >(n_blocks)), dim3(threads), lds_bytes, stream, a, cbuf,
                                       int64_t(a.elem_begin_out), xcd_group);
                return 0;
            };
            // the diagonal and the off-diagonal blocks as two launches (a register allocation each); one launch over all pairs
            // where the workgroups of an element should meet in one L2 (stored row-major matrices) or on request
            const bool one_launch = a.K != nullptr && !tune.assemble_two_launches;
            int        rc         = 0;
            if (a.K_tiled == 2)
                rc = launch(assembleSumfactKernel< K, P, NQ, 2 >, U * U, S::threadsFor(true, 0), S::ldsFor(true, 0));
            else if (a.K_tiled)
                rc = launch(assembleSumfactKernel< K, P, NQ, 1 >, U * U, S::threadsFor(true, 0), S::ldsFor(true, 0));
            else if (one_launch)
                rc = launch(assembleSumfactKernel< K, P, NQ >, U * (U + 1) / 2, S::threadsFor(false, 0), S::ldsFor(false, 0));
            else
            {
                rc = launch(assembleSumfactKernel< K, P, NQ, 0, 2 >, U * (U - 1) / 2, S::threadsFor(false, 2), S::ldsFor(false, 2));
                if (rc == 0)
                    rc = launch(assembleSumfactKernel< K, P, NQ, 0, 1 >, U, S::threadsFor(false, 1), S::ldsFor(false, 1));
            }
            if (rc)
                return rc;
        }
    if (dense)
        hipLaunchKernelGGL(kg, dim3(C::NLT, static_cast< unsigned >(a.elem_count)), dim3(256), C::lds, stream, a, cbuf,
                           int64_t(a.elem_begin_out));
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess)
    {
        setError("assembly kernel launch failed: %s", hipGetErrorString(err));
        return -3;
    }
    return 0;
}
template < typename K, int P, int NQ >
constexpr size_t assembleWorkspaceDoublesPerElem()
{
    return size_t(coeffStride< K >()) * NQ * NQ * NQ;
}
} // namespace l3k::dev
#endif
