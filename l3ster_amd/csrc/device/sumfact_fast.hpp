// sumfact_fast.hpp -- register-resident, software-pipelined sum-factorised apply for hex elements (single column).
//
// Same mathematics as sumfact_apply.hpp (collocation-derivative form of evalLocalOperatorSumFact,
// algsys/SumFactorization.hpp:882-917, fused with gatherSumFact / scatterSumFact, algsys/MatrixFreeSystem.hpp:421-537)
// re-organised for the CDNA4 execution model, where this kernel is FP64-VALU bound (about 250 fp64 flop per dof,
// SURVEY.md D9) with the LDS (store path ~85 B/clk/CU) as the second limiter:
//
//  * one WAVE owns its element(s): a "team" of M*M lanes (M = max(n, nq)) per element, 64 / (M*M) elements per wave
//    (p = 6: 49 of 64 lanes; p = 4: 2 x 25; p = 3: 4 x 16).  A workgroup is a single wave, so there is NO workgroup
//    barrier anywhere: LDS instructions of one wave execute in order, and the 5-7 resident waves of a CU drift apart,
//    which overlaps one wave's LDS phases with another wave's FMA phases (with 256-thread workgroups in lockstep the two
//    pipes were each ~35 % busy and serialised: profiles/r01_pmc_fast_v2b_compute_only.txt);
//  * every lane owns ONE 1-D pencil of ALL fields in registers, so each sweep is register-only FMAs with the even-odd
//    decomposition (the reference's own flop cut, algsys/SumFactorization.hpp:88-343: 34 instead of 49 instructions per
//    7-point pencil); the LDS only carries the pencil re-orientations (two buffers per element), with the fields
//    interleaved in pairs so that every LDS access is 16 bytes wide;
//  * the global gather goes straight to registers one element ahead (node ids two ahead) and stays in flight behind the
//    whole compute phase of the current element; waves are persistent and walk the element batches;
//  * the result is staged once through LDS into dof order so that the scatter's wave-instructions cover contiguous
//    bytes; nodes touched by exactly one element (the element-internal nodes of the reference's numbering,
//    mesh/LocalMeshView.hpp:425-458) use plain read-modify-write, only shared nodes use f64 atomics.
#ifndef L3K_DEVICE_SUMFACT_FAST_HPP
#define L3K_DEVICE_SUMFACT_FAST_HPP

#include "sumfact_apply.hpp"

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>
#include <type_traits>

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
        double A = 0.;
#pragma unroll
        for (int r = 0; r < RI; ++r)
            A += e[r] * We[r * RO + q];
        if constexpr (ACC)
        {
            // lo = A + B as one chain on top of A, the mirror output from lo and A: 10 instead of 11 instructions per pair
            double lo = A;
#pragma unroll
            for (int r = 0; r < HI; ++r)
                lo += o[r] * Wo[r * RO + q];
            out[q] += lo;
            out[NOUT - 1 - q] += ANTI ? lo - 2. * A : 2. * A - lo;
        }
        else
        {
            // lo = A + B with the odd part accumulated onto A (no separate B, no add); hi = A - B = 2A - lo: one
            // instruction less per output pair than forming A, B, A + B, A - B
            double lo = A;
#pragma unroll
            for (int r = 0; r < HI; ++r)
                lo += o[r] * Wo[r * RO + q];
            out[q]            = lo;
            out[NOUT - 1 - q] = ANTI ? lo - 2. * A : 2. * A - lo;
        }
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

// the second field of a field group: swept where the group is fully used, zeros where it is the padding of an odd field count
// (`live` is a compile-time constant after unrolling: no branch in the code)
template < int NIN, int NOUT, bool ANTI, bool ACC >
__device__ __forceinline__ void sweepEOSecond(bool live, const double (&in)[NIN], double (&out)[NOUT], const double* __restrict__ eo)
{
    if (live)
        sweepEO< NIN, NOUT, ANTI, ACC >(in, out, eo);
    else if constexpr (!ACC)
    {
#pragma unroll
        for (int q = 0; q < NOUT; ++q)
            out[q] = 0.;
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

// A zero the compiler cannot see through, in an SGPR.  Table pointers are offset by a fresh one in every stage so that
// the coefficient loads (s_load from the kernel-argument segment) are re-issued next to their use instead of being
// hoisted out of the element loop all at once, which would need ~280 SGPRs and spill them to VGPR lanes.
__device__ __forceinline__ int opaqueZero()
{
    int z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z) : : "memory"); // ("memory": the stage's LDS reads stay behind the table loads)
    return z;
}
// A copy of a per-lane value the compiler cannot see through (keeps values derived from it out of loop-invariant hoisting).
__device__ __forceinline__ int opaqueCopy(int x)
{
    int y;
    asm volatile("v_mov_b32 %0, %1" : "=v"(y) : "v"(x));
    return y;
}
__device__ __forceinline__ double opaqueCopy(double x)
{
    double y;
    asm volatile("v_mov_b64 %0, %1" : "=v"(y) : "v"(x));
    return y;
}
// Copies a 1-D table out of the kernel-argument segment (scalar loads) and pins the loads in front of everything that
// follows in program order, so that their latency overlaps with the stage's LDS reads instead of following it.
template < int N >
__device__ __forceinline__ void loadTable(double (&t)[N], const double* src)
{
#pragma unroll
    for (int i = 0; i < N; ++i)
        t[i] = src[i];
    __builtin_amdgcn_sched_barrier(0);
}
// Stage separator inside a wave: LDS instructions of one wave execute in issue order, so no s_barrier is needed; this
// only stops the COMPILER from moving LDS accesses across the stage boundary.
__device__ __forceinline__ void stageFence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Stage timeline of one wave (tools/kbench.py --stamps; ablation builds only): shader clock at the stage boundaries of
// workgroup 0, 16 slots per element iteration
#ifdef L3K_ABLATION
#define L3K_STAMP(k)                                                                                                   \
    do                                                                                                                 \
    {                                                                                                                  \
        if (a.stamps != nullptr && blockIdx.x == 0 && stamp_it < 256)                                                  \
        {                                                                                                              \
            const long long t_ = __builtin_readcyclecounter();                                                         \
            if (lane == 0)                                                                                             \
                a.stamps[stamp_it * 16 + (k)] = t_;                                                                    \
        }                                                                                                              \
    } while (0)
#else
#define L3K_STAMP(k)                                                                                                   \
    do                                                                                                                 \
    {                                                                                                                  \
    } while (0)
#endif

#ifndef L3K_FAST_MIN_WAVES
#define L3K_FAST_MIN_WAVES 2
#endif
// A kernel that never reads in.field_ders may say so (static constexpr bool field_derivatives = false): the derivative
// sweeps of the external fields and their half of the second LDS buffer are then skipped
template < typename K >
constexpr bool kernelUsesFieldDers()
{
    if constexpr (requires { K::field_derivatives; })
        return K::field_derivatives;
    else
        return true;
}

// bit u of the result: dof u of the node is a Dirichlet dof (mask bytes [node * U, node * U + U), one load where U allows it)
template < int U >
__device__ __forceinline__ uint32_t dirichletBits(const uint8_t* __restrict__ mask, int64_t node)
{
    if constexpr (U == 4)
    {
        const uint32_t w = *reinterpret_cast< const uint32_t* >(mask + node * 4);
        return ((w & 0xffu) ? 1u : 0u) | ((w & 0xff00u) ? 2u : 0u) | ((w & 0xff0000u) ? 4u : 0u) | ((w & 0xff000000u) ? 8u : 0u);
    }
    else if constexpr (U == 2)
    {
        const uint32_t w = *reinterpret_cast< const uint16_t* >(mask + node * 2);
        return ((w & 0xffu) ? 1u : 0u) | ((w & 0xff00u) ? 2u : 0u);
    }
    else
    {
        uint32_t dm = 0;
#pragma unroll
        for (int u = 0; u < U; ++u)
            dm |= uint32_t(mask[node * U + u] != 0) << u;
        return dm;
    }
}

// Waves per SIMD the single-wave kernel is compiled for (its register budget: 256 VGPRs at 2, 168 at 3).  A 4-unknown, 7-equation
// kernel needs 252 registers at order 6 (three waves per SIMD spill); a SCALAR kernel holds a quarter of the pencil
// state (order 6: 144 registers) and runs three waves per SIMD without spills.  A functor may say so itself
// (static constexpr int waves_per_simd = 2 or 3).
template < typename K, int NQ_, int NFD >
constexpr int kernelWavesPerSimd()
{
    if constexpr (requires { K::waves_per_simd; })
        return K::waves_per_simd;
    else
        return K::params.n_unknowns == 1 && NQ_ * NFD <= 36 ? 3 : L3K_FAST_MIN_WAVES;
}

template < typename K, int P, int NQ >
struct FastCfg
{
    static constexpr int N1 = P + 1, M = cmax(N1, NQ), TEAM = M * M, NN = N1 * N1 * N1;
    static constexpr int U = K::params.n_unknowns, F = K::params.n_fields, NF = U + F;
    static constexpr int NG = (NF + 1) / 2; // field groups of 2 (16-byte LDS accesses); last may be half used
    static constexpr int UG = (U + 1) / 2;
    static constexpr int EW = 64 / TEAM > 0 ? 64 / TEAM : 1; // elements per wave
    // LDS array strides: point (c, b, a) lives at b*PS + a*M + c (in 16-byte units within a field group)
#ifdef L3K_FAST_PS
    static constexpr int PS = M == 7 ? L3K_FAST_PS : M * M, OS = PS * M;
#else
    // plane stride: M * M, padded where that removes bank conflicts (16-byte units, 16 consecutive lanes per LDS pass: the unit
    // index modulo 16 must be distinct among them).  M = 8: stride 64 = 0 (mod 16) put the eight x-pencils of a lane row on ONE
    // bank group (8-way conflicts on the x-oriented accesses: 352 -> 128 LDS passes per element and field group with 65); M = 6:
    // 90 -> 66 with 41; M = 4: 144 -> 64 with 19.  M = 7 (49 = 1 mod 16: one orientation 2.5-way) and M = 5 have no better stride,
    // and at M = 7 the LDS has no room (7 waves per CU).  Model: tools/lds_bank_model.py.  (Kernels with many fields keep the plain
    // stride where the padded buffers would pass the 64 KB of a workgroup: NS3D, 14 fields, at order 2 with nq = 4.)
    static constexpr int PS_PAD = M == 8 ? 65 : (M == 6 ? 41 : (M == 4 ? 19 : M * M));
    static constexpr int ldsWith(int ps) // (the formula of `lds` below)
    {
        const int ng = (NF + 1) / 2, dg = kernelUsesFieldDers< K >() ? ng : (U + 1) / 2, os = ps * M;
        const bool alias = dg < ng && ng - dg >= dg;
        return 8 * EW * ((alias ? 2 * ng * os : 2 * ng * os + 2 * dg * os) + 26) + 16 * N1 * N1;
    }
    static constexpr int PS = ldsWith(PS_PAD) <= 64 * 1024 ? PS_PAD : M * M, OS = PS * M;
#endif
    // per team: bufA | bufB (NG groups of OS double2 each) | vertices
    // groups whose reference derivatives reach the quadrature-point stage: all of them, or the unknowns' only
    static constexpr int    DG       = kernelUsesFieldDers< K >() ? NG : UG;
    static constexpr int    DF       = kernelUsesFieldDers< K >() ? NF : U; // fields with derivatives
    static constexpr int    BUF_D    = 2 * NG * OS;                         // buffer A (doubles)
    static constexpr int    BUFB_D   = 2 * DG * OS;                         // buffer B: derivative groups only
    // Fields that enter by value only (DG < NG: the kernel declared field_derivatives = false) are dead in buffer A once the
    // x interpolation has taken them into registers -- buffer B (first written by the eta-derivative stage, DG groups) then lives in
    // buffer A's groups [DG, 2 DG): no LDS of its own.  Config 5's kernel (U = 4, F = 3) at order 4: 16.4 instead of 24.4 KB per
    // wave, 8 instead of 6 waves per CU; the scalar advection kernel (U = 1, F = 3) at order 6: 11.9 instead of 17.4 KB, 12 waves
    static constexpr bool   ALIAS_B  = DG < NG && NG - DG >= DG;
    static constexpr int    OFF_B    = ALIAS_B ? 2 * DG * OS : BUF_D;       // buffer B's offset in the team's block (doubles)
    static constexpr int    OFF_V    = ALIAS_B ? BUF_D : BUF_D + BUFB_D;    // vertices, energy accumulator
    static constexpr int    TEAM_D   = OFF_V + 24 + 2; // buffers, 8 vertices, energy accumulator (+ pad)
    static constexpr int    SLOT_B   = 16 * N1 * N1; // scatter-slot table of the mesh: [N1*N1 lanes][8] uint16, one copy per wave
    static constexpr size_t lds      = sizeof(double) * size_t(EW) * TEAM_D + SLOT_B;
#ifndef L3K_FAST_PS
    static_assert(lds == size_t(ldsWith(PS)));
#endif
    static constexpr int    SG       = (64 / EW) / U * U; // lanes that scatter one element (the team's lanes + helpers): a multiple of U
    static constexpr int    NSH      = NN - (N1 - 2) * (N1 - 2) * (N1 - 2); // nodes on the element's shell
    // (any number of unknowns: an odd U leaves the second half of its last field group to the first external field, or unused)
    static constexpr bool   feasible = TEAM <= 64 && lds <= 64 * 1024 && NF * cmax(N1, NQ) <= 64 && NN * U <= BUFB_D && N1 <= 8 && SG >= U;
    // resident single-wave workgroups per CU by LDS capacity; with at most one per SIMD the wave may use all 512
    // registers (VGPR + AGPR) of its SIMD lane instead of spilling to scratch (order 7: 33 KB of LDS per wave)
    static constexpr int waves_by_lds = int((160 * 1024) / (lds > 0 ? lds : 1));
    // (three waves per SIMD for the small-LDS shapes was tried: 168 registers, spills, order 4 5.1 -> 7.5 ns per element)
    static constexpr int min_waves    = waves_by_lds <= 4 ? 1 : (waves_by_lds <= 8 ? cmin(2, kernelWavesPerSimd< K, NQ, NF + DF >()) : kernelWavesPerSimd< K, NQ, NF + DF >());
    // (the multi-column variant carries the column loop's state on top: two waves per SIMD at most)
    static constexpr int wavesPerSimd(bool multi) { return multi ? cmin(2, min_waves) : min_waves; }
    // field groups whose LDS reads are issued together in the LDS -> LDS stages (2 * N1 doubles of registers per extra group).
    // (order 4 at three waves per SIMD, 168 registers, was tried again with read_block = 1: still 34 spilled registers.)
    static constexpr int read_block   = NG;
    // the multi-column variant (all n_rhs columns per element pass) is used where the kernel has registers to spare: with
    // several elements per wave (orders <= 4: +2 % over column-by-column launches at order 4, 3 columns).  With one element
    // per wave the variant spills (order 6: 45 registers) and loses 12-17 % against column-by-column launches of the
    // single-column kernel (tools/bench_multicol.py), which those shapes therefore keep.
    static constexpr bool multi_column = EW > 1;
};

// SPLIT: ghost rows live in buffers of their own (a.xg / a.yg, the reference's import / export buffers): every node needs
// an owned-or-ghost select.  SPLIT = false (no ghost buffers in this launch -- one rank, or interior elements -- or ghost
// rows directly behind the owned rows): one base pointer, ~150 instructions per element less.
// ENERGY: the kernel also accumulates x^T A x of its elements into *a.energy (for <p, A p> of the PCG: saves the separate
// dot-product pass over two vectors): per element x_e . y_e, formed where the result leaves the registers (I^T z stage).
// AFFINE: every element of the launch is a parallelepiped (its tri-linear map is affine: l3k_mesh_create checks the vertices):
// one Jacobian per element, inverted once per element instead of once per quadrature point.
// MULTI: a.n_cols columns per element pass (the reference applies all n_rhs columns in one sweep over the elements,
// MatrixFreeSystem.hpp:678-688): node ids, vertices, flags and the work ticket are fetched once per element, the stages run
// once per column (the geometry is recomputed: no registers to keep 343 Jacobians).  A variant of its own, so that the
// single-column kernel's code and register allocation stay what they are.
// STRIDED: the kernel's unknowns are a subset of the node's dofs (dof = node * dofs_per_node + field_inds[u]: detail::getDofs,
// algsys/MatrixFreeSystem.hpp:298-311): 8-byte gather and scatter accesses, every node through the atomic path (the rows of such
// a vector hold other kernels' dofs, which the pre-scaling pass must not skip).  Plain applies only (no fused energy, one column).
// RHS: the right-hand side with Dirichlet lifting instead of the apply (precomputeOperatorDiagonalAndRhs' rhs term,
// algsys/EvaluateLocalOperator.hpp:172-208: rhs += B^T W (f - B g) with g = the Dirichlet values on the Dirichlet dofs and 0
// elsewhere): the gather reads a.dirichlet_vals where the mask is set, the quadrature stage runs in its RHS form, the scatter adds
// into every row (Dirichlet rows are overwritten by the finalize pass) through the atomic path, alpha = 1.  One rhs column.
template < typename K, int P, int NQ, bool SPLIT, bool ENERGY, bool AFFINE = false, bool MULTI = false, bool STRIDED = false, bool RHS = false >
__global__ __launch_bounds__(64, (FastCfg< K, P, NQ >::wavesPerSimd(MULTI))) void sumfactFastKernel(const ElemArgs a, const K kern, int64_t n_batches,
                                                        int xcd_chunk, const FastTables< P + 1, NQ > tab)
{
    using Cfg = FastCfg< K, P, NQ >;
    constexpr int N1 = Cfg::N1, M = Cfg::M, PS = Cfg::PS, OS = Cfg::OS, TEAM = Cfg::TEAM, EW = Cfg::EW;
    constexpr int U = Cfg::U, F = Cfg::F, NF = Cfg::NF, NG = Cfg::NG, UG = Cfg::UG, NN = N1 * N1 * N1;
    constexpr int DG = Cfg::DG, DF = Cfg::DF; // groups / fields whose derivatives are formed
    static_assert(!STRIDED || (!ENERGY && !AFFINE && !MULTI));
    static_assert(!RHS || (!ENERGY && !AFFINE && !MULTI && !STRIDED));
    [[maybe_unused]] const int dpn = a.dofs_per_node; // (STRIDED)
    // bit u: dof u of the kernel at `node` is a Dirichlet dof
    auto dirBits = [&](int64_t node) -> uint32_t {
        if constexpr (STRIDED)
        {
            uint32_t dm = 0;
#pragma unroll
            for (int u = 0; u < U; ++u)
                dm |= uint32_t(a.dirichlet[node * dpn + a.field_inds[u]] != 0) << u;
            return dm;
        }
        else
            return dirichletBits< U >(a.dirichlet, node);
    };
    // ENERGY: x^T A x either inside the quadrature stage (sum_q wgt |B x|^2: an accumulator where no register is free) or as
    // x_e . y_e where the result leaves the registers (a second fetch of the element's x rows).  Measured per shape
#if defined(L3K_FLAGGED_SCATTER)
    constexpr bool ENERGY_AT_END = false; // (the A/B form does not zero the staged Dirichlet dofs, which x_e . y_e relies on)
#elif defined(L3K_ENERGY_AT_END)
    constexpr bool ENERGY_AT_END = L3K_ENERGY_AT_END != 0;
#else
    constexpr bool ENERGY_AT_END = EW == 1;
#endif
    constexpr int HN = (N1 + 1) / 2, HQ = (NQ + 1) / 2;
    constexpr int GB = Cfg::read_block; // field groups whose LDS reads are batched (register cost: 2 * N1 doubles per extra group)
    // with fewer derivative groups than groups, buffer B is too small for the y / x interpolation of all groups: those two
    // sweeps then run in place in buffer A (a lane reads its whole pencil before it writes it; pencils are disjoint)
    constexpr bool INPLACE = DG < NG;
    // LDS position of point (c, b, a_) = (x, y, z index), in 16-byte units.  With the lanes running over two of the three
    // indices (first one fastest) the unit index modulo 16 must be distinct within 16 consecutive lanes: strides (1, 7) and
    // (49 = 1, 7) are, (1, 49 = 1) is not -- one of the three pencil orientations is always 3-way bank-conflicted.  y
    // gets the plane stride so that it is the z-oriented accesses (84 per element) and not the y-oriented ones (112).
    auto          at = [](int c, int b, int a_) { return b * PS + a_ * M + c; };

    extern __shared__ double lds[];
    const int                lane = threadIdx.x;
    const int                team = lane / TEAM, l = lane - team * TEAM;
    // single-wave workgroup without barriers.  Lanes beyond the teams' pencils do no sweeps but help in the scatter:
    // the 64 / EW lanes [steam * SG, (steam + 1) * SG) scatter the element of team `steam`
    const bool               worker = team < EW && l < cmax(N1 * N1, cmax(N1 * NQ, NQ * NQ));
    constexpr int            SG     = Cfg::SG;
    double* const            base = lds + size_t(worker ? team : 0) * Cfg::TEAM_D;
    double2* const           bufA = reinterpret_cast< double2* >(base);
    double2* const           bufB = reinterpret_cast< double2* >(base + Cfg::OFF_B);
    double* const            vs   = base + Cfg::OFF_V; // [8][3]

    const double* const eoI  = tab.eoI;
    const double* const eoC  = tab.eoC;
    const double* const eoIt = tab.eoIt;
    const double* const eoCt = tab.eoCt;
    const double* const qw   = tab.qw;
    const double* const qp   = tab.qx;

    // pencil coordinates of this lane in the different stage grids (first index fastest)
    const int  i1 = l % N1, j1 = l / N1; // (N1 x N1) grid: gather / z-sweep / final z-sweep
    const bool on_nn = worker && (N1 == NQ ? true : l < N1 * N1);
    const int  iq = l % N1, kq = l / N1; // (N1 x NQ) grid: y-sweeps at node-x
    const int  qa = l % NQ, qb = l / NQ; // (NQ x NQ) grid

    // field-group accessors: group g holds fields 2g, 2g+1 (the second may be unused padding)
    auto ldg = [&](const double2* buf, int g, int idx) { return buf[g * OS + idx]; };
    auto stg = [&](double2* buf, int g, int idx, double x0, double x1) { buf[g * OS + idx] = make_double2(x0, x1); };

    // XCD-aware walk: workgroups are dealt round-robin to the 8 XCDs (workgroups b and b + 8 share an XCD and its L2).
    // The c-th group of workgroups takes the c-th contiguous eighth of the (brick-ordered) element batches and sweeps it
    // together, so elements that share node rows of x are resident in the same L2 at the same time.  32-bit loop state:
    // the kernel is at its register limit and 64-bit counters were spilled.
    const int nb     = static_cast< int >(n_batches);
    const int by_xcd = xcd_chunk > 0;
    const int stride = by_xcd ? int(gridDim.x) >> 3 : int(gridDim.x);
    const int first  = by_xcd ? int(blockIdx.x & 7) * xcd_chunk : 0;
    const int last   = by_xcd ? (first + xcd_chunk < nb ? first + xcd_chunk : nb) : nb;
    // Dynamic distribution (a.work_counters != nullptr): every wave draws its next batch from a counter of its XCD, one
    // returning atomic per element, issued at the top of the element and consumed only after the quadrature stage (its
    // latency, behind the previous element's atomics in the in-order memory queue, is never waited for).  With a static
    // deal the waves of one launch finished between 56 % and 100 % of the kernel's duration (a wave alone on its SIMD is
    // faster, boundary elements are slower, ...): profiles/r01_kbench_stage_timeline.log.  Batch = base + ticket * step:
    // the XCD's contiguous chunk (by_xcd), or batches congruent to the XCD index modulo 8 (the static deal's mapping).
    // A wave whose XCD has run dry continues with the next XCD's counter (the chunks that hold the Dirichlet faces take
    // ~7 % longer): at most 7 switches per wave, each with one exposed atomic round trip, at the very end of the launch.
    const bool dyn     = a.work_counters != nullptr;
    const bool sharded = by_xcd || (gridDim.x & 7u) == 0;
    int        victim  = sharded ? int(blockIdx.x & 7u) : 0, switches = 0;
    auto       vBase   = [&](int v) { return by_xcd ? v * xcd_chunk : v; };
    auto       vLimit  = [&](int v) { return by_xcd ? (v * xcd_chunk + xcd_chunk < nb ? v * xcd_chunk + xcd_chunk : nb) : nb; };
    const int  dyn_step = by_xcd || !sharded ? 1 : 8;
    auto       drawTicket = [&]() -> uint32_t {
        uint32_t t = 0;
        if (dyn && lane == 0)
            t = __hip_atomic_fetch_add(a.work_counters + 32 * victim, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // 128 B apart
        return t;
    };
    // batch of a ticket drawn from the current victim, or the end marker nb when every counter is exhausted
    auto ticketBatch = [&](uint32_t t) {
        int b = vBase(victim) + int(__builtin_amdgcn_readfirstlane(t)) * dyn_step;
        while (b >= vLimit(victim))
        {
            if (!by_xcd || ++switches > 7) // (interleaved batches: every XCD has the same mix of elements, nothing to steal)
                return nb;
            victim = (victim + 1) & 7;
            b      = vBase(victim) + int(__builtin_amdgcn_readfirstlane(drawTicket())) * dyn_step;
        }
        return b;
    };
    const int lim   = dyn ? nb : last; // loop bound: dynamic batches are valid below nb
    int       batch = dyn ? ticketBatch(drawTicket()) : first + (by_xcd ? int(blockIdx.x >> 3) : int(blockIdx.x));
    // ---- software pipeline state: node ids two batches ahead, x values one batch ahead
    uint32_t      ids_cur[N1], ids_nxt[N1];
    double        xn[N1][U];
    double        fn[N1][F > 0 ? F : 1];
    uint32_t      dm_nxt[N1];
    const int64_t n_owned_nodes = a.n_owned_dofs / (STRIDED ? a.dofs_per_node : U);
    auto          elemOf = [&](int b) { return a.elem_begin + int64_t(b) * EW + team; };
    auto          valid  = [&](int b) { return (b < lim) & on_nn & ((int64_t(b) * EW + team) < a.elem_count); };
    const bool have_flags = a.elem_flags != nullptr;
    auto       loadIds    = [&](int batch, uint32_t (&ids)[N1], uint32_t& flag) {
        // "element touches a Dirichlet dof", fetched with the ids, one element ahead of its use: bit 0 for the element this
        // lane gathers (its team's), bit 1 for the element it scatters (the one of its scatter group steam)
        // (per-lane addresses rebuilt from an opaque copy of the lane id: see the scatter)
        const int lane_o = opaqueCopy(lane), steam = lane_o / SG;
        uint32_t  fg = 0, fs = 0;
        if (have_flags && batch < lim)
        {
            if (steam < EW && (int64_t(batch) * EW + steam) < a.elem_count)
                fs = a.elem_flags[a.elem_begin + int64_t(batch) * EW + steam];
            if constexpr (EW == 1)
                fg = fs;
            else if (worker && (int64_t(batch) * EW + team) < a.elem_count)
                fg = a.elem_flags[elemOf(batch)];
        }
        // bit 0 / 1: the gathered / the scattered element touches a Dirichlet dof (bit 1 of the mesh's flags marks affine
        // elements: used at launch time to select the AFFINE kernel variant for all-affine meshes)
        flag = ((fg & 1u) ? 1u : 0u) | ((fs & 1u) ? 2u : 0u);
        if (valid(batch))
        {
            const uint32_t* en = a.elem_nodes + elemOf(batch) * NN + (lane_o - team * TEAM); // + i1 + N1 * j1
#pragma unroll
            for (int k = 0; k < N1; ++k)
                ids[k] = en[k * N1 * N1];
        }
    };
    [[maybe_unused]] int col = 0; // MULTI: the column of this pass
    auto loadX = [&](bool mine, const uint32_t (&ids)[N1], bool flagged) {
        if (!mine)
            return;
        const double* const ax  = MULTI ? a.x + size_t(col) * a.ldx : a.x;
        const double* const axg = MULTI && SPLIT ? a.xg + size_t(col) * a.ldxg : a.xg;
        // node-interleaved dofs with the kernel's unknowns = all dofs of a node (the launcher sends every other layout
        // to the generic kernel): 16-byte vector loads, adjacent lanes read adjacent nodes
        // (dofs/NodeToDofMap.hpp:250-264 layout)
#pragma unroll
        for (int k = 0; k < N1; ++k)
        {
            const int64_t node = ids[k];
            const double* p    = STRIDED ? (!SPLIT || node < n_owned_nodes ? ax + node * dpn : axg + (node - n_owned_nodes) * dpn)
                                         : (!SPLIT || node < n_owned_nodes ? ax + node * U : axg + (node - n_owned_nodes) * U);
            if constexpr (RHS) // g: the Dirichlet values on the Dirichlet dofs (one array over all local dofs), 0 elsewhere
            {
                const uint32_t dm = flagged ? dirBits(node) : 0u;
#pragma unroll
                for (int u = 0; u < U; ++u)
                    xn[k][u] = (dm >> u) & 1u ? a.dirichlet_vals[node * U + u] : 0.;
            }
            else if constexpr (STRIDED) // a subset of the node's dofs: one 8-byte load per unknown
            {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    xn[k][u] = p[a.field_inds[u]];
            }
            else if constexpr (U % 2 == 0)
            {
#pragma unroll
                for (int hh = 0; hh < U / 2; ++hh)
                {
                    const double2 t = (L3K_DBG(a) & 2) ? make_double2(1e-9 * double(node), 1e-9)
                                                  : *reinterpret_cast< const double2* >(p + 2 * hh);
                    xn[k][2 * hh]     = t.x;
                    xn[k][2 * hh + 1] = t.y;
                }
            }
            else // an odd number of unknowns: a node's row is not 16-byte aligned -- 8-byte loads (adjacent lanes, adjacent nodes)
            {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    xn[k][u] = (L3K_DBG(a) & 2) ? 1e-9 * double(node) : p[u];
            }
            dm_nxt[k] = !RHS && flagged ? dirBits(node) : 0u;
#pragma unroll
            for (int f = 0; f < F; ++f)
                fn[k][f] = a.fields[node + f * a.ldf];
        }
    };

    // lane-dependent table entries are vector loads from the kernel-argument segment: read them once
    const double eta_l = qp[qa < NQ ? qa : 0], zeta_l = qp[qb < NQ ? qb : 0];
    const double wyz_l = qw[qa < NQ ? qa : 0] * qw[qb < NQ ? qb : 0];

    // the mesh's scatter-slot table: one 16-byte row (8 x uint16) per gathering lane, kept in LDS behind the teams' buffers
    uint4* const slotRows = reinterpret_cast< uint4* >(lds + size_t(EW) * Cfg::TEAM_D);
    if (lane < N1 * N1)
        slotRows[lane] = reinterpret_cast< const uint4* >(a.slot_tab)[lane];
    if constexpr (ENERGY)
        if (worker && l == 0)
            vs[24] = 0.;
    stageFence();
    uint32_t flag_cur, flag_nxt = 0;
    loadIds(batch, ids_cur, flag_cur);

    [[maybe_unused]] int stamp_it = 0;
#ifdef L3K_ABLATION
    if (a.stamps != nullptr && blockIdx.x < 4096 && lane == 0)
        a.stamps[256 * 16 + 2 * blockIdx.x] = __builtin_readcyclecounter();
#endif
    while (batch < lim)
    {
        const uint32_t ticket = drawTicket(); // the batch after this one
        L3K_STAMP(0);
        // this team has an element in this batch (always, with one element per wave: batch < last <= elem_count)
        const bool act = (int64_t(batch) * EW + team) < a.elem_count;
        // one mask per pencil grid; with N1 == NQ they are the same value, so that the stages form one masked region
        const bool w_all = worker & act;
        const bool w_nn = N1 == NQ ? w_all : (w_all & (l < N1 * N1)), w_nq = N1 == NQ ? w_all : (w_all & (l < N1 * NQ)),
                   w_qq = N1 == NQ ? w_all : (w_all & (l < NQ * NQ));
        int        batch_next = 0;
        col                   = 0;
        do // (MULTI: once per column; otherwise a single pass and no loop in the code)
        {
        // all gather loads of this element are issued back to back (one exposed latency, hidden by the other resident
        // waves); only the next element's node ids are prefetched: holding the next x values in registers across the
        // compute phase made the compiler spill them and wait on every load (profiles/r01 notes in DESIGN.md)
        // one masked region for everything up to the scatter: the lanes without a pencil (helpers of the scatter) skip it
        if (w_all)
        {
        loadX(w_nn, ids_cur, RHS ? (a.dirichlet_vals != nullptr && ((flag_cur & 1u) != 0 || !have_flags)) : (flag_cur & 1u) != 0);
        // ---- take over the prefetched data (gatherSumFact: Dirichlet dofs read as 0, MatrixFreeSystem.hpp:441-466)
        double u0[N1][2 * NG];
        if (w_nn)
        {
#pragma unroll
            for (int k = 0; k < N1; ++k)
            {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    u0[k][u] = xn[k][u];
                if (!RHS && (flag_cur & 1u) != 0) // only elements touching a Dirichlet dof pay for the masking
                {
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        u0[k][u] = (dm_nxt[k] >> u) & 1u ? 0. : u0[k][u];
                }
#pragma unroll
                for (int f = 0; f < F; ++f)
                    u0[k][U + f] = fn[k][f];
                if constexpr (NF % 2)
                    u0[k][NF] = 0.;
            }
        }
        // vertices: requested here (one coordinate per lane), used at the quadrature stage: the load's latency is never
        // waited for.  Teams of fewer than 24 lanes (orders <= 3) store them to the team's LDS block right away
        constexpr bool VREG = TEAM >= 24;
        double         vreg = 0.;
        if constexpr (VREG)
        {
            if (l < 24)
                vreg = a.elem_verts[elemOf(batch) * 24 + opaqueCopy(l)];
        }
        else
        {
            for (int t = opaqueCopy(l); t < 24; t += TEAM)
                vs[t] = a.elem_verts[elemOf(batch) * 24 + t];
        }

        // ---- S1: z interpolation in registers; write (c=i, b=j, a=qz) into bufA
        if (w_nn)
        {
            double tI[2 * HN * HQ];
            loadTable(tI, eoI + opaqueZero());
#pragma unroll
            for (int g = 0; g < NG; ++g)
            {
                double in0[N1], in1[N1], o0[NQ], o1[NQ];
#pragma unroll
                for (int k = 0; k < N1; ++k)
                {
                    in0[k] = u0[k][2 * g];
                    in1[k] = u0[k][2 * g + 1];
                }
                sweepEO< N1, NQ, false, false >(in0, o0, tI);
                sweepEOSecond< N1, NQ, false, false >(2 * g + 1 < NF, in1, o1, tI);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    stg(bufA, g, at(i1, j1, q), o0[q], o1[q]);
            }
        }
        stageFence();
        L3K_STAMP(1);
        // ---- S2: y interpolation, lane (i, qz): bufA (c=i, b=j, a=qz) -> bufB (c=i, b=qy, a=qz)
        // (here and in the later LDS -> LDS stages the reads of all field groups are issued before the first group's
        // sweeps: one exposed LDS round trip per stage instead of one per group)
        if (w_nq)
        {
            double tI[2 * HN * HQ];
            loadTable(tI, eoI + opaqueZero());
            // (GB field groups per block: their LDS reads are issued before the first group's sweeps)
#pragma unroll
            for (int gb = 0; gb < NG; gb += GB)
            {
            double in0[GB][N1], in1[GB][N1];
#pragma unroll
            for (int g = gb; g < gb + GB && g < NG; ++g)
#pragma unroll
                for (int j = 0; j < N1; ++j)
                {
                    const double2 t = ldg(bufA, g, at(iq, j, kq));
                    in0[g - gb][j] = t.x;
                    in1[g - gb][j] = t.y;
                }
#pragma unroll
            for (int g = gb; g < gb + GB && g < NG; ++g)
            {
                double o0[NQ], o1[NQ];
                sweepEO< N1, NQ, false, false >(in0[g - gb], o0, tI);
                sweepEOSecond< N1, NQ, false, false >(2 * g + 1 < NF, in1[g - gb], o1, tI);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    stg(INPLACE ? bufA : bufB, g, at(iq, q, kq), o0[q], o1[q]);
            }
            }
        }
        stageFence();
        L3K_STAMP(2);
        // ---- S3/S4: x interpolation + xi-derivative, lane (qy, qz) = (qa, qb); values to bufA (c=qx, b=qy, a=qz)
        double v[NQ][2 * NG], dxi[NQ][2 * NG];
        if (w_qq)
        {
            double tI[2 * HN * HQ];
            loadTable(tI, eoI + opaqueZero());
            // (GB field groups per block: their LDS reads are issued before the first group's sweeps)
#pragma unroll
            for (int gb = 0; gb < NG; gb += GB)
            {
            double in0[GB][N1], in1[GB][N1];
#pragma unroll
            for (int g = gb; g < gb + GB && g < NG; ++g)
#pragma unroll
                for (int i = 0; i < N1; ++i)
                {
                    const double2 t = ldg(INPLACE ? bufA : bufB, g, at(i, qa, qb));
                    in0[g - gb][i] = t.x;
                    in1[g - gb][i] = t.y;
                }
#pragma unroll
            for (int g = gb; g < gb + GB && g < NG; ++g)
            {
                double o0[NQ], o1[NQ];
                sweepEO< N1, NQ, false, false >(in0[g - gb], o0, tI);
                sweepEOSecond< N1, NQ, false, false >(2 * g + 1 < NF, in1[g - gb], o1, tI);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                {
                    v[q][2 * g]     = o0[q];
                    v[q][2 * g + 1] = o1[q];
                    if (g < DG) // (only the groups whose eta / zeta derivatives are formed go back to LDS)
                        stg(bufA, g, at(q, qa, qb), o0[q], o1[q]);
                }
            }
            }
            double tC[2 * HQ * HQ];
            loadTable(tC, eoC + opaqueZero());
#pragma unroll
            for (int o = 0; o < DF; ++o)
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
        stageFence();
        L3K_STAMP(3);
        // ---- S5: eta-derivative, lane (qx, qz) = (qa, qb): y-pencils of bufA -> bufB
        if (w_qq)
        {
            double tC[2 * HQ * HQ];
            loadTable(tC, eoC + opaqueZero());
            // (GB field groups per block: their LDS reads are issued before the first group's sweeps)
#pragma unroll
            for (int gb = 0; gb < DG; gb += GB)
            {
            double in0[GB][NQ], in1[GB][NQ];
#pragma unroll
            for (int g = gb; g < gb + GB && g < DG; ++g)
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                {
                    const double2 t = ldg(bufA, g, at(qa, q, qb));
                    in0[g - gb][q] = t.x;
                    in1[g - gb][q] = t.y;
                }
#pragma unroll
            for (int g = gb; g < gb + GB && g < DG; ++g)
            {
                double o0[NQ], o1[NQ];
                sweepEO< NQ, NQ, true, false >(in0[g - gb], o0, tC);
                sweepEOSecond< NQ, NQ, true, false >(2 * g + 1 < DF, in1[g - gb], o1, tC);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    stg(bufB, g, at(qa, q, qb), o0[q], o1[q]);
            }
            }
        }
        stageFence();
        L3K_STAMP(4);
        // ---- S6: zeta-derivative in place in bufA, lane (qx, qy) = (qa, qb)
        if (w_qq)
        {
            double tC[2 * HQ * HQ];
            loadTable(tC, eoC + opaqueZero());
            // (GB field groups per block: their LDS reads are issued before the first group's sweeps)
#pragma unroll
            for (int gb = 0; gb < DG; gb += GB)
            {
            double in0[GB][NQ], in1[GB][NQ];
#pragma unroll
            for (int g = gb; g < gb + GB && g < DG; ++g)
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                {
                    const double2 t = ldg(bufA, g, at(qa, qb, q));
                    in0[g - gb][q] = t.x;
                    in1[g - gb][q] = t.y;
                }
#pragma unroll
            for (int g = gb; g < gb + GB && g < DG; ++g)
            {
                double o0[NQ], o1[NQ];
                sweepEO< NQ, NQ, true, false >(in0[g - gb], o0, tC);
                sweepEOSecond< NQ, NQ, true, false >(2 * g + 1 < DF, in1[g - gb], o1, tC);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    stg(bufA, g, at(qa, qb, q), o0[q], o1[q]);
            }
            }
        }
        stageFence();
        L3K_STAMP(5);
        // ---- quadrature points of the x-pencil (qy, qz) = (qa, qb): evalAtHexQPs, SumFactorization.hpp:707-753
        if (w_qq)
        {
            [[maybe_unused]] double en = 0.; // ENERGY (in-stage form): this pencil's share of x^T A x
            // (qw[q] * wyz is formed per point: hoisted, the 7 products cost 14 registers.)  alpha rides on the weight, so the
            // staged result needs no scaling pass; the ENERGY variant accumulates the unscaled x^T A x and scales at the end
            const double wyz = opaqueCopy(wyz_l) * (ENERGY || RHS ? 1. : a.alpha);
            double       G[6][3];
            if constexpr (VREG && EW == 1)
            {
                // one element per wave: the coordinates are wave-uniform -- lane t's value read into SGPRs (v_readlane), which
                // the geometry's FMAs take as operands: no LDS round trips (the LDS path read them back in 8 dependent steps)
                double vtx[24];
#pragma unroll
                for (int t = 0; t < 24; ++t)
                {
                    const int lo = __builtin_amdgcn_readlane(__double2loint(vreg), t), hi = __builtin_amdgcn_readlane(__double2hiint(vreg), t);
                    vtx[t]       = __hiloint2double(hi, lo);
                }
                hexPencilGeom(vtx, eta_l, zeta_l, G);
            }
            else
            {
                if constexpr (VREG)
                {
                    if (l < 24)
                        vs[l] = vreg;
                    stageFence();
                }
                hexPencilGeom(vs, eta_l, zeta_l, G);
            }
            // affine elements: one Jacobian per element -- inverted once here instead of once per quadrature point (50 of the
            // 139 instructions per point)
            {
                [[maybe_unused]] double Jm0[3][3], Ji0[3][3], x0[3], det0 = 0.;
                if constexpr (AFFINE)
                {
                    hexPointOnPencil(G, 0., Jm0, x0); // (G[3] = G[5] = 0: the Jacobian does not depend on xi)
                    det0 = inverse3(Jm0, Ji0);
                }
#pragma unroll
            for (int q = 0; q < NQ; ++q)
            {
                double vv[NF], dv[3][NF], r0[U], rd[3][U];
#pragma unroll
                for (int g = 0; g < DG; ++g)
                {
                    const double2 te = ldg(bufB, g, at(q, qa, qb)), tz = ldg(bufA, g, at(q, qa, qb));
                    dv[1][2 * g] = te.x;
                    dv[2][2 * g] = tz.x;
                    if (2 * g + 1 < NF)
                    {
                        dv[1][2 * g + 1] = te.y;
                        dv[2][2 * g + 1] = tz.y;
                    }
                }
#pragma unroll
                for (int o = 0; o < NF; ++o)
                {
                    vv[o] = v[q][o];
                    if (o < DF)
                        dv[0][o] = dxi[q][o];
                    else // the kernel declared that it never reads the derivatives of the external fields
                        dv[0][o] = dv[1][o] = dv[2][o] = 0.;
                }
                if constexpr (AFFINE)
                {
                    const double xyz[3] = {G[0][0] + qp[q] * G[1][0], G[0][1] + qp[q] * G[1][1], G[0][2] + qp[q] * G[1][2]};
                    qpStageAt< K, 1, RHS, 1, 0, ENERGY && !ENERGY_AT_END >(kern, Ji0, det0, xyz, qw[q] * wyz, a.time, vv, dv, r0, rd, &en, a.ref_z0 != 0);
                }
                else
                    qpStage< K, 1, RHS, 1, 0, ENERGY && !ENERGY_AT_END >(kern, G, qp[q], qw[q] * wyz, a.time, vv, dv, r0, rd, &en, a.ref_z0 != 0);
#pragma unroll
                for (int o = 0; o < U; ++o)
                {
                    v[q][o]   = r0[o];
                    dxi[q][o] = rd[0][o];
                }
#pragma unroll
                for (int g = 0; g < UG; ++g)
                {
                    stg(bufB, g, at(q, qa, qb), rd[1][2 * g], 2 * g + 1 < U ? rd[1][2 * g + 1] : 0.);
                    stg(bufA, g, at(q, qa, qb), rd[2][2 * g], 2 * g + 1 < U ? rd[2][2 * g + 1] : 0.);
                }
            }
            }
            if constexpr (ENERGY && !ENERGY_AT_END) // LDS atomic add of every pencil's share into the team's accumulator
                atomicAdd(vs + 24, en);
        }
        stageFence();
        L3K_STAMP(6);
        // ---- S8: C^T along eta in place in bufB (lane (qx,qz)); S9: C^T along zeta (lane (qx,qy)) added onto it: bufB then
        // holds g2 + g3 and the x-pencil stage reads one array (its 28 reads of two arrays came out one at a time, each with
        // its own LDS round trip, because v and dxi fill the register file there)
        if (w_qq)
        {
            double tCt[2 * HQ * HQ];
            loadTable(tCt, eoCt + opaqueZero());
            double e0[UG][NQ], e1[UG][NQ];
#pragma unroll
            for (int g = 0; g < UG; ++g)
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                {
                    const double2 t = ldg(bufB, g, at(qa, q, qb));
                    e0[g][q] = t.x;
                    e1[g][q] = t.y;
                }
#pragma unroll
            for (int g = 0; g < UG; ++g)
            {
                double o0[NQ], o1[NQ];
                sweepEO< NQ, NQ, true, false >(e0[g], o0, tCt);
                sweepEOSecond< NQ, NQ, true, false >(2 * g + 1 < U, e1[g], o1, tCt);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    stg(bufB, g, at(qa, q, qb), o0[q], o1[q]);
            }
            stageFence();
#pragma unroll
            for (int g = 0; g < UG; ++g)
            {
                double z0[NQ], z1[NQ], b0[NQ], b1[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                {
                    const double2 t = ldg(bufA, g, at(qa, qb, q));
                    z0[q] = t.x;
                    z1[q] = t.y;
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                {
                    const double2 t = ldg(bufB, g, at(qa, qb, q));
                    b0[q] = t.x;
                    b1[q] = t.y;
                }
                sweepEO< NQ, NQ, true, true >(z0, b0, tCt);
                sweepEOSecond< NQ, NQ, true, true >(2 * g + 1 < U, z1, b1, tCt);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    stg(bufB, g, at(qa, qb, q), b0[q], b1[q]);
            }
        }
        stageFence();
        L3K_STAMP(7);
        // ---- x-pencil (qy,qz): w = r0 + C^T r1 + g2 + g3, then I^T along x -> bufB (c=ix, b=qy, a=qz)
        if (w_qq)
        {
            double tCt[2 * HQ * HQ];
            loadTable(tCt, eoCt + opaqueZero());
#pragma unroll
            for (int g = 0; g < UG; ++g)
            {
                double r10[NQ], r11[NQ], w0[NQ], w1[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                {
                    const double2 tb = ldg(bufB, g, at(q, qa, qb));
                    r10[q] = dxi[q][2 * g];
                    r11[q] = dxi[q][2 * g + 1];
                    w0[q]  = v[q][2 * g] + tb.x;
                    w1[q]  = v[q][2 * g + 1] + tb.y;
                }
                sweepEO< NQ, NQ, true, true >(r10, w0, tCt);
                sweepEOSecond< NQ, NQ, true, true >(2 * g + 1 < U, r11, w1, tCt);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                {
                    v[q][2 * g]     = w0[q];
                    v[q][2 * g + 1] = w1[q];
                }
            }
            double tIt[2 * HQ * HN];
            loadTable(tIt, eoIt + opaqueZero());
#pragma unroll
            for (int g = 0; g < UG; ++g)
            {
                double w0[NQ], w1[NQ], o0[N1], o1[N1];
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                {
                    w0[q] = v[q][2 * g];
                    w1[q] = v[q][2 * g + 1];
                }
                sweepEO< NQ, N1, false, false >(w0, o0, tIt);
                sweepEOSecond< NQ, N1, false, false >(2 * g + 1 < U, w1, o1, tIt);
#pragma unroll
                for (int i = 0; i < N1; ++i)
                    stg(bufB, g, at(i, qa, qb), o0[i], o1[i]);
            }
        }
        stageFence();
        L3K_STAMP(8);
        // ---- I^T along y, lane (ix, qz) = (iq, kq): bufB (c=ix, b=qy, a=qz) -> bufA (c=ix, b=iy, a=qz)
        if (w_nq)
        {
            double tIt[2 * HQ * HN];
            loadTable(tIt, eoIt + opaqueZero());
            // (GB field groups per block: their LDS reads are issued before the first group's sweeps)
#pragma unroll
            for (int gb = 0; gb < UG; gb += GB)
            {
            double in0[GB][NQ], in1[GB][NQ];
#pragma unroll
            for (int g = gb; g < gb + GB && g < UG; ++g)
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                {
                    const double2 t = ldg(bufB, g, at(iq, q, kq));
                    in0[g - gb][q] = t.x;
                    in1[g - gb][q] = t.y;
                }
#pragma unroll
            for (int g = gb; g < gb + GB && g < UG; ++g)
            {
                double o0[N1], o1[N1];
                sweepEO< NQ, N1, false, false >(in0[g - gb], o0, tIt);
                sweepEOSecond< NQ, N1, false, false >(2 * g + 1 < U, in1[g - gb], o1, tIt);
#pragma unroll
                for (int j = 0; j < N1; ++j)
                    stg(bufA, g, at(iq, j, kq), o0[j], o1[j]);
            }
            }
        }
        stageFence();
        L3K_STAMP(9);
        // ---- I^T along z in registers, lane (ix, iy) = (i1, j1); stage the result in bufB as [node][unknown]
        if (w_nn)
        {
            double tIt[2 * HQ * HN];
            loadTable(tIt, eoIt + opaqueZero());
            double*       sb  = reinterpret_cast< double* >(bufB);
            [[maybe_unused]] double              en_e  = 0.;
            [[maybe_unused]] const double* const ax_e  = a.x;
            [[maybe_unused]] const double* const axg_e = a.xg;
            const uint4   srow = slotRows[l]; // scatter slots of this lane's N1 nodes
            const uint32_t sw[4] = {srow.x, srow.y, srow.z, srow.w};
            // (GB field groups per block: their LDS reads are issued before the first group's sweeps)
#pragma unroll
            for (int gb = 0; gb < UG; gb += GB)
            {
            double in0[GB][NQ], in1[GB][NQ];
#pragma unroll
            for (int g = gb; g < gb + GB && g < UG; ++g)
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                {
                    const double2 t = ldg(bufA, g, at(i1, j1, q));
                    in0[g - gb][q] = t.x;
                    in1[g - gb][q] = t.y;
                }
#pragma unroll
            for (int g = gb; g < gb + GB && g < UG; ++g)
            {
                double o0[N1], o1[N1];
                sweepEO< NQ, N1, false, false >(in0[g - gb], o0, tIt);
                sweepEOSecond< NQ, N1, false, false >(2 * g + 1 < U, in1[g - gb], o1, tIt);
                // scatterSumFact skips Dirichlet dofs (MatrixFreeSystem.hpp:517-536).  Here their staged values become 0 -- the
                // scatter then ADDS 0 to those rows (or stores beta * y on exclusive ones), which leaves them as they are -- so
                // that elements touching the Dirichlet boundary take the same unrolled scatter as all others (they used to loop
                // with a mask byte load per dof and round: the chunks holding Dirichlet faces ran 7 % longer).  The mask bytes
                // are fetched again here instead of kept from the gather (registers)
#ifndef L3K_FLAGGED_SCATTER // (A/B switch: the round-2 form with a scatter path of its own for flagged elements)
                if (!RHS && (flag_cur & 1u) != 0) // (RHS: every row receives its share; the finalize pass overwrites the Dirichlet rows)
#else
                if (false)
#endif
                {
#pragma unroll
                    for (int k = 0; k < N1; ++k)
                    {
                        const int64_t  node = static_cast< uint32_t >(opaqueCopy(static_cast< int >(ids_cur[k])));
                        const uint32_t dmk  = dirBits(node) >> (2 * g);
                        o0[k]               = (dmk & 1u) ? 0. : o0[k];
                        o1[k]               = (dmk & 2u) ? 0. : o1[k];
                    }
                }
#pragma unroll
                for (int k = 0; k < N1; ++k)
                {
                    const uint32_t slot = (k & 1) ? sw[k >> 1] >> 16 : sw[k >> 1] & 0xffffu;
                    double*        dstl = sb + slot * U + 2 * g;
                    if constexpr (U % 2 == 0)
                        *reinterpret_cast< double2* >(dstl) = ENERGY ? make_double2(a.alpha * o0[k], a.alpha * o1[k]) : make_double2(o0[k], o1[k]);
                    else // (an odd U: node rows of the staging buffer are 8-byte aligned only)
                    {
                        dstl[0] = ENERGY ? a.alpha * o0[k] : o0[k];
                        if (2 * g + 1 < U)
                            dstl[1] = ENERGY ? a.alpha * o1[k] : o1[k];
                    }
                }
                if constexpr (ENERGY && ENERGY_AT_END)
                {
                    // x^T A x of this element = x_e . y_e: this lane holds the (unscaled) result of exactly the nodes it gathered,
                    // so its share is a dot product with their x values, fetched once more here (L2 hits: the element's rows
                    // were read at the top of the element) -- the accumulation inside the quadrature stage, where no register is
                    // free, made this variant spill (12 registers at order 6, 10 for config 5's kernel at order 4).  Dirichlet
                    // dofs were gathered as 0 and are skipped by the scatter: they do not take part
#pragma unroll
                    for (int k = 0; k < N1; ++k)
                    {
                        // (an opaque copy of the id: with the plain one the compiler merges this load with the gather's and keeps
                        // the element's x values in registers across all stages -- 34 spilled registers)
                        const int64_t node = static_cast< uint32_t >(opaqueCopy(static_cast< int >(ids_cur[k])));
                        const double* px   = (!SPLIT || node < n_owned_nodes ? ax_e + node * U : axg_e + (node - n_owned_nodes) * U) + 2 * g;
                        if constexpr (U % 2 == 0)
                        {
                            const double2 xv = *reinterpret_cast< const double2* >(px);
                            en_e += xv.x * o0[k] + xv.y * o1[k]; // (o0, o1 are 0 on Dirichlet dofs: zeroed above)
                        }
                        else
                        {
                            en_e += px[0] * o0[k];
                            if (2 * g + 1 < U)
                                en_e += px[1] * o1[k];
                        }
                    }
                }
            }
            }
            if constexpr (ENERGY && ENERGY_AT_END) // LDS atomic add of every lane's share into the team's accumulator
                atomicAdd(vs + 24, en_e);
        }
        stageFence();
        L3K_STAMP(10);
        // bufA is free now: stash this element's node ids there in slot order for the scatter (LDS latency instead of a
        // dependent global load per scatter round)
        if (w_nn)
        {
            uint32_t* const idsL = reinterpret_cast< uint32_t* >(bufA);
            const uint4     srow = slotRows[l];
            const uint32_t  sw[4] = {srow.x, srow.y, srow.z, srow.w};
#pragma unroll
            for (int k = 0; k < N1; ++k)
                idsL[(k & 1) ? sw[k >> 1] >> 16 : sw[k >> 1] & 0xffffu] = ids_cur[k];
        }
        } // if (w_all)
        stageFence(); // (every lane: the scatter below reads what the pencil lanes staged)
        L3K_STAMP(11);
        if (!MULTI || col == (a.n_cols > 1 ? a.n_cols : 1) - 1)
        {
            batch_next = dyn ? ticketBatch(ticket) : batch + stride;
            loadIds(batch_next, ids_nxt, flag_nxt); // next element's node ids: in flight behind the scatter
        }
        double* const ay  = MULTI ? a.y + size_t(col) * a.ldy : a.y;
        double* const ayg = MULTI && SPLIT ? a.yg + size_t(col) * a.ldyg : a.yg;
        // ---- scatter (scatterSumFact, MatrixFreeSystem.hpp:494-537) in SLOT order by all 64 lanes: slots [0, nsh) are the
        // shell nodes in ascending-id order of a typical element -- one double per lane and round, so that an atomic
        // wave-instruction covers runs of contiguous dofs with every 64-byte request full (the memory-side atomic units
        // are a chip-wide request-rate limit: DESIGN.md 4.1) -- and slots [nsh, NN) the element's exclusive nodes: 16-byte
        // stores of alpha*A*x + beta*y.  Without an exclusive range (fuse_beta == 0) every node takes the atomic path.
        // (the scatter's per-lane constants are rebuilt here from an opaque copy of the lane id: as loop invariants they
        // would be kept in registers across the sweeps, which have none to spare)
        const int lane_s = opaqueCopy(lane);
        const int steam_s = lane_s / SG, sl = lane_s - steam_s * SG;
        if (steam_s < EW && (int64_t(batch) * EW + steam_s) < a.elem_count)
        {
            const double* const   sb      = lds + size_t(steam_s) * Cfg::TEAM_D + Cfg::OFF_B;
            const uint32_t* const idsS    = reinterpret_cast< const uint32_t* >(lds + size_t(steam_s) * Cfg::TEAM_D);
            // (Dirichlet dofs were zeroed when the result was staged: every element takes the common path)
#ifndef L3K_FLAGGED_SCATTER
            constexpr bool        flagged = false;
#else
            const bool            flagged = (flag_cur & 2u) != 0;
#endif
            // (SG is a multiple of U: a lane keeps its unknown and moves SG / U slots per round -- every LDS address below is
            // one per-lane base plus a compile-time offset)
            static_assert(SG % U == 0);
            // exclusive (element-internal) nodes: 16-byte stores, two unknowns per lane -- or, with an odd number of unknowns
            // (8-byte aligned rows), one unknown per lane like the shell slots: XW doubles per lane, UX = U / XW lanes per node
            constexpr int         XW = U % 2 == 0 ? 2 : 1, UX = U / XW;
            using xval_t = std::conditional_t< XW == 2, double2, double >;
            const int             sl_node = sl / U, sl_o = sl % U, sl_node2 = sl / UX, sl_o2 = XW * (sl % UX);
            [[maybe_unused]] const int sl_f = STRIDED ? a.field_inds[sl_o] : sl_o; // (STRIDED: the node dof of this lane's unknown)
            const uint32_t* const ids1 = idsS + sl_node;
            const double* const   sb1  = sb + sl;
            const uint32_t* const ids2 = idsS + sl_node2;
            const xval_t* const   sb2  = reinterpret_cast< const xval_t* >(sb) + sl;
            // FLAGGED: the element touches a Dirichlet dof (those dofs are skipped / written as beta*y, :517-536).  Only the
            // common variant (exclusive range present, no Dirichlet dof) is unrolled: without branches inside the rounds
            // its LDS reads are issued ahead of the address arithmetic of earlier rounds
            auto shellRound = [&]< int NSH_, bool FLAGGED >(int r) {
                if ((r + 1) * SG > NSH_ * U && r * SG + sl >= NSH_ * U)
                    return;
                // ablation 128: every other shell round dropped -- half the atomic adds (and their LDS reads and address arithmetic):
                // a generous upper bound for what summing shared faces of an element block in LDS could save (wrong results)
                if ((L3K_DBG(a) & 128) && (r & 1))
                    return;
                const int64_t node = ids1[r * (SG / U)];
                const int64_t dof  = STRIDED ? node * dpn + sl_f : node * U + sl_o;
                const double  val  = sb1[r * SG];
                double*       dst  = !SPLIT || node < n_owned_nodes ? ay + dof : ayg + (dof - a.n_owned_dofs);
                if constexpr (FLAGGED)
                    if (a.dirichlet[dof] != 0)
                        return;
                if (L3K_DBG(a) & (1 | 32 | 64)) // ablation: 1 / 64 = no memory operation, 32 = plain store
                {
                    if ((L3K_DBG(a) & 32) || val == 1.2345e300)
                        *dst = val;
                }
                else if (L3K_DBG(a) & 16)
                    *dst += val;
                else
                    unsafeAtomicAdd(dst, val);
            };
            // out = val + beta * old on an exclusive row (the pre-scaling pass skipped it), for both widths
            auto exclStore = [&](double* dst, xval_t out) {
                if constexpr (XW == 2)
                {
                    if (a.beta != 0.)
                    {
                        const double2 old = *reinterpret_cast< const double2* >(dst);
                        out.x += a.beta * old.x;
                        out.y += a.beta * old.y;
                    }
                    *reinterpret_cast< double2* >(dst) = out;
                }
                else
                    *dst = a.beta != 0. ? out + a.beta * *dst : out;
            };
            auto exclRound = [&]< int NSH_, bool FLAGGED >(int r) {
                if (NSH_ * UX + r * SG + sl >= NN * UX)
                    return;
                const int64_t node = ids2[NSH_ + r * (SG / UX)];
                const int64_t dof  = node * U + sl_o2;
                const xval_t  val  = sb2[NSH_ * UX + r * SG];
                double*       dst  = ay + dof; // (exclusive nodes are owned)
                xval_t        out  = val;
                if constexpr (FLAGGED)
                {
                    if constexpr (XW == 2)
                    {
                        out.x = a.dirichlet[dof] != 0 ? 0. : val.x;
                        out.y = a.dirichlet[dof + 1] != 0 ? 0. : val.y;
                    }
                    else
                        out = a.dirichlet[dof] != 0 ? 0. : val;
                }
                if (L3K_DBG(a) & 1)
                {
                    double first;
                    if constexpr (XW == 2)
                        first = val.x;
                    else
                        first = val;
                    if (first == 1.2345e300)
                        *dst = first;
                    return;
                }
                exclStore(dst, out);
            };
            constexpr int RS = (Cfg::NSH * U + SG - 1) / SG, RX = ((NN - Cfg::NSH) * UX + SG - 1) / SG;
            if (!STRIDED && !RHS && a.fuse_beta && !flagged)
            {
#ifdef L3K_ABLATION
#pragma unroll
                for (int r = 0; r < RS; ++r)
                    shellRound.template operator()< Cfg::NSH, false >(r);
#pragma unroll
                for (int r = 0; r < RX; ++r)
                    exclRound.template operator()< Cfg::NSH, false >(r);
#else
                // the common case in two phases: every LDS read of the element first (ids and values of all rounds: one LDS
                // round trip instead of one per round -- the registers of the sweeps are free here), then the address
                // arithmetic and the memory instructions back to back.  Full rounds are unconditional; only the last round
                // of each kind has lanes beyond the end
                constexpr int  NSHU = Cfg::NSH * U, NXH = (NN - Cfg::NSH) * UX;
                uint32_t       nid[RS], nid2[RX > 0 ? RX : 1]; // (order 1 has no exclusive slots)
                double         val[RS];
                xval_t         val2[RX > 0 ? RX : 1];
                constexpr bool part1 = RS * SG > NSHU, part2 = RX * SG > NXH;
                const bool     in1 = !part1 || (RS - 1) * SG + sl < NSHU, in2 = !part2 || (RX - 1) * SG + sl < NXH;
#pragma unroll
                for (int r = 0; r < RS; ++r)
                {
                    nid[r] = ids1[r * (SG / U)]; // (beyond the shell range these read the exclusive slots: unused)
                    val[r] = sb1[r * SG];
                }
#pragma unroll
                for (int r = 0; r < RX; ++r)
                    if (r + 1 < RX || in2)
                    {
                        nid2[r] = ids2[Cfg::NSH + r * (SG / UX)];
                        val2[r] = sb2[Cfg::NSH * UX + r * SG];
                    }
#pragma unroll
                for (int r = 0; r < RS; ++r)
                    if (r + 1 < RS || in1)
                    {
                        const int64_t node = nid[r];
                        const int64_t dof  = node * U + sl_o;
                        double*       dst  = !SPLIT || node < n_owned_nodes ? ay + dof : ayg + (dof - a.n_owned_dofs);
                        unsafeAtomicAdd(dst, val[r]);
                    }
#pragma unroll
                for (int r = 0; r < RX; ++r)
                    if (r + 1 < RX || in2)
                    {
                        exclStore(ay + int64_t(nid2[r]) * U + sl_o2, val2[r]); // (exclusive nodes are owned)
                    }
#endif
            }
            else if (!STRIDED && !RHS && a.fuse_beta)
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
        stageFence(); // the buffers are rewritten by the next batch
        L3K_STAMP(12);
        } while (MULTI && ++col < a.n_cols);
#pragma unroll
        for (int k = 0; k < N1; ++k)
            ids_cur[k] = ids_nxt[k];
        flag_cur = flag_nxt;
        batch    = batch_next;
        ++stamp_it;
    }
    if constexpr (ENERGY)
    {
        stageFence();
        if (worker && l == 0 && vs[24] != 0.)
            unsafeAtomicAdd(a.energy, vs[24]); // one global atomic per team and launch
    }
#ifdef L3K_ABLATION
    if (a.stamps != nullptr && blockIdx.x < 4096 && lane == 0)
        a.stamps[256 * 16 + 2 * blockIdx.x + 1] = __builtin_readcyclecounter();
#endif
}

// What a launch of the single-wave kernel will do: decided in ONE place (planSumfactFast), used by the launcher and by the route
// report of l3k_mf_route.  The settings come from the context's l3k_tuning (read from the environment once, at l3k_ctx_create).
struct FastRoute
{
    bool     generic = false; // the launch goes to the generic LDS kernel instead (small launch, or a non-dense dof layout)
    bool     split = false, affine = false, energy = false, multi = false, dynamic = false;
    bool     strided = false; // the kernel's unknowns are a subset of the node's dofs: 8-byte gather / scatter variant
    int      waves_cu = 0, xcd_chunk = 0, n_cus = 0;
    unsigned grid = 0;
    int64_t  n_batches = 0;
};
template < typename K, int P, int NQ, bool MULTI >
int planSumfactFast(const ElemArgs& a, FastRoute& r)
{
    using Cfg = FastCfg< K, P, NQ >;
    const l3k_tuning& tune = tuneOf(a);
    r       = FastRoute{};
    r.multi = MULTI;
    // Small launches are latency-bound: one element takes ~23 us through a single wave here, ~15 us through the 6-wave
    // workgroup of the generic kernel; below ~3 elements per CU the generic kernel wins (profiles/r01_kbench_small_meshes.log:
    // order 6, 216 elements 16 vs 32 us, 1000 elements 34 vs 44 us, crossover at ~1700 elements): l3k_tuning::generic_below
    constexpr bool generic_fits = applyLdsBytes< K, P, NQ, 1 >() <= lds_limit_bytes;
    // (non-dense dof layouts: the strided variant -- single-column applies; it does not accumulate x^T A x: the caller's
    // l3k_mf_energy_end then reports "not fused" and the dot product runs as a pass of its own)
    if ((!a.dense && (MULTI || a.fuse_beta)) || (generic_fits && a.elem_count < tune.generic_below))
    {
        r.generic = true;
        return 0;
    }
    r.strided = !a.dense;
    // (ghost rows directly behind the owned rows of every column: one base pointer per column serves both)
    const bool contiguous = (a.xg == nullptr || (a.xg == a.x + a.n_owned_dofs && (!MULTI || a.ldxg == a.ldx))) &&
                            (a.yg == nullptr || (a.yg == a.y + a.n_owned_dofs && (!MULTI || a.ldyg == a.ldy)));
    r.split  = !contiguous;
    // (the affine variant exists for the plain apply: no ghost buffers, no fused energy)
    r.affine = !MULTI && a.all_affine && !r.split && !a.energy && !tune.no_affine && !r.strided;
    r.energy = !MULTI && a.energy != nullptr && !r.strided;
    // launch configuration per device (several contexts of one process may sit on different GPUs): the dynamic-LDS
    // attribute of the variants is set once on each device, under a lock
    struct PerDevice
    {
        bool ready = false;
        int  n_cus = 0, waves_cu = 0;
    };
    static PerDevice  per_device[64];
    static std::mutex per_device_mutex;
    int               dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64)
    {
        setError("device index %d not supported", dev);
        return -3;
    }
    {
        std::lock_guard< std::mutex > lock{per_device_mutex};
        PerDevice&                    pd = per_device[dev];
        if (!pd.ready)
        {
            std::vector< const void* > variants;
            if constexpr (MULTI)
                variants = {reinterpret_cast< const void* >(sumfactFastKernel< K, P, NQ, true, false, false, true >),
                            reinterpret_cast< const void* >(sumfactFastKernel< K, P, NQ, false, false, false, true >)};
            else
                variants = {reinterpret_cast< const void* >(sumfactFastKernel< K, P, NQ, true, true >),
                            reinterpret_cast< const void* >(sumfactFastKernel< K, P, NQ, false, true >),
                            reinterpret_cast< const void* >(sumfactFastKernel< K, P, NQ, true, false >),
                            reinterpret_cast< const void* >(sumfactFastKernel< K, P, NQ, false, false >),
                            reinterpret_cast< const void* >(sumfactFastKernel< K, P, NQ, false, false, true >),
                            reinterpret_cast< const void* >(sumfactFastKernel< K, P, NQ, true, false, false, false, true >),
                            reinterpret_cast< const void* >(sumfactFastKernel< K, P, NQ, false, false, false, false, true >)};
            bool ok = true;
            for (const void* f : variants)
                ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, int(Cfg::lds)) == hipSuccess;
            if (!ok)
            {
                setError("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu) failed", Cfg::lds);
                return -3;
            }
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) != hipSuccess)
            {
                setError("hipGetDeviceProperties failed");
                return -3;
            }
            pd.n_cus = prop.multiProcessorCount;
            // resident single-wave workgroups per CU: limited by LDS (160 KiB) and by the VGPR budget (min_waves per SIMD)
            const int by_lds = int((160 * 1024) / Cfg::lds);
            pd.waves_cu      = by_lds < 1 ? 1 : (by_lds > 4 * Cfg::wavesPerSimd(MULTI) ? 4 * Cfg::wavesPerSimd(MULTI) : by_lds);
            pd.ready         = true;
        }
        r.n_cus = pd.n_cus, r.waves_cu = pd.waves_cu;
    }
    if (tune.waves_per_cu > 0)
        r.waves_cu = tune.waves_per_cu;
    r.n_batches              = (a.elem_count + Cfg::EW - 1) / Cfg::EW;
    const int64_t max_blocks = int64_t(r.n_cus) * r.waves_cu;
    r.grid                   = static_cast< unsigned >(r.n_batches < max_blocks ? r.n_batches : max_blocks);
    // contiguous eighths of the batches per XCD, where every XCD group gets the same number of persistent workgroups.  Measured
    // (profiles/r01_kbench_xcd_mapping.log): order 6 gains 1 % and re-fetches less, order 4 loses 11 % (the elements in flight on
    // one XCD are neighbours: their atomics meet on the same lines) -- hence orders >= 6 only
    r.xcd_chunk = (P >= 6 && r.grid % 8 == 0 && r.n_batches >= int64_t(r.grid) && r.n_batches < (int64_t(1) << 30)) ? int((r.n_batches + 7) / 8) : 0;
    r.dynamic   = a.work_counters != nullptr;
    return 0;
}

template < typename K, int P, int NQ, bool MULTI >
int launchSumfactFastImpl(const ElemArgs& a, const void* kparam_blob, hipStream_t stream)
{
    using Cfg = FastCfg< K, P, NQ >;
    if (a.elem_count <= 0)
        return 0;
    if (MULTI && a.n_cols < 1)
    {
        setError("multi-column element launch without a column count");
        return -1;
    }
    FastRoute r;
    if (int rc = planSumfactFast< K, P, NQ, MULTI >(a, r))
        return rc;
    if (r.generic)
    {
        if constexpr (!MULTI)
            return launchSumfactApply< K, P, NQ, 1, false >(a, kparam_blob, stream);
        else
        {
            for (int c = 0; c < a.n_cols; ++c) // column by column through the generic single-column kernel
            {
                ElemArgs ac = a;
                ac.n_cols   = 1;
                ac.x        = a.x + a.ldx * c;
                ac.xg       = a.xg ? a.xg + a.ldxg * c : nullptr;
                ac.y        = a.y + a.ldy * c;
                ac.yg       = a.yg ? a.yg + a.ldyg * c : nullptr;
                if (int rc = launchSumfactApply< K, P, NQ, 1, false >(ac, kparam_blob, stream))
                    return rc;
            }
            return 0;
        }
    }
    K kern{};
    if (kparam_blob)
        __builtin_memcpy(&kern, kparam_blob, sizeof(K));
    decltype(&sumfactFastKernel< K, P, NQ, false, false >) kernel;
    if constexpr (MULTI) // (no fused energy, no affine variant: plain applies of several columns)
        kernel = r.split ? sumfactFastKernel< K, P, NQ, true, false, false, true > : sumfactFastKernel< K, P, NQ, false, false, false, true >;
    else if (r.strided)
        kernel = r.split ? sumfactFastKernel< K, P, NQ, true, false, false, false, true > : sumfactFastKernel< K, P, NQ, false, false, false, false, true >;
    else
        kernel = r.affine ? sumfactFastKernel< K, P, NQ, false, false, true >
                 : r.energy ? (r.split ? sumfactFastKernel< K, P, NQ, true, true > : sumfactFastKernel< K, P, NQ, false, true >)
                            : (r.split ? sumfactFastKernel< K, P, NQ, true, false > : sumfactFastKernel< K, P, NQ, false, false >);
    constexpr TableLayout   TL{P + 1, NQ};
    FastTables< P + 1, NQ > tab;
    const double*           th = a.tables_host;
    __builtin_memcpy(tab.eoI, th + TL.offEoI(), sizeof tab.eoI);
    __builtin_memcpy(tab.eoC, th + TL.offEoC(), sizeof tab.eoC);
    __builtin_memcpy(tab.eoIt, th + TL.offEoIt(), sizeof tab.eoIt);
    __builtin_memcpy(tab.eoCt, th + TL.offEoCt(), sizeof tab.eoCt);
    __builtin_memcpy(tab.qw, th + TL.offW(), sizeof tab.qw);
    __builtin_memcpy(tab.qx, th + TL.offX(), sizeof tab.qx);
    if (a.work_counters && hipMemsetAsync(a.work_counters, 0, 8 * 128, stream) != hipSuccess)
    {
        setError("hipMemsetAsync(work counters) failed");
        return -3;
    }
    hipLaunchKernelGGL(kernel, dim3(r.grid), dim3(64), Cfg::lds, stream, a, kern, r.n_batches, r.xcd_chunk, tab);
    if (r.energy && a.energy_done)
        ++*a.energy_done;
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess)
    {
        setError("sumfactFastKernel launch failed: %s", hipGetErrorString(err));
        return -3;
    }
    return 0;
}
// the route of an apply launch as text (l3k_mf_route): which kernel template, which variant, how it is launched
template < typename K, int P, int NQ >
int describeSumfactFast(const ElemArgs& a, char* buf, size_t n)
{
    using Cfg = FastCfg< K, P, NQ >;
    FastRoute r;
    const bool multi = Cfg::multi_column && a.n_cols > 1;
    if (int rc = multi ? planSumfactFast< K, P, NQ, Cfg::multi_column >(a, r) : planSumfactFast< K, P, NQ, false >(a, r))
        return rc;
    if (r.generic)
        return describeSumfactApply< K, P, NQ, 1 >(a, buf, n);
    std::snprintf(buf, n,
                  "sumfactFastKernel<p=%d,nq=%d,U=%d,F=%d>%s%s%s%s%s: one wave per %d element(s), %d of 64 lanes, %zu B LDS/wave, "
                  "%d waves/CU x %d CUs = grid %u, %s batches%s",
                  P, NQ, Cfg::U, Cfg::F, r.affine ? " affine" : "", r.energy ? " energy" : "", r.split ? " split-ghost" : "",
                  r.multi ? " multi-column" : "", r.strided ? " strided-dofs" : "", Cfg::EW, Cfg::EW * Cfg::TEAM, size_t(Cfg::lds), r.waves_cu, r.n_cus, r.grid,
                  r.dynamic ? "dynamic" : "static", r.xcd_chunk ? ", XCD-chunked" : "");
    return 0;
}
// the right-hand side of one column with Dirichlet lifting on the single-wave kernel (RHS variant); returns 1 where the launch
// belongs to the generic kernel in RHS mode (small launches, other dof layouts, element-local output, several columns)
template < typename K, int P, int NQ >
int launchSumfactFastRhs(const ElemArgs& a, const void* kparam_blob, hipStream_t stream)
{
    using Cfg = FastCfg< K, P, NQ >;
    if (a.elem_count <= 0)
        return 0;
    if (!a.dense || a.local_out || a.n_cols > 1)
        return 1;
    ElemArgs ar  = a;
    ar.energy    = nullptr;
    ar.fuse_beta = 0;
    ar.alpha     = 1.;
    ar.beta      = 1.;
    ar.x         = a.y; // (unused by the gather; the pointer relations below decide the ghost-buffer variant)
    ar.xg        = a.yg;
    FastRoute r;
    if (int rc = planSumfactFast< K, P, NQ, false >(ar, r))
        return rc;
    if (r.generic)
        return 1;
    {
        static std::mutex attr_mutex;
        static bool       attr_set[64] = {};
        int               dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard< std::mutex > lock{attr_mutex};
        if (dev >= 0 && dev < 64 && !attr_set[dev])
        {
            if (hipFuncSetAttribute(reinterpret_cast< const void* >(sumfactFastKernel< K, P, NQ, true, false, false, false, false, true >),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, int(Cfg::lds)) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast< const void* >(sumfactFastKernel< K, P, NQ, false, false, false, false, false, true >),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, int(Cfg::lds)) != hipSuccess)
            {
                setError("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu) failed", Cfg::lds);
                return -3;
            }
            attr_set[dev] = true;
        }
    }
    K kern{};
    if (kparam_blob)
        __builtin_memcpy(&kern, kparam_blob, sizeof(K));
    auto kernel = r.split ? sumfactFastKernel< K, P, NQ, true, false, false, false, false, true >
                          : sumfactFastKernel< K, P, NQ, false, false, false, false, false, true >;
    constexpr TableLayout   TL{P + 1, NQ};
    FastTables< P + 1, NQ > tab;
    const double*           th = a.tables_host;
    __builtin_memcpy(tab.eoI, th + TL.offEoI(), sizeof tab.eoI);
    __builtin_memcpy(tab.eoC, th + TL.offEoC(), sizeof tab.eoC);
    __builtin_memcpy(tab.eoIt, th + TL.offEoIt(), sizeof tab.eoIt);
    __builtin_memcpy(tab.eoCt, th + TL.offEoCt(), sizeof tab.eoCt);
    __builtin_memcpy(tab.qw, th + TL.offW(), sizeof tab.qw);
    __builtin_memcpy(tab.qx, th + TL.offX(), sizeof tab.qx);
    if (ar.work_counters && hipMemsetAsync(ar.work_counters, 0, 8 * 128, stream) != hipSuccess)
    {
        setError("hipMemsetAsync(work counters) failed");
        return -3;
    }
    hipLaunchKernelGGL(kernel, dim3(r.grid), dim3(64), Cfg::lds, stream, ar, kern, r.n_batches, r.xcd_chunk, tab);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess)
    {
        setError("sumfactFastKernel (rhs) launch failed: %s", hipGetErrorString(err));
        return -3;
    }
    return 0;
}
template < typename K, int P, int NQ >
int launchSumfactFast(const ElemArgs& a, const void* kparam_blob, hipStream_t stream)
{
    return launchSumfactFastImpl< K, P, NQ, false >(a, kparam_blob, stream);
}
// a.n_cols columns (x, y: column c at + c * ld) in one pass over the elements
template < typename K, int P, int NQ >
int launchSumfactFastCols(const ElemArgs& a, const void* kparam_blob, hipStream_t stream)
{
    static_assert(FastCfg< K, P, NQ >::multi_column);
    if (a.n_cols <= 1)
        return launchSumfactFastImpl< K, P, NQ, false >(a, kparam_blob, stream);
    return launchSumfactFastImpl< K, P, NQ, true >(a, kparam_blob, stream);
}
} // namespace l3k::dev
#endif
