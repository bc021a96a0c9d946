// api_assembled.hip -- scatter of local systems into the global system on the device: the assembled path's hand-off.
//
// Reference: algsys/ScatterLocalSystem.hpp:24-54 (per local row one CrsMatrix::sumIntoLocalValues(row, cols, vals) and an
// atomic add per right-hand side) called per element from algsys/AssembleGlobalSystem.hpp:20-53.  Here a whole batch of
// element matrices (the output of l3k_local_assemble, resident in HBM) is summed into the VALUES ARRAY of the caller's
// CSR graph (local row pointers / sorted local column indices, i.e. what Tpetra::CrsGraph::getLocalRowPtrsDevice /
// getLocalIndicesDevice hand out): the Tpetra side takes the finished values with one setAllValues, or one
// sumIntoLocalValues per row batch -- not one call per element row.
#include "objects.hpp"

#include <cstdlib>

namespace
{
struct ScatterArgs
{
    const uint32_t* elem_nodes;
    const uint8_t*  dirichlet; // per local dof, or null
    const double *  K, *F;
    const int64_t*  row_ptr;
    const int32_t*  col_ind;
    double *        values, *rhs;
    size_t          ldr;
    unsigned long long* n_missing;
    int64_t         first, count;
    int             NN, dpn, n_rhs, skip_dirichlet, U_rt;
    int             field_inds[l3k::dev::max_unknowns];
};

// The round-2 form, kept behind l3k_tuning::scatter_per_entry as the cross-check (tests) and the baseline (tools/bench_assembled_pipeline.py):
// one wave per (element, local row), a binary search per ENTRY
__global__ __launch_bounds__(64) void assembledScatterPerEntryKernel(const ScatterArgs a)
{
    const int     Nd   = a.NN * a.U_rt;
    const int64_t e    = blockIdx.x / Nd; // element of the batch
    const int     i    = int(blockIdx.x - e * Nd);
    const uint32_t* en = a.elem_nodes + (a.first + e) * a.NN;
    const int64_t row  = int64_t(en[i / a.U_rt]) * a.dpn + a.field_inds[i % a.U_rt];
    if (a.skip_dirichlet && a.dirichlet && a.dirichlet[row])
        return;
    const int lane = threadIdx.x;
    if (a.F && a.rhs)
        for (int r = lane; r < a.n_rhs; r += 64) // (more than 64 right-hand sides: several rounds)
            unsafeAtomicAdd(a.rhs + size_t(r) * a.ldr + row, a.F[(e * a.n_rhs + r) * Nd + i]);
    if (!a.K || !a.values)
        return;
    const int64_t  rb = a.row_ptr[row], re = a.row_ptr[row + 1];
    const double*  Kr = a.K + (e * Nd + i) * int64_t(Nd);
    unsigned       missing = 0;
    for (int j = lane; j < Nd; j += 64)
    {
        const int64_t col = int64_t(en[j / a.U_rt]) * a.dpn + a.field_inds[j % a.U_rt];
        if (a.skip_dirichlet && a.dirichlet && a.dirichlet[col])
            continue;
        int64_t lo = rb, hi = re; // first position with col_ind >= col
        while (lo < hi)
        {
            const int64_t mid = (lo + hi) >> 1;
            if (a.col_ind[mid] < col)
                lo = mid + 1;
            else
                hi = mid;
        }
        if (lo < re && a.col_ind[lo] == col)
            unsafeAtomicAdd(a.values + lo, Kr[j]);
        else
            ++missing; // (Tpetra's sumIntoLocalValues skips entries outside the graph and reports how many it took)
    }
    if (missing && a.n_missing)
        atomicAdd(a.n_missing, static_cast< unsigned long long >(missing));
}

// first position in [lo, hi) with col_ind >= col
__device__ __forceinline__ int64_t lowerBound(const int32_t* __restrict__ col_ind, int64_t lo, int64_t hi, int64_t col)
{
    while (lo < hi)
    {
        const int64_t mid = (lo + hi) >> 1;
        if (col_ind[mid] < col)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// One wave per (element, ROW NODE b): the U rows (b, u) of K_e, the lanes over the Nd consecutive ENTRIES of a row, so that a
// wave-instruction reads 512 contiguous bytes of K_e and its atomic adds fall on runs of neighbouring CSR values (dense
// 64-byte requests: the memory-side atomic units are a request-rate limit -- a first form with one column NODE per lane,
// i.e. atomics at a stride of U values, ran at 0.6 x the rate of the per-entry kernel above although it searched 16 x less).
// The position of an entry's column in the CSR row of (b, u = 0) is found by ONE binary search and reused for the rows
// u = 1 .. U-1: the rows of one node carry the same columns in a finite-element graph; the candidate position is checked
// against col_ind and a row without that structure falls back to its own search (same results).
template < int U >
__global__ __launch_bounds__(64) void assembledScatterKernel(const ScatterArgs a)
{
    const int       Nd   = a.NN * U;
    const int64_t   e    = blockIdx.x / a.NN; // element of the batch
    const int       b    = int(blockIdx.x - e * a.NN);
    const uint32_t* en   = a.elem_nodes + (a.first + e) * a.NN;
    const int       lane = threadIdx.x;
    const int64_t   nb   = int64_t(en[b]) * a.dpn;
    int64_t         row[U], rb[U], re[U];
    bool            live[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
        row[u]  = nb + a.field_inds[u];
        live[u] = !(a.skip_dirichlet && a.dirichlet && a.dirichlet[row[u]]);
    }
    if (a.F && a.rhs)
        for (int t = lane; t < a.n_rhs * U; t += 64) // (n_rhs * U may exceed the wave: e.g. 17 right-hand sides of 4 unknowns)
        {
            const int r = t / U, u = t - r * U;
            bool      lv = false;
            int64_t   rw = 0;
#pragma unroll
            for (int uu = 0; uu < U; ++uu) // (compile-time indexing of the per-unknown registers)
                if (uu == u)
                {
                    lv = live[uu];
                    rw = row[uu];
                }
            if (lv)
                unsafeAtomicAdd(a.rhs + size_t(r) * a.ldr + rw, a.F[(e * a.n_rhs + r) * Nd + b * U + u]);
        }
    if (!a.K || !a.values)
        return;
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
        rb[u] = a.row_ptr[row[u]];
        re[u] = a.row_ptr[row[u] + 1];
    }
    const double* Kb      = a.K + (e * Nd + int64_t(b) * U) * Nd; // rows (b, 0 .. U-1)
    unsigned      missing = 0;
    for (int j = lane; j < Nd; j += 64)
    {
        const int     bp  = j / U;
        const int64_t col = int64_t(en[bp]) * a.dpn + a.field_inds[j - bp * U];
        if (a.skip_dirichlet && a.dirichlet && a.dirichlet[col])
            continue;
        const int64_t rel = lowerBound(a.col_ind, rb[0], re[0], col) - rb[0]; // the entry's one search
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            if (!live[u])
                continue;
            int64_t pos = rb[u] + rel;
            if (u > 0 && !(pos < re[u] && a.col_ind[pos] == col))
                pos = lowerBound(a.col_ind, rb[u], re[u], col);
            if (pos < re[u] && a.col_ind[pos] == col)
                unsafeAtomicAdd(a.values + pos, Kb[int64_t(u) * Nd + j]);
            else
                ++missing; // (Tpetra's sumIntoLocalValues skips entries outside the graph and reports how many it took)
        }
    }
    if (missing && a.n_missing)
        atomicAdd(a.n_missing, static_cast< unsigned long long >(missing));
}
// The scatter for element matrices in the TILED layout of the assembly kernel (device/assemble.hpp, TILED): blocks
// [u][u'][bx'][bz][bx][by][by'][bz'].  One wave per (element, row node b): for every row (b, u) the 4 n runs of n^2 doubles that
// hold the row (one per (u', bx')) are read -- consecutive lanes, consecutive addresses -- and laid down in LDS in the row's
// column order (b', u'), u' fastest: from there the lanes walk the row's consecutive entries exactly as assembledScatterKernel
// does (dense atomic requests), and the CSR position of an entry is searched once, for the first live row of the node (kept
// in LDS as an offset into the row), and re-used for the others after a check.
template < int U, int N1 >
__global__ __launch_bounds__(64) void assembledScatterTiledKernel(const ScatterArgs a)
{
    extern __shared__ double lds[];
    constexpr int   N2 = N1 * N1, NN = N2 * N1, Nd = NN * U;
    double* const   rowv = lds;                                      // [Nd] the row in column order
    int32_t* const  relv = reinterpret_cast< int32_t* >(lds + Nd);  // [Nd] position of the entry's column in the node's rows
    const int64_t   e    = blockIdx.x / NN;
    const int       b    = int(blockIdx.x - e * NN);
    const uint32_t* en   = a.elem_nodes + (a.first + e) * NN;
    const int       lane = threadIdx.x;
    const int64_t   nb   = int64_t(en[b]) * a.dpn;
    const int       bx = b % N1, by = (b / N1) % N1, bz = b / N2; // row node b = bx + N1 (by + N1 bz)
    if (a.F && a.rhs)
        for (int t = lane; t < a.n_rhs * U; t += 64) // (n_rhs * U may exceed the wave)
        {
            const int     r = t / U, u = t - r * U;
            const int64_t row = nb + a.field_inds[u];
            if (!(a.skip_dirichlet && a.dirichlet && a.dirichlet[row]))
                unsafeAtomicAdd(a.rhs + size_t(r) * a.ldr + row, a.F[(e * a.n_rhs + r) * Nd + b * U + u]);
        }
    if (!a.K || !a.values)
        return;
    const double* Ke       = a.K + e * int64_t(Nd) * Nd;
    const int64_t row_off  = ((int64_t(bz) * N1 + bx) * N1 + by) * N2; // + cx * NN * N2: start of the run (u', bx' = cx)
    unsigned      missing  = 0;
    bool          have_rel = false;
#pragma unroll 1
    for (int u = 0; u < U; ++u)
    {
        const int64_t row = nb + a.field_inds[u];
        if (a.skip_dirichlet && a.dirichlet && a.dirichlet[row]) // (wave-uniform)
            continue;
        __syncthreads(); // the previous row has been consumed
        // ---- the row's U N1 runs of N2 doubles (by', bz'; bz' fastest) -> LDS in column order (b', u'), u' fastest: entry i of
        // the concatenated runs, the lanes over i
        const double* Kr = Ke + int64_t(u) * U * NN * NN + row_off;
#pragma unroll 4
        for (int i = lane; i < Nd; i += 64)
        {
            const int run = i / N2, k = i - run * N2, up = run / N1, cx = run - up * N1, cy = k / N1, cz = k - cy * N1;
            rowv[((cz * N1 + cy) * N1 + cx) * U + up] = Kr[int64_t(up) * NN * NN + int64_t(cx) * NN * N2 + k];
        }
        __syncthreads();
        const int64_t rb = a.row_ptr[row], re = a.row_ptr[row + 1];
        for (int j = lane; j < Nd; j += 64)
        {
            const int     bp  = j / U;
            const int64_t col = int64_t(en[bp]) * a.dpn + a.field_inds[j - bp * U];
            if (a.skip_dirichlet && a.dirichlet && a.dirichlet[col])
                continue;
            int64_t pos;
            if (!have_rel)
            {
                pos     = lowerBound(a.col_ind, rb, re, col);
                relv[j] = int32_t(pos - rb);
            }
            else
            {
                pos = rb + relv[j];
                if (!(pos < re && a.col_ind[pos] == col))
                    pos = lowerBound(a.col_ind, rb, re, col);
            }
            if (pos < re && a.col_ind[pos] == col)
                unsafeAtomicAdd(a.values + pos, rowv[j]);
            else
                ++missing;
        }
        have_rel = true;
    }
    if (missing && a.n_missing)
        atomicAdd(a.n_missing, static_cast< unsigned long long >(missing));
}
template < int U >
int launchScatterTiled(const ScatterArgs& a, int N1, int64_t blocks, hipStream_t s)
{
    const size_t lds = size_t(a.NN) * U * (sizeof(double) + sizeof(int32_t));
    switch (N1)
    {
    case 2: hipLaunchKernelGGL((assembledScatterTiledKernel< U, 2 >), dim3(unsigned(blocks)), dim3(64), lds, s, a); break;
    case 3: hipLaunchKernelGGL((assembledScatterTiledKernel< U, 3 >), dim3(unsigned(blocks)), dim3(64), lds, s, a); break;
    case 4: hipLaunchKernelGGL((assembledScatterTiledKernel< U, 4 >), dim3(unsigned(blocks)), dim3(64), lds, s, a); break;
    case 5: hipLaunchKernelGGL((assembledScatterTiledKernel< U, 5 >), dim3(unsigned(blocks)), dim3(64), lds, s, a); break;
    case 6: hipLaunchKernelGGL((assembledScatterTiledKernel< U, 6 >), dim3(unsigned(blocks)), dim3(64), lds, s, a); break;
    case 7: hipLaunchKernelGGL((assembledScatterTiledKernel< U, 7 >), dim3(unsigned(blocks)), dim3(64), lds, s, a); break;
    case 8: hipLaunchKernelGGL((assembledScatterTiledKernel< U, 8 >), dim3(unsigned(blocks)), dim3(64), lds, s, a); break;
    default: return 1;
    }
    return 0;
}
} // namespace

// Tiled layout -> the reference's row-major K_e (AssembleLocalSystem.hpp:168-182).  The assembly kernels cannot store row-major
// efficiently -- a 64-byte line of K_e holds the entries of four unknown pairs (u, u') and two b_x', i.e. of four workgroups and two
// iterations: 3.9 x write traffic, 22 % full requests (profiles/r03_tcc_assembly_stored.txt) -- so the stored mode forms the tiled
// layout (coalesced) and THIS kernel turns it: one workgroup per (element, row node b) reads the U * U * n runs of n^2 doubles that
// belong to the node's U rows (392-byte runs at order 6), keeps them in LDS in read order and writes the U rows -- adjacent rows of
// the row-major matrix, one contiguous block of U * Nd doubles -- with consecutive lanes on consecutive addresses: whole 64-byte
// lines only.
template < int U, int N1 >
__global__ __launch_bounds__(256) void tiledToRowMajorKernel(const double* __restrict__ Kt, double* __restrict__ K)
{
    constexpr int N2 = N1 * N1, NN = N2 * N1, Nd = NN * U, TOTAL = U * U * N1 * N2; // (= U * Nd)
    extern __shared__ double rowbuf[];                                           // [u][u'][bx'][by' * N1 + bz']: read order
    const int64_t  e  = blockIdx.x / NN;
    const int      b  = int(blockIdx.x - e * NN);
    const int      bx = b % N1, by = (b / N1) % N1, bz = b / N2;
    const double*  Ke = Kt + e * int64_t(Nd) * Nd;
    // (eight independent loads in flight per thread before the first LDS store: the kernel is a pure memory mover)
    constexpr int NB = 8;
    for (int t0 = threadIdx.x; t0 < TOTAL; t0 += 256 * NB)
    {
        double v[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k)
        {
            const int t = t0 + 256 * k;
            if (t < TOTAL)
            {
                const int j = t % N2, r = t / N2, bxp = r % N1, uu = r / N1; // uu = u * U + u'
                v[k]        = Ke[((((int64_t(uu) * N1 + bxp) * N1 + bz) * N1 + bx) * N1 + by) * N2 + j];
            }
        }
#pragma unroll
        for (int k = 0; k < NB; ++k)
            if (t0 + 256 * k < TOTAL)
                rowbuf[t0 + 256 * k] = v[k];
    }
    __syncthreads();
    double* out = K + e * int64_t(Nd) * Nd + int64_t(b) * U * Nd; // rows (b, 0 .. U-1)
    for (int o = threadIdx.x; o < TOTAL; o += 256)
    {
        const int u = o / Nd, gj = o - u * Nd, up = gj % U, bp = gj / U;
        const int bxp = bp % N1, byp = (bp / N1) % N1, bzp = bp / N2;
        out[o]        = rowbuf[((u * U + up) * N1 + bxp) * N2 + byp * N1 + bzp];
    }
}
template < int U >
int launchTiledToRowMajorU(int N1, int64_t blocks, const double* Kt, double* K, hipStream_t s)
{
    const size_t lds = sizeof(double) * size_t(U) * U * N1 * N1 * N1;
    switch (N1)
    {
    case 2: hipLaunchKernelGGL((tiledToRowMajorKernel< U, 2 >), dim3(unsigned(blocks)), dim3(256), lds, s, Kt, K); break;
    case 3: hipLaunchKernelGGL((tiledToRowMajorKernel< U, 3 >), dim3(unsigned(blocks)), dim3(256), lds, s, Kt, K); break;
    case 4: hipLaunchKernelGGL((tiledToRowMajorKernel< U, 4 >), dim3(unsigned(blocks)), dim3(256), lds, s, Kt, K); break;
    case 5: hipLaunchKernelGGL((tiledToRowMajorKernel< U, 5 >), dim3(unsigned(blocks)), dim3(256), lds, s, Kt, K); break;
    case 6: hipLaunchKernelGGL((tiledToRowMajorKernel< U, 6 >), dim3(unsigned(blocks)), dim3(256), lds, s, Kt, K); break;
    case 7: hipLaunchKernelGGL((tiledToRowMajorKernel< U, 7 >), dim3(unsigned(blocks)), dim3(256), lds, s, Kt, K); break;
    case 8: hipLaunchKernelGGL((tiledToRowMajorKernel< U, 8 >), dim3(unsigned(blocks)), dim3(256), lds, s, Kt, K); break;
    default: return 1;
    }
    return 0;
}
// The x-major tiled layout -> the reference's row-major K_e, bitwise symmetric, in ONE pass over the matrix
// (AssembleLocalSystem.hpp:168-182: getSystem returns the lower triangle mirrored).  The assembly kernels form K[i][j] and K[j][i]
// in two summation orders (equal to rounding); only the lower triangle is read here and every entry is written twice, to (i, j)
// and to (j, i).  The layout that makes this a streaming kernel is the tiled one with the roles of x and z exchanged
// (ElemArgs::K_tiled == 2: the coefficient kernel hands the quadrature points and the reference derivatives over with x and z
// swapped, the assembly kernel is the same code): [u][u'][bz'][bx][bz][by][by'][bx'] -- a run of n^2 doubles is then the n^2 column
// nodes of one z'-slab in the order of the row-major matrix, for one u'.  One workgroup per (element, x-line l = (by, bz) of row
// nodes: n U adjacent rows); per z'-slab it reads the runs of the slab's column lines l' <= l (whole runs: every byte of the lower
// triangle once), keeps the n U x n^2 U block in LDS and writes
//   * the block itself: n U row segments of n^2 U contiguous doubles (1 568 bytes at order 6),
//   * its mirror image: n^2 U row segments of n U contiguous doubles (224 bytes); the line block l' == l mirrors itself.
// Segments start on 32-byte boundaries; the half lines at their ends are completed by the neighbouring x-line's workgroup, which
// runs on the same XCD (workgroups of one element share an index modulo 8) at about the same time.
template < int U, int N1 >
__global__ __launch_bounds__(256) void tiledXToRowMajorSymKernel(const double* __restrict__ Kt, double* __restrict__ K, int64_t count)
{
    constexpr int N2 = N1 * N1, NN = N2 * N1, Nd = NN * U, R = N1 * U, CW = N2 * U, LD = CW + 1;
    extern __shared__ double tile[]; // [R][LD]: row (bx, u) of the x-line, column (by', bx', u') of the slab
    const int64_t chunk = blockIdx.x / (8 * N2);
    const int     w     = int(blockIdx.x - chunk * (8 * N2));
    const int64_t e     = chunk * 8 + (w & 7);
    if (e >= count)
        return;
    const int     l  = N2 - 1 - (w >> 3); // (the lines with most columns first)
    const int     by = l % N1, bz = l / N1;
    const int     tid = threadIdx.x;
    const double* Te  = Kt + e * int64_t(Nd) * Nd;
    double*       Ke  = K + e * int64_t(Nd) * Nd;
    // (no run-time divisions: wave w takes the runs / rows w, w + 4, ..., lane x the entries x, x + 64, ... of a run / row)
    const int wv = tid >> 6, x = tid & 63;
    for (int bzp = 0; bzp <= bz; ++bzp)
    {
        const int run = bzp < bz ? N2 : (by + 1) * N1; // column nodes (by', bx') of the slab on lines l' <= l
        constexpr int NR = U * U * N1;                 // runs (u, u', bx) of this x-line in the slab
        constexpr int NB = 8;                          // independent loads in flight per thread
        const double* Ts = Te + (int64_t(bzp) * N1 * N1 + bz) * N1 * N2 + by * N2; // + ((uu * N1 * N1 + bx) * N1) * N1 * N2
        for (int r0 = wv; r0 < NR; r0 += 4 * NB)
            for (int j = x; j < run; j += 64)
            {
                double v[NB];
#pragma unroll
                for (int k = 0; k < NB; ++k)
                {
                    const int r = r0 + 4 * k;
                    if (r < NR)
                    {
                        const int bx = r % N1, uu = r / N1;
                        v[k]         = Ts[((int64_t(uu) * N1 * N1 + bx) * N1) * N1 * N2 + j];
                    }
                }
#pragma unroll
                for (int k = 0; k < NB; ++k)
                {
                    const int r = r0 + 4 * k;
                    if (r < NR)
                    {
                        const int bx = r % N1, uu = r / N1, u = uu / U, up = uu - u * U;
                        tile[(bx * U + u) * LD + j * U + up] = v[k];
                    }
                }
            }
        __syncthreads();
        const int ncol = run * U;
        for (int r = wv; r < R; r += 4)
        {
            double* const out = Ke + (int64_t(l) * R + r) * Nd + bzp * CW;
            for (int c = x; c < ncol; c += 64)
            {
                double v = tile[r * LD + c];
                if (bzp == bz)
                {
                    const int cc = c - by * R; // position within the line block l' == l: its upper part comes from the mirror position
                    if (cc > r)
                        v = tile[cc * LD + by * R + r];
                }
                out[c] = v;
            }
        }
        const int nm = (bzp < bz ? N2 : by * N1) * U; // rows of the mirror image (the line block l' == l has none)
        for (int o = tid; o < nm * R; o += 256)
        {
            const int rr = o / R, r = o - rr * R;
            Ke[(int64_t(bzp) * CW + rr) * Nd + l * R + r] = tile[r * LD + rr];
        }
        __syncthreads();
    }
}
template < int U, int N1 >
int launchTiledXOne(int64_t count, const double* Kt, double* K, hipStream_t s)
{
    constexpr size_t lds    = sizeof(double) * size_t(N1) * U * (N1 * N1 * U + 1);
    if constexpr (lds > 64 * 1024) // (order 7 with 4 unknowns; an attribute of the function on the CURRENT device: set per launch)
        if (hipFuncSetAttribute(reinterpret_cast< const void* >(&tiledXToRowMajorSymKernel< U, N1 >), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)) != hipSuccess)
            return 1;
    const int64_t blocks = ((count + 7) / 8) * 8 * N1 * N1;
    hipLaunchKernelGGL((tiledXToRowMajorSymKernel< U, N1 >), dim3(unsigned(blocks)), dim3(256), lds, s, Kt, K, count);
    return 0;
}
template < int U >
int launchTiledXU(int N1, int64_t count, const double* Kt, double* K, hipStream_t s)
{
    switch (N1)
    {
    case 2: return launchTiledXOne< U, 2 >(count, Kt, K, s);
    case 3: return launchTiledXOne< U, 3 >(count, Kt, K, s);
    case 4: return launchTiledXOne< U, 4 >(count, Kt, K, s);
    case 5: return launchTiledXOne< U, 5 >(count, Kt, K, s);
    case 6: return launchTiledXOne< U, 6 >(count, Kt, K, s);
    case 7: return launchTiledXOne< U, 7 >(count, Kt, K, s);
    case 8: return launchTiledXOne< U, 8 >(count, Kt, K, s);
    default: return 1;
    }
}
// `count` element matrices: x-major tiled (d_Kt, formed with ElemArgs::K_tiled == 2) -> row-major, bitwise symmetric (d_K), on stream s
int launchTiledXToRowMajorSym(int U, int N1, int64_t count, const double* d_Kt, double* d_K, hipStream_t s)
{
    const int64_t blocks = ((count + 7) / 8) * 8 * int64_t(N1) * N1;
    if (blocks > int64_t(0x7fffffff) || U < 1 || U > 4 || N1 < 2 || N1 > 8)
    {
        setError("x-major tiled -> row-major: shape (order %d, %d unknowns, %lld elements) not supported", N1 - 1, U, (long long)count);
        return -1;
    }
    int rc = 1;
    switch (U)
    {
    case 1: rc = launchTiledXU< 1 >(N1, count, d_Kt, d_K, s); break;
    case 2: rc = launchTiledXU< 2 >(N1, count, d_Kt, d_K, s); break;
    case 3: rc = launchTiledXU< 3 >(N1, count, d_Kt, d_K, s); break;
    case 4: rc = launchTiledXU< 4 >(N1, count, d_Kt, d_K, s); break;
    default: break;
    }
    if (rc || hipGetLastError() != hipSuccess)
    {
        setError("x-major tiled -> row-major kernel launch failed (order %d, %d unknowns)", N1 - 1, U);
        return -3;
    }
    return 0;
}
int launchTiledToRowMajor(int U, int N1, int64_t count, const double* d_Kt, double* d_K, hipStream_t s)
{
    const int64_t blocks = count * N1 * N1 * N1;
    if (blocks > int64_t(0x7fffffff) || U < 1 || U > 4 || size_t(U) * U * N1 * N1 * N1 * sizeof(double) > 64 * 1024)
    {
        setError("tiled -> row-major: shape (order %d, %d unknowns, %lld elements) not supported", N1 - 1, U, (long long)count);
        return -1;
    }
    int rc = 1;
    switch (U)
    {
    case 1: rc = launchTiledToRowMajorU< 1 >(N1, blocks, d_Kt, d_K, s); break;
    case 2: rc = launchTiledToRowMajorU< 2 >(N1, blocks, d_Kt, d_K, s); break;
    case 3: rc = launchTiledToRowMajorU< 3 >(N1, blocks, d_Kt, d_K, s); break;
    case 4: rc = launchTiledToRowMajorU< 4 >(N1, blocks, d_Kt, d_K, s); break;
    default: break;
    }
    if (rc || hipGetLastError() != hipSuccess)
    {
        setError("tiled -> row-major kernel launch failed (order %d, %d unknowns)", N1 - 1, U);
        return -3;
    }
    return 0;
}

// the launch of the batch scatter on `s` (shared by l3k_assembled_scatter and the pipelined l3k_assemble_global)
int launchAssembledScatter(l3k_mf* mf, int64_t first, int64_t count, const double* d_K, const double* d_F, const int64_t* d_row_ptr,
                           const int32_t* d_col_ind, double* d_values, double* d_rhs, size_t ldr, int skip_dirichlet,
                           unsigned long long* d_count, hipStream_t s, int tiled)
{
    const l3k_mesh* m  = mf->mesh;
    const int       N1 = m->order + 1, NN = N1 * N1 * N1;
    const int64_t   blocks = count * NN;
    if (blocks > int64_t(0x7fffffff))
    {
        setError("batch too large: %lld element rows in one launch", (long long)blocks);
        return -1;
    }
    ScatterArgs a{};
    a.elem_nodes     = m->elem_nodes.ptr;
    a.dirichlet      = m->dirichlet.ptr;
    a.K              = d_K;
    a.F              = d_F;
    a.row_ptr        = d_row_ptr;
    a.col_ind        = d_col_ind;
    a.values         = d_values;
    a.rhs            = d_rhs;
    a.ldr            = ldr;
    a.n_missing      = d_count;
    a.first          = first;
    a.count          = count;
    a.NN             = NN;
    a.dpn            = m->dofs_per_node;
    a.n_rhs          = mf->n_rhs;
    a.skip_dirichlet = skip_dirichlet;
    for (int u = 0; u < l3k::dev::max_unknowns; ++u)
        a.field_inds[u] = mf->field_inds[u];
    a.U_rt = mf->kp.n_unknowns;
    if (tiled)
    {
        int rc = 1;
        switch (mf->kp.n_unknowns)
        {
        case 1: rc = launchScatterTiled< 1 >(a, N1, blocks, s); break;
        case 2: rc = launchScatterTiled< 2 >(a, N1, blocks, s); break;
        case 3: rc = launchScatterTiled< 3 >(a, N1, blocks, s); break;
        case 4: rc = launchScatterTiled< 4 >(a, N1, blocks, s); break;
        default: break;
        }
        if (rc)
        {
            setError("tiled scatter: shape (order %d, %d unknowns) not instantiated", m->order, mf->kp.n_unknowns);
            return -1;
        }
    }
    else if (mf->ctx->tune.scatter_per_entry) // (the round-2 kernel, kept as the cross-check of the per-row-node one)
    {
        const int64_t rows = count * NN * mf->kp.n_unknowns;
        if (rows > int64_t(0x7fffffff))
        {
            setError("batch too large: %lld element rows in one launch", (long long)rows);
            return -1;
        }
        hipLaunchKernelGGL(assembledScatterPerEntryKernel, dim3(unsigned(rows)), dim3(64), 0, s, a);
    }
    else
        switch (mf->kp.n_unknowns)
        {
        case 1: hipLaunchKernelGGL(assembledScatterKernel< 1 >, dim3(unsigned(blocks)), dim3(64), 0, s, a); break;
        case 2: hipLaunchKernelGGL(assembledScatterKernel< 2 >, dim3(unsigned(blocks)), dim3(64), 0, s, a); break;
        case 3: hipLaunchKernelGGL(assembledScatterKernel< 3 >, dim3(unsigned(blocks)), dim3(64), 0, s, a); break;
        case 4: hipLaunchKernelGGL(assembledScatterKernel< 4 >, dim3(unsigned(blocks)), dim3(64), 0, s, a); break;
        case 5: hipLaunchKernelGGL(assembledScatterKernel< 5 >, dim3(unsigned(blocks)), dim3(64), 0, s, a); break;
        case 6: hipLaunchKernelGGL(assembledScatterKernel< 6 >, dim3(unsigned(blocks)), dim3(64), 0, s, a); break;
        case 7: hipLaunchKernelGGL(assembledScatterKernel< 7 >, dim3(unsigned(blocks)), dim3(64), 0, s, a); break;
        case 8: hipLaunchKernelGGL(assembledScatterKernel< 8 >, dim3(unsigned(blocks)), dim3(64), 0, s, a); break;
        default: setError("l3k_assembled_scatter: %d unknowns not supported (1..8)", mf->kp.n_unknowns); return -1;
        }
    L3K_HIP(hipGetLastError());
    return 0;
}

extern "C" {
int l3k_assembled_scatter(l3k_mf* mf, int64_t first, int64_t count, const double* d_K, const double* d_F, const int64_t* d_row_ptr,
                          const int32_t* d_col_ind, double* d_values, double* d_rhs, size_t ldr, int skip_dirichlet,
                          int64_t* n_missing)
{
    if (!mf)
    {
        setError("null mf");
        return -1;
    }
    const l3k_mesh* m = mf->mesh;
    if (first < 0 || count < 0 || first + count > m->n_elems)
    {
        setError("element range [%lld, %lld) outside [0, %lld)", (long long)first, (long long)(first + count), (long long)m->n_elems);
        return -1;
    }
    if ((d_K != nullptr) != (d_values != nullptr) || (d_K && (!d_row_ptr || !d_col_ind)))
    {
        setError("l3k_assembled_scatter: K needs values, row_ptr and col_ind (and the other way round)");
        return -1;
    }
    if ((d_F != nullptr) != (d_rhs != nullptr))
    {
        setError("l3k_assembled_scatter: F and rhs go together");
        return -1;
    }
    const int64_t n_local_dofs = (m->n_owned_nodes + m->n_ghost_nodes) * m->dofs_per_node;
    if (d_rhs && ldr < size_t(n_local_dofs))
    {
        setError("rhs leading dimension smaller than the number of local dofs");
        return -1;
    }
    if (n_missing)
        *n_missing = 0;
    if (count == 0 || (!d_K && !d_F))
        return 0;
    L3K_HIP(hipSetDevice(mf->ctx->device));
    hipStream_t         s       = mf->ctx->stream;
    unsigned long long* d_count = nullptr; // (a counter of the context: no allocation per call)
    if (n_missing)
    {
        d_count = mf->ctx->missCounter();
        L3K_HIP(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), s));
    }
    if (int rc = launchAssembledScatter(mf, first, count, d_K, d_F, d_row_ptr, d_col_ind, d_values, d_rhs, ldr, skip_dirichlet, d_count, s, 0))
        return rc;
    if (n_missing)
    {
        unsigned long long h = 0;
        L3K_HIP(hipMemcpyAsync(&h, d_count, sizeof h, hipMemcpyDeviceToHost, s));
        L3K_HIP(hipStreamSynchronize(s));
        *n_missing = int64_t(h);
    }
    return 0;
}
} // extern "C"
