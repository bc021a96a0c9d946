// tables.cpp -- see tables.hpp.  Well-conditioned generators (Newton on Legendre recurrences, barycentric Lagrange) in
// long double; the reference's route (companion-matrix eigenvalues, monomial coefficients, Golub-Welsch:
// math/Polynomial.hpp:98-122, math/LagrangeInterpolation.hpp:13-41, math/ComputeGaussRule.hpp:26-60) yields the same
// numbers to ~1e-14 at p <= 6.
#include "tables.hpp"

#include <cmath>

namespace l3k::host
{
namespace
{
using ld = long double;
struct Leg
{
    ld P, dP;
};
Leg legendre(int n, ld x)
{
    if (n == 0)
        return {1, 0};
    ld pm = 1, pc = x;
    for (int k = 2; k <= n; ++k)
    {
        const ld pn = ((2 * k - 1) * x * pc - (k - 1) * pm) / k;
        pm          = pc;
        pc          = pn;
    }
    return {pc, n * (x * pc - pm) / (x * x - 1)};
}
const ld pi_l = 3.14159265358979323846264338327950288L;
} // namespace

std::vector< double > gllNodes(int n)
{
    std::vector< double > x(n);
    const int             N = n - 1;
    x.front()               = -1.;
    x.back()                = 1.;
    for (int k = 1; k < N; ++k)
    {
        ld xk = -std::cos(pi_l * k / N);
        for (int it = 0; it < 64; ++it)
        {
            const auto [P, dP] = legendre(N, xk);
            const ld d2P       = (2 * xk * dP - ld(N) * (N + 1) * P) / (1 - xk * xk);
            const ld step      = dP / d2P;
            xk -= step;
            if (std::fabs(step) < 1e-19L)
                break;
        }
        x[k] = static_cast< double >(xk);
    }
    for (int k = 0; k < n / 2; ++k)
    {
        const double a = .5 * (x[n - 1 - k] - x[k]);
        x[k]           = -a;
        x[n - 1 - k]   = a;
    }
    if (n % 2)
        x[n / 2] = 0.;
    return x;
}

void glRule(int nq, std::vector< double >& x, std::vector< double >& w)
{
    x.assign(nq, 0.);
    w.assign(nq, 0.);
    for (int i = 0; i < nq; ++i)
    {
        ld xi = -std::cos(pi_l * (i + .75L) / (nq + .5L));
        for (int it = 0; it < 64; ++it)
        {
            const auto [P, dP] = legendre(nq, xi);
            const ld step      = P / dP;
            xi -= step;
            if (std::fabs(step) < 1e-19L)
                break;
        }
        const auto [P, dP] = legendre(nq, xi);
        x[i]               = static_cast< double >(xi);
        w[i]               = static_cast< double >(2 / ((1 - xi * xi) * dP * dP));
    }
    if (nq % 2)
        x[nq / 2] = 0.;
}

void lagrange(const std::vector< double >& nodes, double x, double* vals, double* ders)
{
    const int n = static_cast< int >(nodes.size());
    for (int b = 0; b < n; ++b)
    {
        ld den = 1, val = 1, der = 0;
        for (int j = 0; j < n; ++j)
            if (j != b)
            {
                den *= ld(nodes[b]) - nodes[j];
                val *= ld(x) - nodes[j];
            }
        for (int k = 0; k < n; ++k)
        {
            if (k == b)
                continue;
            ld pr = 1;
            for (int j = 0; j < n; ++j)
                if (j != b && j != k)
                    pr *= ld(x) - nodes[j];
            der += pr;
        }
        vals[b] = static_cast< double >(val / den);
        if (ders)
            ders[b] = static_cast< double >(der / den);
    }
}

void basis1d(int p, int nq, std::vector< double >& I, std::vector< double >& D)
{
    const int             n   = p + 1;
    const auto            gll = gllNodes(n);
    std::vector< double > qx, qw, v(n), d(n);
    glRule(nq, qx, qw);
    I.assign(size_t(n) * nq, 0.);
    D.assign(size_t(n) * nq, 0.);
    for (int q = 0; q < nq; ++q)
    {
        lagrange(gll, qx[q], v.data(), d.data());
        for (int b = 0; b < n; ++b)
        {
            I[b * nq + q] = v[b];
            D[b * nq + q] = d[b];
        }
    }
}

std::vector< double > collocDeriv(int nq)
{
    std::vector< double > qx, qw, v(nq), d(nq), C(size_t(nq) * nq);
    glRule(nq, qx, qw);
    for (int q = 0; q < nq; ++q)
    {
        lagrange(qx, qx[q], v.data(), d.data());
        for (int qp = 0; qp < nq; ++qp)
            C[qp * nq + q] = d[qp];
    }
    return C;
}

// Even-odd split of W (nin x nout, row-major) with W[nin-1-b][nout-1-q] = s*W[b][q]:
//   out[q]        = A + B,   out[nout-1-q] = s*(A - B),   A = sum_r e[r]*We[r][q],  B = sum_r o[r]*Wo[r][q],
//   e[r] = in[r] + in[nin-1-r], o[r] = in[r] - in[nin-1-r] (r < nin/2), e[nin/2] = in[nin/2] for odd nin.
// Same decomposition as algsys/SumFactorization.hpp:88-203 (psi+ / psi-), with the 1/2 folded into the tables.
void evenOddTables(const std::vector< double >& W, int nin, int nout, bool anti, std::vector< double >& We,
                   std::vector< double >& Wo)
{
    const int hi = nin / 2, ho = nout / 2, ri = (nin + 1) / 2, ro = (nout + 1) / 2;
    We.assign(size_t(ri) * ro, 0.);
    Wo.assign(size_t(ri) * ro, 0.);
    for (int q = 0; q < ro; ++q)
    {
        const bool mid_col = (nout % 2) && q == ho;
        for (int r = 0; r < hi; ++r)
        {
            const double a = W[r * nout + q], b = W[(nin - 1 - r) * nout + q];
            if (mid_col) // out[mid] = sum_r e[r]*W[r][mid] (sym) or sum_r o[r]*W[r][mid] (anti)
                (anti ? Wo : We)[r * ro + q] = a;
            else
            {
                We[r * ro + q] = .5 * (a + b);
                Wo[r * ro + q] = .5 * (a - b);
            }
        }
        if (nin % 2)
            We[hi * ro + q] = (mid_col && anti) ? 0. : W[hi * nout + q];
    }
}

std::vector< double > deviceTableBlock(int p, int nq)
{
    std::vector< double > I, D, qx, qw;
    basis1d(p, nq, I, D);
    glRule(nq, qx, qw);
    const auto            C   = collocDeriv(nq);
    const auto            gll = gllNodes(p + 1);
    std::vector< double > out;
    out.insert(out.end(), I.begin(), I.end());
    out.insert(out.end(), C.begin(), C.end());
    out.insert(out.end(), qw.begin(), qw.end());
    out.insert(out.end(), qx.begin(), qx.end());
    out.insert(out.end(), D.begin(), D.end());
    out.insert(out.end(), gll.begin(), gll.end());
    const int             n = p + 1;
    std::vector< double > It(size_t(nq) * n), Ct(size_t(nq) * nq), We, Wo;
    for (int b = 0; b < n; ++b)
        for (int q = 0; q < nq; ++q)
            It[q * n + b] = I[b * nq + q];
    for (int a = 0; a < nq; ++a)
        for (int q = 0; q < nq; ++q)
            Ct[q * nq + a] = C[a * nq + q];
    auto append = [&](const std::vector< double >& W, int nin, int nout, bool anti) {
        evenOddTables(W, nin, nout, anti, We, Wo);
        out.insert(out.end(), We.begin(), We.end());
        out.insert(out.end(), Wo.begin(), Wo.end());
    };
    append(I, n, nq, false);
    append(C, nq, nq, true);
    append(It, nq, n, false);
    append(Ct, nq, nq, true);
    // elementwise product tables for the sum-factorised diagonal (device/diag.hpp)
    for (int which = 0; which < 3; ++which)
        for (size_t i = 0; i < I.size(); ++i)
            out.push_back(which == 0 ? I[i] * I[i] : (which == 1 ? I[i] * D[i] : D[i] * D[i]));
    // derivative of the basis at the two ends of the reference interval (normal derivative on an element side)
    std::vector< double > v(n), d(n);
    for (double x : {-1., 1.})
    {
        lagrange(gll, x, v.data(), d.data());
        out.insert(out.end(), d.begin(), d.end());
    }
    // derivative of the basis at the nodes themselves, [b][q] = phi_b'(gll_q) (computeValuesAtNodes)
    std::vector< double > dg(size_t(n) * n);
    for (int q = 0; q < n; ++q)
    {
        lagrange(gll, gll[q], v.data(), d.data());
        for (int b = 0; b < n; ++b)
            dg[size_t(b) * n + q] = d[b];
    }
    out.insert(out.end(), dg.begin(), dg.end());
    // even-odd tables of D^T (the assembly kernel's last contraction; at the end: the earlier offsets stay what they are)
    std::vector< double > Dt(size_t(nq) * n);
    for (int b = 0; b < n; ++b)
        for (int q = 0; q < nq; ++q)
            Dt[q * n + b] = D[b * nq + q];
    append(Dt, nq, n, true);
    return out;
}
} // namespace l3k::host
