// tables.hpp -- 1-D spectral-element tables (host).  Product code: independent of oracle/.
#ifndef L3K_HOST_TABLES_HPP
#define L3K_HOST_TABLES_HPP
#include <vector>

namespace l3k::host
{
// Gauss-Lobatto-Legendre abscissae on [-1,1], ascending (math::getLobattoRuleAbsc, math/LobattoRuleAbsc.hpp:30-35)
std::vector< double > gllNodes(int n);
// Gauss-Legendre rule, ascending (quad::getReferenceQuadrature, quad/ReferenceQuadrature.hpp:24-51)
void glRule(int nq, std::vector< double >& x, std::vector< double >& w);
// Lagrange basis on `nodes` and its derivative at x (basisfun/ReferenceBasisFunction.hpp:28-72)
void lagrange(const std::vector< double >& nodes, double x, double* vals, double* ders);
// I[b][q] = phi_b(x_q), D[b][q] = phi_b'(x_q), row-major (p+1) x nq (algsys/SumFactorization.hpp:25-65)
void basis1d(int p, int nq, std::vector< double >& I, std::vector< double >& D);
// collocation derivative on the Gauss points: C[q'][q] = l_q''(x_q) for the Lagrange basis l on the nq Gauss points
std::vector< double > collocDeriv(int nq);
void evenOddTables(const std::vector< double >& W, int nin, int nout, bool anti, std::vector< double >& We,
                   std::vector< double >& Wo);
// device table block in dev::TableLayout order: I | C | qw | qx | D | gll | even-odd tables of I, C, I^T, C^T
std::vector< double > deviceTableBlock(int p, int nq);
} // namespace l3k::host
#endif
