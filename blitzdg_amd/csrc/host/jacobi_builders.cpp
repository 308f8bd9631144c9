// Jacobi polynomial machinery (setup path). Follows the algorithms of the
// reference's src/JacobiBuilders.cpp:18-127 and
// include/VandermondeBuilders.hpp:46-73 (three-term recurrence in the
// orthonormal normalisation, Golub-Welsch for the Gauss-Jacobi points).
#include "blitzdg/JacobiBuilders.hpp"
#include <cmath>
#include <limits>
#include <vector>

namespace blitzdg {

namespace {
// Norm^2 of P_0^{(a,b)}: int (1-x)^a (1+x)^b dx.  (src/JacobiBuilders.cpp:25)
inline double gammaZero(double a, double b) {
    return std::pow(2.0, a + b + 1) / (a + b + 1) * std::tgamma(a + 1) * std::tgamma(b + 1) /
           std::tgamma(a + b + 1);
}
// Off-diagonal recurrence coefficient a_{i+1}.  (src/JacobiBuilders.cpp:46, :73)
inline double recurA(int i, double a, double b) {
    const double h1 = 2.0 * i + a + b;
    return 2.0 / (h1 + 2) *
           std::sqrt((i + 1) * (i + 1 + a + b) * (i + 1 + a) * (i + 1 + b) / (h1 + 1) / (h1 + 3));
}
} // namespace

void JacobiBuilders::computeJacobiPolynomial(const real_vector_type& x, real_type alpha, real_type beta,
                                             index_type N, real_vector_type& p) const {
    const int n = x.size();
    if (p.size() != n) p.resize(n);
    const double gamma0 = gammaZero(alpha, beta);
    const double p0 = 1 / std::sqrt(gamma0);
    if (N == 0) {
        for (int k = 0; k < n; ++k) p(k) = p0;
        return;
    }
    const double gamma1 = (alpha + 1) * (beta + 1) / (alpha + beta + 3) * gamma0;
    const double s1 = std::sqrt(gamma1);
    std::vector<double> prev(n, p0), cur(n);
    for (int k = 0; k < n; ++k) cur[k] = ((alpha + beta + 2) * x(k) / 2 + (alpha - beta) / 2) / s1;
    if (N > 1) {
        double aold = 2 / (2 + alpha + beta) * std::sqrt((alpha + 1) * (beta + 1) / (alpha + beta + 3));
        for (int i = 1; i <= N - 1; ++i) {
            const double h1 = 2 * i + alpha + beta;
            const double anew = recurA(i, alpha, beta);
            const double bnew = -(alpha * alpha - beta * beta) / h1 / (h1 + 2);
            for (int k = 0; k < n; ++k) {
                const double next = 1 / anew * (-aold * prev[k] + (x(k) - bnew) * cur[k]);
                prev[k] = cur[k];
                cur[k] = next;
            }
            aold = anew;
        }
    }
    for (int k = 0; k < n; ++k) p(k) = cur[k];
}

void JacobiBuilders::computeJacobiQuadWeights(real_type alpha, real_type beta, index_type N,
                                              real_vector_type& x, real_vector_type& w) const {
    if (x.size() != N + 1) x.resize(N + 1);
    if (w.size() != N + 1) w.resize(N + 1);
    if (N == 0) {
        x(0) = -(alpha - beta) / (alpha + beta + 2);
        w(0) = 2.0;
        return;
    }
    const double eps = std::numeric_limits<double>::epsilon();
    // Symmetric tridiagonal Jacobi matrix of the recurrence.
    real_matrix_type J(N + 1, N + 1);
    for (int i = 0; i <= N; ++i) {
        const double h1 = 2.0 * i + alpha + beta;
        J(i, i) = -0.5 * (alpha * alpha - beta * beta) / (h1 + 2.) / h1 * 2.0; // diag of J+J^T
        if (i < N) J(i, i + 1) = J(i + 1, i) = recurA(i, alpha, beta);
    }
    if ((alpha + beta) < 10 * eps) J(0, 0) = 0.0;
    real_matrix_type vecs(N + 1, N + 1);
    EigSolver.solve(J, x, vecs);
    const double gamma0 = gammaZero(alpha, beta);
    for (int k = 0; k <= N; ++k) w(k) = vecs(0, k) * vecs(0, k) * gamma0;
}

void JacobiBuilders::computeGaussLobottoPoints(real_type alpha, real_type beta, index_type N,
                                               real_vector_type& x) const {
    if (x.size() != N + 1) x.resize(N + 1);
    x(0) = -1.0;
    x(N) = 1.0;
    if (N == 1) return;
    real_vector_type xg(N - 1), wg(N - 1);
    computeJacobiQuadWeights(alpha + 1., beta + 1., N - 2, xg, wg);
    for (int i = 1; i < N; ++i) x(i) = xg(i - 1);
}

void JacobiBuilders::computeGradJacobi(const real_vector_type& x, real_type alpha, real_type beta,
                                       index_type N, real_vector_type& dp) const {
    const int n = x.size();
    if (dp.size() != n) dp.resize(n);
    if (N == 0) {
        dp.fill(0.0);
        return;
    }
    real_vector_type p(n);
    computeJacobiPolynomial(x, alpha + 1, beta + 1, N - 1, p);
    const double scale = std::sqrt(N * (N + alpha + beta + 1));
    for (int k = 0; k < n; ++k) dp(k) = scale * p(k);
}

void VandermondeBuilders::computeVandermondeMatrix(const real_vector_type& r, real_matrix_type& V,
                                                   real_matrix_type& Vinv, bool includeInverse) const {
    const int ncols = V.cols(), nrows = V.rows();
    real_vector_type p(nrows);
    for (int j = 0; j < ncols; ++j) {
        Jacobi.computeJacobiPolynomial(r, 0.0, 0.0, j, p);
        for (int i = 0; i < nrows; ++i) V(i, j) = p(i);
    }
    if (includeInverse) Inverter.computeInverse(V, Vinv);
}

void VandermondeBuilders::computeGradVandermonde(const real_vector_type& r, real_matrix_type& DVr) const {
    const int n = r.size();
    real_vector_type dp(n);
    for (int j = 0; j < n; ++j) {
        Jacobi.computeGradJacobi(r, 0.0, 0.0, j, dp);
        for (int i = 0; i < n; ++i) DVr(i, j) = dp(i);
    }
}

} // namespace blitzdg
