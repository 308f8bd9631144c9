// Dense LU / inverse / symmetric eigensolver for setup-time operator
// construction. See include/blitzdg/DenseLinAlg.hpp for the reference call
// sites these stand in for.
#include "blitzdg/DenseLinAlg.hpp"
#include <algorithm>
#include <cmath>
#include <numeric>
#include <stdexcept>
#include <vector>

namespace blitzdg {
namespace {

// In-place LU with partial pivoting on a row-major n x n copy of A.
// Returns false when a pivot is exactly zero.
bool luFactor(std::vector<double>& a, std::vector<int>& piv, int n) {
    piv.resize(n);
    for (int c = 0; c < n; ++c) {
        int p = c;
        double best = std::fabs(a[c * n + c]);
        for (int r = c + 1; r < n; ++r) {
            const double v = std::fabs(a[r * n + c]);
            if (v > best) { best = v; p = r; }
        }
        piv[c] = p;
        if (best == 0.0) return false;
        if (p != c)
            for (int j = 0; j < n; ++j) std::swap(a[c * n + j], a[p * n + j]);
        const double inv = 1.0 / a[c * n + c];
        for (int r = c + 1; r < n; ++r) {
            const double m = a[r * n + c] * inv;
            a[r * n + c] = m;
            if (m != 0.0)
                for (int j = c + 1; j < n; ++j) a[r * n + j] -= m * a[c * n + j];
        }
    }
    return true;
}

// Solves LU x = P b for nrhs right-hand sides stored row-major in b (n x nrhs).
void luSolve(const std::vector<double>& lu, const std::vector<int>& piv, int n,
             std::vector<double>& b, int nrhs) {
    for (int c = 0; c < n; ++c)
        if (piv[c] != c)
            for (int j = 0; j < nrhs; ++j) std::swap(b[c * nrhs + j], b[piv[c] * nrhs + j]);
    for (int r = 1; r < n; ++r)
        for (int c = 0; c < r; ++c) {
            const double m = lu[r * n + c];
            if (m != 0.0)
                for (int j = 0; j < nrhs; ++j) b[r * nrhs + j] -= m * b[c * nrhs + j];
        }
    for (int r = n - 1; r >= 0; --r) {
        for (int c = r + 1; c < n; ++c) {
            const double m = lu[r * n + c];
            if (m != 0.0)
                for (int j = 0; j < nrhs; ++j) b[r * nrhs + j] -= m * b[c * nrhs + j];
        }
        const double inv = 1.0 / lu[r * n + r];
        for (int j = 0; j < nrhs; ++j) b[r * nrhs + j] *= inv;
    }
}

} // namespace

void DirectSolver::solve(const real_matrix_type& A, const real_matrix_type& B, real_matrix_type& X) const {
    const int n = A.rows();
    if (A.cols() != n || B.rows() != n)
        throw std::runtime_error("DirectSolver::solve: dimension mismatch");
    const int nrhs = B.cols();
    std::vector<double> lu(A.data(), A.data() + static_cast<std::size_t>(n) * n);
    std::vector<int> piv;
    if (!luFactor(lu, piv, n))
        throw std::runtime_error("DirectSolver::solve: matrix is singular");
    std::vector<double> x(B.data(), B.data() + static_cast<std::size_t>(n) * nrhs);
    luSolve(lu, piv, n, x, nrhs);
    // One step of iterative refinement in fp64 (the reference's dsgesv_ also
    // refines to fp64 backward-error level).
    std::vector<double> r(static_cast<std::size_t>(n) * nrhs);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < nrhs; ++j) {
            long double s = B(i, j);
            for (int k = 0; k < n; ++k) s -= static_cast<long double>(A(i, k)) * x[k * nrhs + j];
            r[i * nrhs + j] = static_cast<double>(s);
        }
    luSolve(lu, piv, n, r, nrhs);
    if (X.rows() != n || X.cols() != nrhs) X.resize(n, nrhs);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < nrhs; ++j) X(i, j) = x[i * nrhs + j] + r[i * nrhs + j];
}

void DenseMatrixInverter::computeInverse(const real_matrix_type& A, real_matrix_type& Ainv) const {
    const int n = A.rows();
    if (A.cols() != n) throw std::runtime_error("DenseMatrixInverter: matrix is not square");
    real_matrix_type I(n, n);
    for (int i = 0; i < n; ++i) I(i, i) = 1.0;
    try {
        DirectSolver{}.solve(A, I, Ainv);
    } catch (const std::runtime_error&) {
        throw std::runtime_error("Unable to compute inverse: matrix is singular");
    }
}

void EigenSolver::solve(const real_matrix_type& A, real_vector_type& eigenvalues, real_matrix_type& eigenvectors) const {
    const int n = A.rows();
    if (A.cols() != n) throw std::runtime_error("EigenSolver: matrix is not square");
    std::vector<double> a(A.data(), A.data() + static_cast<std::size_t>(n) * n);
    std::vector<double> v(static_cast<std::size_t>(n) * n, 0.0);
    for (int i = 0; i < n; ++i) v[i * n + i] = 1.0;

    // Cyclic Jacobi rotations; converges quadratically, accurate to a few ulp
    // for the small symmetric tridiagonal Golub-Welsch matrices it is used on.
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) off += a[p * n + q] * a[p * n + q];
        if (off == 0.0) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = a[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = a[k * n + p], akq = a[k * n + q];
                    a[k * n + p] = c * akp - s * akq;
                    a[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = a[p * n + k], aqk = a[q * n + k];
                    a[p * n + k] = c * apk - s * aqk;
                    a[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = v[k * n + p], vkq = v[k * n + q];
                    v[k * n + p] = c * vkp - s * vkq;
                    v[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int i, int j) { return a[i * n + i] < a[j * n + j]; });
    if (eigenvalues.size() != n) eigenvalues.resize(n);
    if (eigenvectors.rows() != n || eigenvectors.cols() != n) eigenvectors.resize(n, n);
    for (int k = 0; k < n; ++k) {
        eigenvalues(k) = a[order[k] * n + order[k]];
        for (int i = 0; i < n; ++i) eigenvectors(i, k) = v[i * n + order[k]];
    }
}

void DenseCholeskyFactorizer::computeCholesky(const real_matrix_type& A, real_matrix_type& R) const {
    const index_type n = A.rows();
    if (A.cols() != n) throw std::runtime_error("computeCholesky: matrix is not square");
    if (R.rows() != n || R.cols() != n) R.resize(n, n);
    else R.fill(0.0);
    // row-oriented (Cholesky-Banachiewicz on the upper factor): R(i,i) = sqrt(A(i,i) - sum_k R(k,i)^2),
    // R(i,j) = (A(i,j) - sum_k R(k,i) R(k,j)) / R(i,i) for j > i
    for (index_type i = 0; i < n; ++i) {
        real_type d = A(i, i);
        for (index_type k = 0; k < i; ++k) d -= R(k, i) * R(k, i);
        if (!(d > 0.0))
            throw std::runtime_error("The leading minor of order " + std::to_string(i + 1) +
                                     " is not positive definite. The Cholesky factorization could not be completed.");
        const real_type rii = std::sqrt(d);
        R(i, i) = rii;
        for (index_type j = i + 1; j < n; ++j) {
            real_type v = A(i, j);
            for (index_type k = 0; k < i; ++k) v -= R(k, i) * R(k, j);
            R(i, j) = v / rii;
        }
    }
}

} // namespace blitzdg
