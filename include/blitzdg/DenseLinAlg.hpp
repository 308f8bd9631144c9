// Small dense linear algebra for operator construction (Np x Np, Np <= ~66).
//
// The reference delegates these to LAPACK (src/DirectSolver.cpp:24-63 dsgesv_,
// src/DenseMatrixinverter.cpp:26-60 dgetrf_/dgetri_, src/EigenSolver.cpp:33-75
// dsyevd_). LAPACK is not in this image, and these run once at setup on tiny
// matrices, so they are implemented directly: LU with partial pivoting in fp64
// and a cyclic-Jacobi symmetric eigensolver. Class and method names follow the
// reference so call sites read the same.
#pragma once
#include "Types.hpp"

namespace blitzdg {

class DirectSolver {
public:
    /// Solves A X = B (A: n x n, B and X: n x nrhs). A and B are not modified.
    void solve(const real_matrix_type& A, const real_matrix_type& B, real_matrix_type& X) const;
};

class DenseMatrixInverter {
public:
    /// Ainv = A^{-1}; throws std::runtime_error if A is singular.
    void computeInverse(const real_matrix_type& A, real_matrix_type& Ainv) const;
};

class EigenSolver {
public:
    /// Symmetric eigenproblem. Eigenvalues ascending; eigenvector k is COLUMN k of
    /// `eigenvectors`, unit 2-norm (as src/EigenSolver.cpp:67 unpacks dsyevd_).
    void solve(const real_matrix_type& A, real_vector_type& eigenvalues, real_matrix_type& eigenvectors) const;
};

class DenseCholeskyFactorizer {
public:
    /// R = upper triangular factor with A = R^T R (the reference calls dpotrf_('U'),
    /// src/DenseCholeskyFactorizer.cpp:23-53, and zeroes the strict lower triangle); throws
    /// std::runtime_error when a leading minor is not positive definite.
    void computeCholesky(const real_matrix_type& A, real_matrix_type& R) const;
};

} // namespace blitzdg
