// Orthonormal Jacobi polynomials, their derivatives, Gauss-Jacobi and
// Gauss-Lobatto points; 1-D Legendre Vandermonde matrices.
// Same API as the reference's include/JacobiBuilders.hpp:15-77 and
// include/VandermondeBuilders.hpp:26-73 (algorithms: src/JacobiBuilders.cpp).
#pragma once
#include "DenseLinAlg.hpp"
#include "Types.hpp"

namespace blitzdg {

class JacobiBuilders {
public:
    /// p = P_N^{(alpha,beta)}(x), orthonormal w.r.t. (1-x)^alpha (1+x)^beta.
    void computeJacobiPolynomial(const real_vector_type& x, real_type alpha, real_type beta,
                                 index_type N, real_vector_type& p) const;
    /// (N+1)-point Gauss-Jacobi rule (Golub-Welsch).
    void computeJacobiQuadWeights(real_type alpha, real_type beta, index_type N,
                                  real_vector_type& x, real_vector_type& w) const;
    /// N+1 Gauss-Lobatto points (sic: the reference spells it "Lobotto").
    void computeGaussLobottoPoints(real_type alpha, real_type beta, index_type N,
                                   real_vector_type& x) const;
    /// dp = d/dx P_N^{(alpha,beta)}(x).
    void computeGradJacobi(const real_vector_type& x, real_type alpha, real_type beta,
                           index_type N, real_vector_type& dp) const;
private:
    EigenSolver EigSolver;
};

class VandermondeBuilders {
public:
    /// V(i,j) = P_j^{(0,0)}(r_i); optionally Vinv = V^{-1}.
    void computeVandermondeMatrix(const real_vector_type& r, real_matrix_type& V,
                                  real_matrix_type& Vinv, bool includeInverse = true) const;
    /// DVr(i,j) = d/dr P_j^{(0,0)}(r_i).
    void computeGradVandermonde(const real_vector_type& r, real_matrix_type& DVr) const;
private:
    JacobiBuilders Jacobi;
    DenseMatrixInverter Inverter;
};

} // namespace blitzdg
