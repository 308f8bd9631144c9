// TriangleCubatureRules: cubature on the reference triangle {r,s >= -1, r+s <= 0} (area 2), exact
// for polynomials of total degree NCubature. Interface of the reference's
// include/TriangleCubatureRules.hpp:1808-1830 (NumCubaturePoints / rCoord / sCoord / weights).
//
// The reference tabulates symmetric rules (1831 lines of 15-digit constants, orders 1..28, then a
// 225-point fallback). Those constants are not reproduced here: this class COMPUTES a rule of the
// requested degree instead -- the conical (collapsed-coordinate) product of an n-point
// Gauss-Legendre rule in a with an n-point Gauss-Jacobi(1,0) rule in b, n = ceil((NCubature+1)/2),
// mapped by r = (1+a)(1-b)/2 - 1, s = b. It is exact to degree 2n-1 >= NCubature with positive
// weights and all points interior; it has n^2 points, i.e. MORE points than the tabulated rule of the
// same degree (49 against 36 at degree 12), so NumCubaturePoints differs from the reference's while
// every integral the rule is exact for -- in particular the cubature mass matrix -- agrees.
#pragma once
#include "JacobiBuilders.hpp"
#include "Types.hpp"

namespace blitzdg {

class TriangleCubatureRules {
public:
    TriangleCubatureRules() = default;
    explicit TriangleCubatureRules(index_type NCubature);
    index_type NCubature() const { return NCubature_; }
    index_type NumCubaturePoints() const { return r_.size(); }
    real_vector_type rCoord() const { return r_; }
    real_vector_type sCoord() const { return s_; }
    real_vector_type weights() const { return w_; }
private:
    index_type NCubature_ = 0;
    real_vector_type r_, s_, w_;
};

} // namespace blitzdg
