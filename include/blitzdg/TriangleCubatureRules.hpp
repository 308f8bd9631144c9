// TriangleCubatureRules: cubature on the reference triangle {r,s >= -1, r+s <= 0} (area 2), exact
// for polynomials of total degree NCubature. Interface of the reference's
// include/TriangleCubatureRules.hpp:1808-1830 (NumCubaturePoints / rCoord / sCoord / weights).
//
// Degrees 1..28 are the reference's tabulated symmetric rules (:26-1804, the published Cubature2D tables: 1, 3, 6,
// 6, 7, 12, 15, 16, 19, 25, 28, 36, ... 145, 225 points), point for point in the reference's order, so that
// NumCubaturePoints and every table of a CubatureContext2D are what blitzdg hands its caller (36 points at degree
// 12: src/test/TriangleNodesProvisionerTests.cpp:504). The constants live in
// blitzdg_amd/csrc/host/triangle_cubature_table.inc. One deviation: the reference's own 6-point rule of degrees
// 3 and 4 is damaged (four commas missing, :35-40: three entries are differences of two literals and three are
// never assigned); the intended literals are used here.
//
// Beyond degree 28 the reference indexes past its table. Here a rule of the requested degree is COMPUTED instead:
// the conical (collapsed-coordinate) product of an n-point Gauss-Legendre rule in a with an n-point
// Gauss-Jacobi(1,0) rule in b, n = ceil((NCubature+1)/2), mapped by r = (1+a)(1-b)/2 - 1, s = b; exact to degree
// 2n-1 >= NCubature, positive weights, all points interior (conical(degree) below builds it for any degree).
#pragma once
#include "JacobiBuilders.hpp"
#include "Types.hpp"

namespace blitzdg {

class TriangleCubatureRules {
public:
    TriangleCubatureRules() = default;
    explicit TriangleCubatureRules(index_type NCubature);
    // the computed conical-product rule of the given degree (what degrees > 28 get)
    static TriangleCubatureRules conical(index_type NCubature);
    static constexpr index_type NumPreComputed = 28;
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
