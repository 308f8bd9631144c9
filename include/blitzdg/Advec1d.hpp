// 1-D linear advection RHS (CPU plumbing configuration).
// Same signature as the reference's blitzdg::advec1d::computeRHS
// (src/advec1d/main.cpp:126-188).
#pragma once
#include "Nodes1DProvisioner.hpp"
#include "Types.hpp"

namespace blitzdg {
namespace advec1d {
    void computeRHS(const real_matrix_type& u, real_type c, Nodes1DProvisioner& nodes1D, real_matrix_type& RHS);

    /// The reference driver loop (src/advec1d/main.cpp:35-122) with N, K and the
    /// domain as arguments: LSERK4 to t >= finalTime, returns max-norm error vs
    /// exp(-10 (x - c t)^2).
    real_type run(index_type N, index_type K, real_type xmin, real_type xmax, real_type c, real_type CFL,
                  real_type finalTime, index_type* numSteps = nullptr);
}
}
