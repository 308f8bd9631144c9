// 1-D viscous Burgers solver (CPU plumbing beside advec1d; SURVEY 8f.4's tail): u_t + (u^2/2)_x = nu u_xx as the first-order
// system q = sqrt(nu) u_x, u_t = -(u^2/2 - sqrt(nu) q)_x, local Lax-Friedrichs flux, LSERK4.
// Same signatures as the reference's blitzdg::burgers1d (src/burgers1d/Burgers1d.hpp:13-19; RHS src/burgers1d/main.cpp:129-226).
#pragma once
#include "Nodes1DProvisioner.hpp"
#include "Types.hpp"

namespace blitzdg {
namespace burgers1d {
    /// Travelling-wave solution u = c/alpha - (c/alpha) tanh(c/(2 nu) (x - c t))   (main.cpp:119-126).
    real_type Burgers2(const real_type x, const real_type t, const real_type alpha, const real_type nu, const real_type c);
    void Burgers2(real_matrix_type& u, const real_matrix_type& x, const real_type t, const real_type alpha, const real_type nu,
                  const real_type c);
    void computeRHS(const real_matrix_type& u, const real_matrix_type& x, real_type t, real_type c, real_type alpha, real_type nu,
                    Nodes1DProvisioner& nodes1D, real_matrix_type& RHS);

    /// The reference driver loop (src/burgers1d/main.cpp:28-115) with its constants as arguments: LSERK4 to t >= finalTime with
    /// dt = CFL min(dx / |c|, dx^2 / sqrt(nu)), dx the first node spacing; returns the max-norm error against Burgers2.
    real_type run(index_type N, index_type K, real_type xmin, real_type xmax, real_type alpha, real_type nu, real_type c,
                  real_type CFL, real_type finalTime, index_type* numSteps = nullptr);
}
}
