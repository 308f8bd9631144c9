// burgers1d: 1-D viscous Burgers equation with the LSERK4 loop, on the host (CPU plumbing beside advec1d; no GPU involved).
// Algorithm: reference src/burgers1d/main.cpp:28-115 (driver), :119-226 (exact solution, RHS).
#include "blitzdg/Burgers1d.hpp"
#include "blitzdg/BlitzHelpers.hpp"
#include "blitzdg/LSERK4.hpp"
#include <algorithm>
#include <cmath>
#include <stdexcept>

namespace blitzdg {
namespace burgers1d {

real_type Burgers2(const real_type x, const real_type t, const real_type alpha, const real_type nu, const real_type c) {
    return (c / alpha) - (c / alpha) * std::tanh(0.5 * (c / nu) * (x - c * t));
}

void Burgers2(real_matrix_type& u, const real_matrix_type& x, const real_type t, const real_type alpha, const real_type nu,
              const real_type c) {
    if (u.rows() != x.rows() || u.cols() != x.cols()) u.resize(x.rows(), x.cols());
    for (index_type i = 0; i < x.rows(); ++i)
        for (index_type k = 0; k < x.cols(); ++k) u(i, k) = Burgers2(x(i, k), t, alpha, nu, c);
}

void computeRHS(const real_matrix_type& u, const real_matrix_type& x, real_type t, real_type c, real_type alpha, real_type nu,
                Nodes1DProvisioner& nodes1D, real_matrix_type& RHS) {
    const real_matrix_type& Dr = nodes1D.get_Dr();
    const real_matrix_type& rx = nodes1D.get_rx();
    const real_matrix_type& Lift = nodes1D.get_Lift();
    const real_matrix_type& Fscale = nodes1D.get_Fscale();
    const real_matrix_type& nx = nodes1D.get_nx();
    const index_vector_type& vmapM = nodes1D.get_vmapM();
    const index_vector_type& vmapP = nodes1D.get_vmapP();
    const index_type mapO = nodes1D.get_mapO(), mapI = nodes1D.get_mapI();
    const index_type vmapO = nodes1D.get_vmapO(), vmapI = nodes1D.get_vmapI();
    const index_type nFace = Nodes1DProvisioner::NumFaces * Nodes1DProvisioner::NumFacePoints;
    const index_type Np = nodes1D.get_NumLocalPoints(), K = nodes1D.get_NumElements();
    const real_type snu = std::sqrt(nu);

    real_vector_type uVec(Np * K), xVec(Np * K), nxVec(nFace * K), uM(nFace * K), uP(nFace * K);
    fullToVector(nx, nxVec, false);
    fullToVector(u, uVec, false);
    fullToVector(x, xVec, false);
    applyIndexMap(uVec, vmapM, uM);
    applyIndexMap(uVec, vmapP, uP);
    real_type maxvel = 0;                                             // main.cpp:172
    for (index_type i = 0; i < Np * K; ++i) maxvel = std::max(maxvel, std::fabs(uVec(i)));

    // boundary values from the travelling wave (:189-190), jumps with the doubled boundary form (:193-195)
    const real_type uL = Burgers2(xVec(vmapI), t, alpha, nu, c), uR = Burgers2(xVec(vmapO), t, alpha, nu, c);
    real_vector_type du(nFace * K), du2(nFace * K);
    for (index_type i = 0; i < nFace * K; ++i) {
        du(i) = uM(i) - uP(i);
        du2(i) = 0.5 * (uM(i) * uM(i) - uP(i) * uP(i));               // :211
    }
    du(mapI) = 2 * (uVec(vmapI) - uL);
    du(mapO) = 2 * (uVec(vmapO) - uR);
    du2(mapI) = uVec(vmapI) * uVec(vmapI) - uL * uL;
    du2(mapO) = uVec(vmapO) * uVec(vmapO) - uR * uR;
    real_matrix_type duMat(nFace, K);
    vectorToFull(du, duMat, false);

    // q = sqrt(nu) (rx Dr u - Lift (Fscale nx du / 2))     (:199-200)
    real_matrix_type q(Np, K);
    for (index_type i = 0; i < Np; ++i)
        for (index_type k = 0; k < K; ++k) {
            real_type dudr = 0, surf = 0;
            for (index_type m = 0; m < Np; ++m) dudr += Dr(i, m) * u(m, k);
            for (index_type m = 0; m < nFace; ++m) surf += Lift(i, m) * (0.5 * Fscale(m, k) * nx(m, k) * duMat(m, k));
            q(i, k) = snu * (rx(i, k) * dudr - surf);
        }
    real_vector_type qVec(Np * K), qM(nFace * K), qP(nFace * K), dq(nFace * K), flux(nFace * K);
    fullToVector(q, qVec, false);
    applyIndexMap(qVec, vmapM, qM);
    applyIndexMap(qVec, vmapP, qP);
    for (index_type i = 0; i < nFace * K; ++i) dq(i) = 0.5 * (qM(i) - qP(i));   // :206
    dq(mapI) = 0.0;
    dq(mapO) = 0.0;
    for (index_type i = 0; i < nFace * K; ++i)                                   // :216
        flux(i) = nxVec(i) * (0.5 * du2(i) - snu * dq(i)) - (0.5 * maxvel) * du(i);
    real_matrix_type fluxMat(nFace, K);
    vectorToFull(flux, fluxMat, false);

    // RHS = -rx Dr (u^2/2 - sqrt(nu) q) + Lift (Fscale flux)     (:220-224)
    if (RHS.rows() != Np || RHS.cols() != K) RHS.resize(Np, K);
    for (index_type i = 0; i < Np; ++i)
        for (index_type k = 0; k < K; ++k) {
            real_type vol = 0, surf = 0;
            for (index_type m = 0; m < Np; ++m) vol += Dr(i, m) * (0.5 * u(m, k) * u(m, k) - snu * q(m, k));
            for (index_type m = 0; m < nFace; ++m) surf += Lift(i, m) * (Fscale(m, k) * fluxMat(m, k));
            RHS(i, k) = -rx(i, k) * vol + surf;
        }
}

real_type run(index_type N, index_type K, real_type xmin, real_type xmax, real_type alpha, real_type nu, real_type c, real_type CFL,
              real_type finalTime, index_type* numSteps) {
    Nodes1DProvisioner nodes(N, K, xmin, xmax);
    nodes.buildNodes();
    nodes.computeJacobian();
    const index_type Np = nodes.get_NumLocalPoints();
    const real_matrix_type& x = nodes.get_xGrid();
    const real_type dx = x(1, 0) - x(0, 0);
    const real_type dt = CFL * std::min(dx / std::fabs(c), dx * dx / std::sqrt(nu));   // main.cpp:58-60

    real_matrix_type u(Np, K), RHS(Np, K), resRK(Np, K);
    real_type t = 0.0;
    Burgers2(u, x, t, alpha, nu, c);
    index_type count = 0;
    while (t < finalTime) {
        for (index_type s = 0; s < LSERK4::numStages; ++s) {
            computeRHS(u, x, t, c, alpha, nu, nodes, RHS);     // (t, not the stage time: as the reference)
            for (index_type i = 0; i < Np; ++i)
                for (index_type k = 0; k < K; ++k) {
                    resRK(i, k) = LSERK4::rk4a[s] * resRK(i, k) + dt * RHS(i, k);
                    u(i, k) += LSERK4::rk4b[s] * resRK(i, k);
                }
        }
        const real_type umax = normMax(u);
        if (umax > 1e8 || std::isnan(umax)) throw std::runtime_error("A numerical instability has occurred!");
        t += dt;
        ++count;
    }
    real_type err = 0;
    for (index_type i = 0; i < Np; ++i)
        for (index_type k = 0; k < K; ++k) err = std::max(err, std::fabs(u(i, k) - Burgers2(x(i, k), t, alpha, nu, c)));
    if (numSteps) *numSteps = count;
    return err;
}

} // namespace burgers1d
} // namespace blitzdg
