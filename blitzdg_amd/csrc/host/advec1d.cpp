// advec1d: 1-D linear advection with upwind flux and the LSERK4 loop, on the
// host (the reference's CPU-runnable configuration; no GPU involved).
// Algorithm: reference src/advec1d/main.cpp:35-122 (driver), :126-188 (RHS).
#include "blitzdg/Advec1d.hpp"
#include "blitzdg/BlitzHelpers.hpp"
#include "blitzdg/LSERK4.hpp"
#include <cmath>
#include <stdexcept>

namespace blitzdg {
namespace advec1d {

void computeRHS(const real_matrix_type& u, real_type c, Nodes1DProvisioner& nodes1D, real_matrix_type& RHS) {
    const real_matrix_type& Dr = nodes1D.get_Dr();
    const real_matrix_type& rx = nodes1D.get_rx();
    const real_matrix_type& Lift = nodes1D.get_Lift();
    const real_matrix_type& Fscale = nodes1D.get_Fscale();
    const real_matrix_type& nx = nodes1D.get_nx();
    const index_vector_type& vmapM = nodes1D.get_vmapM();
    const index_vector_type& vmapP = nodes1D.get_vmapP();
    const index_type mapO = nodes1D.get_mapO(), mapI = nodes1D.get_mapI();
    const index_type nFace = Nodes1DProvisioner::NumFaces * Nodes1DProvisioner::NumFacePoints;
    const index_type Np = nodes1D.get_NumLocalPoints(), K = nodes1D.get_NumElements();
    const real_type alpha = 0; // 1 = central flux, 0 = upwind

    real_vector_type uVec(Np * K), nxVec(nFace * K), uM(nFace * K), uP(nFace * K), du(nFace * K);
    fullToVector(nx, nxVec, false);
    fullToVector(u, uVec, false);
    applyIndexMap(uVec, vmapM, uM);
    applyIndexMap(uVec, vmapP, uP);
    uP(mapO) = uM(mapO); // outflow
    uP(mapI) = 0;        // inflow
    for (index_type i = 0; i < nFace * K; ++i)
        du(i) = (uM(i) - uP(i)) * 0.5 * (c * nxVec(i) - (1 - alpha) * std::fabs(c * nxVec(i)));
    real_matrix_type duMat(nFace, K);
    vectorToFull(du, duMat, false);

    if (RHS.rows() != Np || RHS.cols() != K) RHS.resize(Np, K);
    for (index_type i = 0; i < Np; ++i)
        for (index_type k = 0; k < K; ++k) {
            real_type dudr = 0;
            for (index_type m = 0; m < Np; ++m) dudr += Dr(i, m) * u(m, k);
            real_type surf = 0;
            for (index_type m = 0; m < nFace; ++m) surf += Lift(i, m) * (Fscale(m, k) * duMat(m, k));
            RHS(i, k) = -c * rx(i, k) * dudr + surf;
        }
}

real_type run(index_type N, index_type K, real_type xmin, real_type xmax, real_type c, real_type CFL,
              real_type finalTime, index_type* numSteps) {
    Nodes1DProvisioner nodes(N, K, xmin, xmax);
    nodes.buildNodes();
    nodes.computeJacobian();
    const index_type Np = nodes.get_NumLocalPoints();
    const real_matrix_type& x = nodes.get_xGrid();
    const real_type dt = CFL * (x(1, 0) - x(0, 0)) / std::fabs(c);

    real_matrix_type u(Np, K), RHS(Np, K), resRK(Np, K);
    for (index_type i = 0; i < Np; ++i)
        for (index_type k = 0; k < K; ++k) u(i, k) = std::exp(-10 * (x(i, k) * x(i, k)));

    real_type t = 0.0;
    index_type count = 0;
    while (t < finalTime) {
        for (index_type s = 0; s < LSERK4::numStages; ++s) {
            computeRHS(u, c, nodes, RHS);
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
    const real_type shift = c * t;
    real_type err = 0;
    for (index_type i = 0; i < Np; ++i)
        for (index_type k = 0; k < K; ++k) {
            const real_type d = x(i, k) - shift;
            err = std::max(err, std::fabs(u(i, k) - std::exp(-10 * (d * d))));
        }
    if (numSteps) *numSteps = count;
    return err;
}

} // namespace advec1d
} // namespace blitzdg
