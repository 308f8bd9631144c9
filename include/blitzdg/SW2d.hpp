// Shallow-water (sw2d) right-hand side and device-resident stepping for C++ drivers.
//
// blitzdg::sw2d::computeRHS has the signature of the reference's
// src/sw2d-simple/SW2d.hpp:15 (fields by value, provisioner by reference, three
// caller-allocated (Np, K) outputs that are overwritten); it runs on the GPU
// through the C ABI (include/blitzdg_hip.h). DeviceSolver keeps the state in
// HBM for whole time loops (what the reference's main() does on the host,
// src/sw2d-simple/main.cpp:121-171).
#pragma once
#include "DGContext2D.hpp"
#include "TriangleNodesProvisioner.hpp"
#include "Types.hpp"
#include <stdexcept>
#include <string>

struct bdg_sw2d; // C-ABI handle

namespace blitzdg {
namespace sw2d {

/// Drop-in for the reference's computeRHS. The device image of the provisioner's
/// tables is created on first use and cached per provisioner object; throws
/// std::runtime_error if no HIP device / kernel is available (there is no CPU path).
void computeRHS(real_matrix_type h, real_matrix_type hu, real_matrix_type hv, real_type g,
                TriangleNodesProvisioner& triangleNodesProvisioner, real_matrix_type& RHS1,
                real_matrix_type& RHS2, real_matrix_type& RHS3);

// ---- the C++ "sw2d" driver's vocabulary (reference src/sw2d/SW2d.hpp:15-55, src/sw2d/main.cpp)
struct physParams {
    const real_type g = 9.81;
    real_type CD;
    real_type f;
    real_type initTime;
    real_type finalTime;
};

struct numParams {
    index_type N;  // order of polynomials
    real_type CFL;
    index_type outputInterval;
    real_type filterPercent = 0.95;
    index_type filterOrder = 4;
};

struct fields {
    real_matrix_type h, hu, hv, H, Hx, Hy;
    real_matrix_type eta, u, v;
    real_matrix_type RHS1, RHS2, RHS3;
    real_matrix_type resRK1, resRK2, resRK3;
};

/// Variant-B right-hand side at time t (reference src/sw2d/main.cpp:279-484): reads fds.h/hu/hv and
/// fds.H/Hx/Hy, overwrites fds.RHS1..3. Open-boundary nodes are dg.bcmap()[BCTag::Out], walls
/// dg.bcmap()[BCTag::Wall]. The reference takes `fields` by value because blitz arrays share storage
/// on copy; real_matrix_type is a value type, so the struct is taken by reference here (the call
/// `sw2d::computeRHS(fields_n, n, p, dg, t)` is unchanged). Runs on the GPU; the device image is
/// cached per DGContext2D table set and re-uploaded when H, Hx or Hy change.
void computeRHS(fields& fds, const numParams& num, const physParams& phys, const DGContext2D& dg, real_type t);
/// dt of the variant-B driver (main.cpp:253-277); also refreshes fds.u, fds.v as the reference does.
double computeTimeStep(fields& fds, const physParams& phys, const numParams& num, const DGContext2D& dg);
/// Sponge coefficient around the open boundary (main.cpp:516-556).
void buildSpongeCoeff(const DGContext2D& dg, real_type spongeStrength, real_type radInfl, real_matrix_type& spongeCoeff);
/// Bed slopes Hx, Hy = Filter (grad H) as the driver builds them (main.cpp:128-133).
void computeBedSlopes(const DGContext2D& dg, const real_matrix_type& H, real_matrix_type& Hx, real_matrix_type& Hy);
/// Releases the cached device image of a context's tables.
void releaseDeviceImage(const DGContext2D& dg);

/// Releases the cached device image of a provisioner (call before destroying it).
void releaseDeviceImage(const TriangleNodesProvisioner& triangleNodesProvisioner);

/// RAII wrapper of the device-resident solver.
class DeviceSolver {
public:
    DeviceSolver(const TriangleNodesProvisioner& nodes, real_type g, bool withFilter, int device = 0,
                 unsigned flags = 0);
    ~DeviceSolver();
    DeviceSolver(const DeviceSolver&) = delete;
    DeviceSolver& operator=(const DeviceSolver&) = delete;

    void setState(const real_matrix_type& h, const real_matrix_type& hu, const real_matrix_type& hv);
    void getState(real_matrix_type& h, real_matrix_type& hu, real_matrix_type& hv);
    void setBathymetry(const real_matrix_type& H);
    void computeRHS(const real_matrix_type& h, const real_matrix_type& hu, const real_matrix_type& hv,
                    real_matrix_type& RHS1, real_matrix_type& RHS2, real_matrix_type& RHS3, bool filter = false);
    void stepLSERK4(real_type dt, index_type numSteps = 1);
    void stepRK2(real_type dt, index_type numSteps = 1, bool filter = true);
    /// Switches the solver to the variant-B physics (see computeRHS(fields&, ...) above). mapO: open-
    /// boundary face nodes; sponge: optional (Np, K) coefficient field used by stepSSPRK2.
    void enableVariantB(const real_matrix_type& H, const real_matrix_type& Hx, const real_matrix_type& Hy,
                        const std::vector<index_type>& mapO, real_type CD, real_type f,
                        const real_matrix_type* sponge = nullptr, real_type tideAmplitude = 3.0,
                        real_type tidePeriod = 3600 * 12.42, real_type tideRamp = 0.15 / 3600);
    /// Heun steps of the variant-B driver loop (main.cpp:211-236); advances time() by dt per step.
    void stepSSPRK2(real_type dt, index_type numSteps = 1, bool filter = false, real_type spongeCoeff = 0.0);
    /// Output step: eta = h - H (h without bathymetry), u = hu/h, v = hv/h of the resident state; with
    /// IM (TriangleNodesProvisioner::splitOperators) interpolated on the device to the equispaced
    /// lattice the *.vtu writer cuts into triangles.
    void outputFields(real_matrix_type& eta, real_matrix_type& u, real_matrix_type& v,
                      const real_matrix_type* IM = nullptr);
    void setTime(real_type t);
    real_type time() const;
    /// dt = CFL / ((N+1)^2 * 0.5 * max|Fscale|*(|u| + sqrt(g h))); throws
    /// std::runtime_error("A numerical instability has occurred!") on NaN / |eta| > 1e8.
    real_type computeTimeStep(real_type CFL, real_type* etaMax = nullptr);
    bdg_sw2d* handle() { return h_; }

private:
    bdg_sw2d* h_ = nullptr;
    index_type Np_ = 0, K_ = 0;
};

} // namespace sw2d
} // namespace blitzdg
