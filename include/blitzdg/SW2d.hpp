// Shallow-water (sw2d) right-hand side and device-resident stepping for C++ drivers.
//
// blitzdg::sw2d::computeRHS has the signature of the reference's
// src/sw2d-simple/SW2d.hpp:15 (fields by value, provisioner by reference, three
// caller-allocated (Np, K) outputs that are overwritten); it runs on the GPU
// through the C ABI (include/blitzdg_hip.h). DeviceSolver keeps the state in
// HBM for whole time loops (what the reference's main() does on the host,
// src/sw2d-simple/main.cpp:121-171).
#pragma once
#include "TriangleNodesProvisioner.hpp"
#include "Types.hpp"
#include <stdexcept>

struct bdg_sw2d; // C-ABI handle

namespace blitzdg {
namespace sw2d {

/// Drop-in for the reference's computeRHS. The device image of the provisioner's
/// tables is created on first use and cached per provisioner object; throws
/// std::runtime_error if no HIP device / kernel is available (there is no CPU path).
void computeRHS(real_matrix_type h, real_matrix_type hu, real_matrix_type hv, real_type g,
                TriangleNodesProvisioner& triangleNodesProvisioner, real_matrix_type& RHS1,
                real_matrix_type& RHS2, real_matrix_type& RHS3);

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
