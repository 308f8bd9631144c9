// C++ front end of the device sw2d path: the reference's computeRHS signature
// (src/sw2d-simple/main.cpp:181, decl SW2d.hpp:15) and an RAII solver, both thin
// callers of the C ABI. No numerical work happens here.
#include "blitzdg/SW2d.hpp"
#include "blitzdg/MeshManager.hpp"
#include "blitzdg_hip.h"
#include <map>
#include <mutex>
#include <string>

namespace blitzdg {
namespace sw2d {

namespace {

void check(int rc) {
    if (rc != BDG_OK) throw std::runtime_error(bdg_last_error());
}

bdg_sw2d* createFrom(const TriangleNodesProvisioner& p, real_type g, bool withFilter, int device, unsigned flags) {
    bdg_sw2d_desc d{};
    d.order = p.get_NOrder();
    d.num_elements = p.get_NumElements();
    d.Dr = p.get_Dr().data(); d.Ds = p.get_Ds().data(); d.Lift = p.get_Lift().data();
    d.Filter = withFilter ? p.get_Filter().data() : nullptr;
    d.rx = p.get_rx().data(); d.sx = p.get_sx().data(); d.ry = p.get_ry().data(); d.sy = p.get_sy().data();
    d.nx = p.get_nx().data(); d.ny = p.get_ny().data(); d.Fscale = p.get_Fscale().data();
    d.vmapM = p.get_vmapM().data(); d.vmapP = p.get_vmapP().data();
    const index_hashmap& bc = p.get_bcMap();
    const auto it = bc.find(BCTag::Wall);
    if (it != bc.end()) { d.mapW = it->second.data(); d.num_wall = static_cast<int>(it->second.size()); }
    d.g = g; d.device = device; d.flags = static_cast<int>(flags);
    bdg_sw2d* h = nullptr;
    check(bdg_sw2d_create(&d, &h));
    return h;
}

void requireShape(const real_matrix_type& m, index_type Np, index_type K, const char* what) {
    if (m.rows() != Np || m.cols() != K) throw std::runtime_error(std::string(what) + ": field must be (Np, K)");
}

struct CacheEntry { bdg_sw2d* solver; real_type g; };
std::map<const TriangleNodesProvisioner*, CacheEntry>& cache() {
    static std::map<const TriangleNodesProvisioner*, CacheEntry> c;
    return c;
}
std::mutex& cacheMutex() { static std::mutex m; return m; }

} // namespace

void computeRHS(real_matrix_type h, real_matrix_type hu, real_matrix_type hv, real_type g,
                TriangleNodesProvisioner& nodes, real_matrix_type& RHS1, real_matrix_type& RHS2,
                real_matrix_type& RHS3) {
    const index_type Np = nodes.get_NumLocalPoints(), K = nodes.get_NumElements();
    requireShape(h, Np, K, "computeRHS"); requireShape(hu, Np, K, "computeRHS"); requireShape(hv, Np, K, "computeRHS");
    bdg_sw2d* solver = nullptr;
    {
        std::lock_guard<std::mutex> lock(cacheMutex());
        auto it = cache().find(&nodes);
        if (it != cache().end() && it->second.g != g) {
            bdg_sw2d_destroy(it->second.solver);
            cache().erase(it);
            it = cache().end();
        }
        if (it == cache().end()) it = cache().emplace(&nodes, CacheEntry{createFrom(nodes, g, false, 0, 0), g}).first;
        solver = it->second.solver;
    }
    if (RHS1.rows() != Np || RHS1.cols() != K) RHS1.resize(Np, K);
    if (RHS2.rows() != Np || RHS2.cols() != K) RHS2.resize(Np, K);
    if (RHS3.rows() != Np || RHS3.cols() != K) RHS3.resize(Np, K);
    check(bdg_sw2d_rhs(solver, h.data(), hu.data(), hv.data(), RHS1.data(), RHS2.data(), RHS3.data(), 0));
}

void releaseDeviceImage(const TriangleNodesProvisioner& nodes) {
    std::lock_guard<std::mutex> lock(cacheMutex());
    const auto it = cache().find(&nodes);
    if (it == cache().end()) return;
    bdg_sw2d_destroy(it->second.solver);
    cache().erase(it);
}

DeviceSolver::DeviceSolver(const TriangleNodesProvisioner& nodes, real_type g, bool withFilter, int device,
                           unsigned flags)
    : h_{createFrom(nodes, g, withFilter, device, flags)}, Np_{nodes.get_NumLocalPoints()},
      K_{nodes.get_NumElements()} {}

DeviceSolver::~DeviceSolver() { bdg_sw2d_destroy(h_); }

void DeviceSolver::setState(const real_matrix_type& h, const real_matrix_type& hu, const real_matrix_type& hv) {
    requireShape(h, Np_, K_, "setState"); requireShape(hu, Np_, K_, "setState"); requireShape(hv, Np_, K_, "setState");
    check(bdg_sw2d_set_state(h_, h.data(), hu.data(), hv.data()));
}

void DeviceSolver::getState(real_matrix_type& h, real_matrix_type& hu, real_matrix_type& hv) {
    for (real_matrix_type* m : {&h, &hu, &hv})
        if (m->rows() != Np_ || m->cols() != K_) m->resize(Np_, K_);
    check(bdg_sw2d_get_state(h_, h.data(), hu.data(), hv.data()));
}

void DeviceSolver::setBathymetry(const real_matrix_type& H) {
    requireShape(H, Np_, K_, "setBathymetry");
    check(bdg_sw2d_set_bathymetry(h_, H.data()));
}

void DeviceSolver::computeRHS(const real_matrix_type& h, const real_matrix_type& hu, const real_matrix_type& hv,
                              real_matrix_type& R1, real_matrix_type& R2, real_matrix_type& R3, bool filter) {
    requireShape(h, Np_, K_, "computeRHS"); requireShape(hu, Np_, K_, "computeRHS"); requireShape(hv, Np_, K_, "computeRHS");
    for (real_matrix_type* m : {&R1, &R2, &R3})
        if (m->rows() != Np_ || m->cols() != K_) m->resize(Np_, K_);
    check(bdg_sw2d_rhs(h_, h.data(), hu.data(), hv.data(), R1.data(), R2.data(), R3.data(), filter ? 1 : 0));
}

void DeviceSolver::stepLSERK4(real_type dt, index_type n) { check(bdg_sw2d_step_lserk4(h_, dt, n)); }
void DeviceSolver::stepRK2(real_type dt, index_type n, bool filter) { check(bdg_sw2d_step_rk2(h_, dt, n, filter ? 1 : 0)); }

real_type DeviceSolver::computeTimeStep(real_type CFL, real_type* etaMax) {
    double dt = 0, em = 0;
    check(bdg_sw2d_compute_dt(h_, CFL, &dt, &em));
    if (etaMax) *etaMax = em;
    return dt;
}

} // namespace sw2d
} // namespace blitzdg
