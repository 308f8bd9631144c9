// C++ front end of the device sw2d path: the reference's computeRHS signature
// (src/sw2d-simple/main.cpp:181, decl SW2d.hpp:15) and an RAII solver, both thin
// callers of the C ABI. No numerical work happens here.
#include "blitzdg/SW2d.hpp"
#include "blitzdg/MeshManager.hpp"
#include "blitzdg_hip.h"
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

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

// ---------------------------------------------------------------- variant B (src/sw2d/main.cpp)

namespace {

struct VbEntry {
    bdg_sw2d* solver = nullptr;
    real_type g = 0, CD = 0, f = 0;
    std::vector<real_type> H, Hx, Hy; // what the device image was built from
};
std::map<const void*, VbEntry>& vbCache() {
    static std::map<const void*, VbEntry> c;
    return c;
}

bool same(const std::vector<real_type>& kept, const real_matrix_type& m) {
    const size_t n = static_cast<size_t>(m.rows()) * m.cols();
    return kept.size() == n && std::memcmp(kept.data(), m.data(), n * sizeof(real_type)) == 0;
}

const std::vector<index_type>& nodesOf(const index_hashmap& bc, index_type tag) {
    static const std::vector<index_type> none;
    const auto it = bc.find(tag);
    return it == bc.end() ? none : it->second;
}

bdg_sw2d* variantBSolver(const fields& fds, const physParams& phys, const DGContext2D& dg) {
    const index_type Np = dg.numLocalPoints(), K = dg.numElements();
    for (const real_matrix_type* m : {&fds.h, &fds.hu, &fds.hv, &fds.H, &fds.Hx, &fds.Hy}) requireShape(*m, Np, K, "computeRHS");
    VbEntry& e = vbCache()[dg.vmapP().data()];
    if (e.solver && (e.g != phys.g)) {
        bdg_sw2d_destroy(e.solver);
        e = VbEntry{};
    }
    if (!e.solver) {
        bdg_sw2d_desc d{};
        d.order = dg.order();
        d.num_elements = K;
        d.Dr = dg.Dr().data(); d.Ds = dg.Ds().data(); d.Lift = dg.lift().data();
        d.Filter = dg.filter().rows() == Np ? dg.filter().data() : nullptr;
        d.rx = dg.rx().data(); d.sx = dg.sx().data(); d.ry = dg.ry().data(); d.sy = dg.sy().data();
        d.nx = dg.nx().data(); d.ny = dg.ny().data(); d.Fscale = dg.fscale().data();
        d.vmapM = dg.vmapM().data(); d.vmapP = dg.vmapP().data();
        const std::vector<index_type>& mapW = nodesOf(dg.bcmap(), BCTag::Wall);
        d.mapW = mapW.data(); d.num_wall = static_cast<int>(mapW.size());
        d.g = phys.g;
        check(bdg_sw2d_create(&d, &e.solver));
        e.g = phys.g;
    }
    if (!same(e.H, fds.H) || !same(e.Hx, fds.Hx) || !same(e.Hy, fds.Hy) || e.CD != phys.CD || e.f != phys.f) {
        const std::vector<index_type>& mapO = nodesOf(dg.bcmap(), BCTag::Out);
        bdg_sw2d_vb_desc v{};
        v.H = fds.H.data(); v.Hx = fds.Hx.data(); v.Hy = fds.Hy.data();
        v.mapO = mapO.data(); v.num_out = static_cast<int>(mapO.size());
        v.drag = phys.CD; v.coriolis = phys.f;
        v.tide_amplitude = 3.0; v.tide_period = 3600 * 12.42; v.tide_ramp = 0.15 / 3600; // main.cpp:280-282,352
        check(bdg_sw2d_enable_variant_b(e.solver, &v));
        const size_t n = static_cast<size_t>(Np) * K;
        e.H.assign(fds.H.data(), fds.H.data() + n);
        e.Hx.assign(fds.Hx.data(), fds.Hx.data() + n);
        e.Hy.assign(fds.Hy.data(), fds.Hy.data() + n);
        e.CD = phys.CD; e.f = phys.f;
    }
    return e.solver;
}

} // namespace

void computeRHS(fields& fds, const numParams&, const physParams& phys, const DGContext2D& dg, real_type t) {
    std::lock_guard<std::mutex> lock(cacheMutex());
    bdg_sw2d* solver = variantBSolver(fds, phys, dg);
    const index_type Np = dg.numLocalPoints(), K = dg.numElements();
    for (real_matrix_type* m : {&fds.RHS1, &fds.RHS2, &fds.RHS3})
        if (m->rows() != Np || m->cols() != K) m->resize(Np, K);
    check(bdg_sw2d_set_time(solver, t));
    check(bdg_sw2d_rhs(solver, fds.h.data(), fds.hu.data(), fds.hv.data(), fds.RHS1.data(), fds.RHS2.data(),
                       fds.RHS3.data(), 0));
}

double computeTimeStep(fields& fds, const physParams& phys, const numParams& num, const DGContext2D& dg) {
    std::lock_guard<std::mutex> lock(cacheMutex());
    bdg_sw2d* solver = variantBSolver(fds, phys, dg);
    const index_type Np = dg.numLocalPoints(), K = dg.numElements();
    for (real_matrix_type* m : {&fds.u, &fds.v})
        if (m->rows() != Np || m->cols() != K) m->resize(Np, K);
    const size_t n = static_cast<size_t>(Np) * K;
    for (size_t i = 0; i < n; ++i) {                       // main.cpp:259-260 (side effect the caller may read)
        fds.u.data()[i] = fds.hu.data()[i] / fds.h.data()[i];
        fds.v.data()[i] = fds.hv.data()[i] / fds.h.data()[i];
    }
    check(bdg_sw2d_set_state(solver, fds.h.data(), fds.hu.data(), fds.hv.data()));
    // the reference does not look at eta here: report only dt (a NaN state still yields NaN)
    double dt = 0, em = 0;
    const int rc = bdg_sw2d_compute_dt(solver, num.CFL, &dt, &em);
    if (rc != BDG_OK && rc != BDG_ERR_UNSTABLE) check(rc);
    return dt;
}

void buildSpongeCoeff(const DGContext2D& dg, real_type spongeStrength, real_type radInfl, real_matrix_type& spongeCoeff) {
    const index_type Np = dg.numLocalPoints(), K = dg.numElements();
    if (spongeCoeff.rows() != Np || spongeCoeff.cols() != K) spongeCoeff.resize(Np, K);
    const std::vector<index_type>& mapO = nodesOf(dg.bcmap(), BCTag::Out);
    const real_matrix_type& x = dg.x(), &y = dg.y();
    const index_vector_type& vmapM = dg.vmapM();
    std::vector<real_type> xo(mapO.size()), yo(mapO.size());
    for (size_t i = 0; i < mapO.size(); ++i) {
        const index_type v = vmapM(mapO[i]);
        xo[i] = x(v % Np, v / Np);
        yo[i] = y(v % Np, v / Np);
    }
    for (index_type k = 0; k < K; ++k)
        for (index_type n = 0; n < Np; ++n) {
            real_type closest = 1.0e12;
            for (size_t i = 0; i < mapO.size(); ++i) {
                const real_type dist = std::hypot(x(n, k) - xo[i], y(n, k) - yo[i]);
                if (dist < radInfl && dist < closest) closest = dist;
            }
            if (closest < 1.0e12) spongeCoeff(n, k) = spongeStrength * (1.0 - closest / radInfl);
        }
}

void computeBedSlopes(const DGContext2D& dg, const real_matrix_type& H, real_matrix_type& Hx, real_matrix_type& Hy) {
    const index_type Np = dg.numLocalPoints(), K = dg.numElements();
    requireShape(H, Np, K, "computeBedSlopes");
    if (dg.filter().rows() != Np) throw std::runtime_error("computeBedSlopes: call buildFilter first");
    for (real_matrix_type* m : {&Hx, &Hy})
        if (m->rows() != Np || m->cols() != K) m->resize(Np, K);
    const real_matrix_type& Dr = dg.Dr(), &Ds = dg.Ds(), &F = dg.filter();
    std::vector<real_type> gx(Np), gy(Np);
    for (index_type k = 0; k < K; ++k) {
        for (index_type i = 0; i < Np; ++i) {
            real_type dr = 0, ds = 0;
            for (index_type m = 0; m < Np; ++m) { dr += Dr(i, m) * H(m, k); ds += Ds(i, m) * H(m, k); }
            gx[i] = dg.rx()(i, k) * dr + dg.sx()(i, k) * ds;
            gy[i] = dg.ry()(i, k) * dr + dg.sy()(i, k) * ds;
        }
        for (index_type i = 0; i < Np; ++i) {
            real_type ax = 0, ay = 0;
            for (index_type m = 0; m < Np; ++m) { ax += F(i, m) * gx[m]; ay += F(i, m) * gy[m]; }
            Hx(i, k) = ax;
            Hy(i, k) = ay;
        }
    }
}

void releaseDeviceImage(const DGContext2D& dg) {
    std::lock_guard<std::mutex> lock(cacheMutex());
    const auto it = vbCache().find(dg.vmapP().data());
    if (it == vbCache().end()) return;
    bdg_sw2d_destroy(it->second.solver);
    vbCache().erase(it);
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

void DeviceSolver::enableVariantB(const real_matrix_type& H, const real_matrix_type& Hx, const real_matrix_type& Hy,
                                  const std::vector<index_type>& mapO, real_type CD, real_type f,
                                  const real_matrix_type* sponge, real_type tideAmplitude, real_type tidePeriod,
                                  real_type tideRamp) {
    requireShape(H, Np_, K_, "enableVariantB"); requireShape(Hx, Np_, K_, "enableVariantB"); requireShape(Hy, Np_, K_, "enableVariantB");
    if (sponge) requireShape(*sponge, Np_, K_, "enableVariantB");
    bdg_sw2d_vb_desc v{};
    v.H = H.data(); v.Hx = Hx.data(); v.Hy = Hy.data();
    v.mapO = mapO.data(); v.num_out = static_cast<int>(mapO.size());
    v.drag = CD; v.coriolis = f;
    v.tide_amplitude = tideAmplitude; v.tide_period = tidePeriod; v.tide_ramp = tideRamp;
    v.sponge = sponge ? sponge->data() : nullptr;
    check(bdg_sw2d_enable_variant_b(h_, &v));
}

void DeviceSolver::stepSSPRK2(real_type dt, index_type n, bool filter, real_type spongeCoeff) {
    check(bdg_sw2d_step_ssprk2(h_, dt, n, filter ? 1 : 0, spongeCoeff));
}

void DeviceSolver::outputFields(real_matrix_type& eta, real_matrix_type& u, real_matrix_type& v,
                                const real_matrix_type* IM) {
    for (real_matrix_type* m : {&eta, &u, &v})
        if (m->rows() != Np_ || m->cols() != K_) m->resize(Np_, K_);
    if (IM && (IM->rows() != Np_ || IM->cols() != Np_)) throw std::runtime_error("outputFields: IM must be (Np, Np)");
    check(bdg_sw2d_output_fields(h_, IM ? IM->data() : nullptr, eta.data(), u.data(), v.data()));
}

void DeviceSolver::setTime(real_type t) { check(bdg_sw2d_set_time(h_, t)); }

real_type DeviceSolver::time() const {
    double t = 0;
    check(bdg_sw2d_get_time(h_, &t));
    return t;
}

real_type DeviceSolver::computeTimeStep(real_type CFL, real_type* etaMax) {
    double dt = 0, em = 0;
    check(bdg_sw2d_compute_dt(h_, CFL, &dt, &em));
    if (etaMax) *etaMax = em;
    return dt;
}

} // namespace sw2d
} // namespace blitzdg
