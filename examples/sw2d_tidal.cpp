// sw2d (tidal, variable depth) on the MI355X path: the reference driver src/sw2d/main.cpp:45-245
// written against this repo's headers. Same structure: filter, depth field clamped from below
// (readDepthData, :486-514 -- here a synthetic bed, the reference's input/H0_try2.oct is not shipped),
// bed slopes through the filter (:128-133), open-boundary faces re-tagged from a vertex set and
// buildBCHash called again (:161-176), sponge layer (:178-181), SSP-RK2 loop with adaptive dt (:199-240).
//   ./bin/sw2d [mesh.msh|box:NXxNY] [order] [maxSteps] [resident|dropin]
// "dropin" runs the reference's loop statements on the host with sw2d::computeRHS(fields, ...) called
// twice per step (host arrays in and out); "resident" keeps the state in HBM (DeviceSolver).
#include "blitzdg/BlitzHelpers.hpp"
#include "blitzdg/MeshManager.hpp"
#include "blitzdg/SW2d.hpp"
#include "blitzdg/TriangleNodesProvisioner.hpp"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <string>
#include <unordered_set>

int main(int argc, char** argv) {
    using namespace blitzdg;
    const std::string meshArg = argc > 1 ? argv[1] : "tests/golden/coarse_box.msh";
    const std::string mode = argc > 4 ? argv[4] : "resident";
    sw2d::physParams p;
    p.CD = 2.5e-3;
    p.f = 1.0070e-4;
    p.initTime = 0.0;
    p.finalTime = 24.0 * 3600.0;
    sw2d::numParams n;
    n.N = argc > 2 ? std::atoi(argv[2]) : 1;
    n.CFL = 0.25;
    n.outputInterval = 20;
    n.filterPercent = 0.90;
    n.filterOrder = 4;
    const index_type maxSteps = argc > 3 ? std::atoi(argv[3]) : 100;
    try {
        MeshManager meshManager;
        if (meshArg.rfind("box:", 0) == 0) {
            const auto x = meshArg.find('x', 4);
            meshManager.buildBoxMesh(std::atoi(meshArg.substr(4, x - 4).c_str()), std::atoi(meshArg.substr(x + 1).c_str()),
                                     -1, 1, -1, 1);
        } else {
            meshManager.readMesh(meshArg);
        }
        TriangleNodesProvisioner triangleNodesProvisioner(n.N, meshManager);
        triangleNodesProvisioner.buildFilter(n.filterPercent * static_cast<real_type>(n.N), n.filterOrder);
        DGContext2D dg = triangleNodesProvisioner.get_DGContext();
        const index_type Np = dg.numLocalPoints(), K = dg.numElements();

        sw2d::fields fields_n;
        for (real_matrix_type* m : {&fields_n.h, &fields_n.hu, &fields_n.hv, &fields_n.H, &fields_n.Hx, &fields_n.Hy,
                                    &fields_n.eta, &fields_n.u, &fields_n.v, &fields_n.RHS1, &fields_n.RHS2,
                                    &fields_n.RHS3})
            m->resize(Np, K);
        const real_matrix_type& x = dg.x(), &y = dg.y();
        for (index_type i = 0; i < Np; ++i)
            for (index_type k = 0; k < K; ++k)
                fields_n.H(i, k) = std::max(150.0, 200.0 + 40.0 * x(i, k) - 25.0 * y(i, k) * y(i, k));
        sw2d::computeBedSlopes(dg, fields_n.H, fields_n.Hx, fields_n.Hy);
        fields_n.h = fields_n.H; // eta = 0, u = v = 0

        // open boundary: faces whose two vertices are both in the set (here: the left edge)
        const real_vector_type& verts = meshManager.get_Vertices();
        const index_vector_type& EToV = meshManager.get_Elements();
        real_type xmin = 1e300;
        for (index_type v = 0; v < meshManager.get_NumVerts(); ++v) xmin = std::min(xmin, verts(3 * v));
        std::unordered_set<index_type> obcNodes;
        for (index_type v = 0; v < meshManager.get_NumVerts(); ++v)
            if (std::abs(verts(3 * v) - xmin) < 1e-12) obcNodes.insert(v);
        index_vector_type bcType = meshManager.get_BCType();
        for (index_type k = 0; k < K; ++k) {
            const index_type v1 = EToV(3 * k), v2 = EToV(3 * k + 1), v3 = EToV(3 * k + 2);
            if (obcNodes.count(v1) > 0 && obcNodes.count(v2) > 0) bcType(3 * k) = BCTag::Out;
            if (obcNodes.count(v2) > 0 && obcNodes.count(v3) > 0) bcType(3 * k + 1) = BCTag::Out;
            if (obcNodes.count(v3) > 0 && obcNodes.count(v1) > 0) bcType(3 * k + 2) = BCTag::Out;
        }
        triangleNodesProvisioner.buildBCHash(bcType);

        real_matrix_type spongeCoeff(Np, K);
        sw2d::buildSpongeCoeff(dg, 1.0e-3, 0.5, spongeCoeff);

        real_type t = p.initTime, dt = 0;
        index_type count = 0;
        if (mode == "dropin") {
            sw2d::fields fields_np1 = fields_n;
            while (t < p.finalTime && count < maxSteps) {
                dt = sw2d::computeTimeStep(fields_n, p, n, dg);
                sw2d::computeRHS(fields_n, n, p, dg, t);
                for (index_type i = 0; i < Np; ++i)
                    for (index_type k = 0; k < K; ++k) {
                        fields_np1.h(i, k) = fields_n.h(i, k) + dt * fields_n.RHS1(i, k);
                        real_type a = fields_n.hu(i, k) + dt * fields_n.RHS2(i, k);
                        real_type b = fields_n.hv(i, k) + dt * fields_n.RHS3(i, k);
                        fields_np1.hu(i, k) = a / (1.0 + spongeCoeff(i, k) * a * a);
                        fields_np1.hv(i, k) = b / (1.0 + spongeCoeff(i, k) * b * b);
                    }
                sw2d::computeRHS(fields_np1, n, p, dg, t);
                for (index_type i = 0; i < Np; ++i)
                    for (index_type k = 0; k < K; ++k) {
                        fields_n.h(i, k) = 0.5 * (fields_n.h(i, k) + fields_np1.h(i, k) + dt * fields_np1.RHS1(i, k));
                        real_type a = 0.5 * (fields_n.hu(i, k) + fields_np1.hu(i, k) + dt * fields_np1.RHS2(i, k));
                        real_type b = 0.5 * (fields_n.hv(i, k) + fields_np1.hv(i, k) + dt * fields_np1.RHS3(i, k));
                        fields_n.hu(i, k) = a / (1.0 + spongeCoeff(i, k) * a * a);
                        fields_n.hv(i, k) = b / (1.0 + spongeCoeff(i, k) * b * b);
                    }
                t += dt;
                ++count;
            }
            sw2d::releaseDeviceImage(dg);
        } else {
            const index_hashmap& bc = dg.bcmap();
            const auto it = bc.find(BCTag::Out);
            const std::vector<index_type> mapO = it == bc.end() ? std::vector<index_type>{} : it->second;
            sw2d::DeviceSolver solver(triangleNodesProvisioner, p.g, /*withFilter=*/true);
            solver.enableVariantB(fields_n.H, fields_n.Hx, fields_n.Hy, mapO, p.CD, p.f, &spongeCoeff);
            solver.setState(fields_n.h, fields_n.hu, fields_n.hv);
            solver.setTime(t);
            while (t < p.finalTime && count < maxSteps) {
                real_type etaMax = 0;
                dt = solver.computeTimeStep(n.CFL, &etaMax); // throws "A numerical instability has occurred!"
                if ((count % n.outputInterval) == 0) std::cout << "t=" << t << ", dt=" << dt << ", eta_max=" << etaMax << "\n";
                solver.stepSSPRK2(dt, 1);
                t += dt;
                ++count;
            }
            solver.getState(fields_n.h, fields_n.hu, fields_n.hv);
        }
        for (index_type i = 0; i < Np; ++i)
            for (index_type k = 0; k < K; ++k) fields_n.eta(i, k) = fields_n.h(i, k) - fields_n.H(i, k);
        std::cout.precision(12);
        std::cout << "done: mode=" << mode << ", steps=" << count << ", t=" << t << ", eta_max=" << normMax(fields_n.eta)
                  << ", |hu|max=" << normMax(fields_n.hu) << ", |hv|max=" << normMax(fields_n.hv) << "\n";
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
