// sw2d-simple on the MI355X path: the reference driver src/sw2d-simple/main.cpp:33-177
// written against this repo's headers. Same set-up (mesh, order, filter, Gaussian hump over
// H = 10, CFL 0.65, midpoint RK2 + filter, adaptive dt, blow-up check, progress line every
// 10 steps); the time loop runs on the device and the host only reads back for output.
//   ./bin/sw2d-simple [mesh.msh|box:NXxNY] [order] [finalTime] [maxSteps] [outputDir]
// With outputDir, eta/u/v are written as *.vtu every 10 steps like the reference (:123-131), from
// fields computed on the device.
#include "blitzdg/BlitzHelpers.hpp"
#include "blitzdg/MeshManager.hpp"
#include "blitzdg/SW2d.hpp"
#include "blitzdg/TriangleNodesProvisioner.hpp"
#include "blitzdg/VtkOutputter.hpp"
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <string>

int main(int argc, char** argv) {
    using namespace blitzdg;
    const std::string meshArg = argc > 1 ? argv[1] : "tests/golden/coarse_box.msh";
    const index_type N = argc > 2 ? std::atoi(argv[2]) : 3;
    const real_type finalTime = argc > 3 ? std::atof(argv[3]) : 0.05;
    const index_type maxSteps = argc > 4 ? std::atoi(argv[4]) : 1000000;
    const std::string outDir = argc > 5 ? argv[5] : "";
    const real_type g = 9.81, CFL = 0.65;
    try {
        MeshManager meshManager;
        if (meshArg.rfind("box:", 0) == 0) {
            const auto x = meshArg.find('x', 4);
            meshManager.buildBoxMesh(std::atoi(meshArg.substr(4, x - 4).c_str()), std::atoi(meshArg.substr(x + 1).c_str()),
                                     -1, 1, -1, 1);
        } else {
            meshManager.readMesh(meshArg);
        }
        const index_type K = meshManager.get_NumElements();
        TriangleNodesProvisioner nodes(N, meshManager);
        nodes.buildFilter(0.9 * N, N);
        const real_matrix_type& x = nodes.get_xGrid();
        const real_matrix_type& y = nodes.get_yGrid();
        const index_type Np = nodes.get_NumLocalPoints();

        real_matrix_type H(Np, K), h(Np, K), hu(Np, K), hv(Np, K), eta(Np, K);
        for (index_type i = 0; i < Np; ++i)
            for (index_type k = 0; k < K; ++k) {
                H(i, k) = 10.0;
                h(i, k) = H(i, k) + std::exp(-10 * (x(i, k) * x(i, k)) - 10 * (y(i, k) * y(i, k)));
            }
        sw2d::DeviceSolver solver(nodes, g, /*withFilter=*/true);
        solver.setState(h, hu, hv);
        solver.setBathymetry(H);

        real_type etaMax = 0, t = 0.0, dt = solver.computeTimeStep(CFL, &etaMax);
        index_type count = 0;
        VtkOutputter outputter(nodes);
        real_matrix_type u(Np, K), v(Np, K);
        while (t < finalTime && count < maxSteps) {
            if ((count % 10) == 0 && !outDir.empty()) {
                solver.outputFields(eta, u, v);
                outputter.writeFieldToFile(outDir + "/" + outputter.generateFileName("eta", count), eta, "eta");
                outputter.writeFieldToFile(outDir + "/" + outputter.generateFileName("u", count), u, "u");
                outputter.writeFieldToFile(outDir + "/" + outputter.generateFileName("v", count), v, "v");
            }
            if ((count % 10) == 0) std::cout << "t=" << t << ", eta_max=" << etaMax << ", dt=" << dt << "\n";
            solver.stepRK2(dt, 1, /*filter=*/true);
            dt = solver.computeTimeStep(CFL, &etaMax); // throws on NaN / |eta| > 1e8
            t += dt;
            ++count;
        }
        solver.getState(h, hu, hv);
        std::cout << "done: steps=" << count << ", t=" << t << ", eta_max=" << etaMax << ", |hu|max=" << normMax(hu) << "\n";
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
