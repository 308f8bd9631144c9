// Sanitizer run of the host-side setup path (no GPU): mesh reader/writer, connectivity, partition,
// provisioners, filter, bed slopes, split elements, vtu writer, advec1d.
#include "blitzdg/Advec1d.hpp"
#include "blitzdg/MeshManager.hpp"
#include "blitzdg/Nodes1DProvisioner.hpp"
#include "blitzdg/TriangleNodesProvisioner.hpp"
#include "blitzdg/VtkOutputter.hpp"
#include <cstdio>
#include <iostream>
using namespace blitzdg;
int main(int argc, char** argv) {
    const std::string mshPath = argc > 1 ? argv[1] : "tests/golden/coarse_box.msh";
    MeshManager m;
    m.readMesh(mshPath);
    m.partitionMesh(3);
    m.writeMesh("/tmp/asan_box.msh");
    MeshManager m2;
    m2.readMesh("/tmp/asan_box.msh");
    for (int order : {1, 2, 3, 5, 8}) {
        TriangleNodesProvisioner nodes(order, m2);
        nodes.buildFilter(0.9 * order, order < 2 ? 2 : order);
        DGContext2D ctx = nodes.get_DGContext();
        (void)nodes.get_gather();
        real_matrix_type f(ctx.numLocalPoints(), ctx.numElements()), xn, yn, fn;
        for (index_type i = 0; i < f.rows(); ++i)
            for (index_type k = 0; k < f.cols(); ++k) f(i, k) = ctx.x()(i, k) + 2 * ctx.y()(i, k);
        nodes.splitElements(ctx.x(), ctx.y(), f, xn, yn, fn);
        VtkOutputter out(nodes);
        out.writeFieldToFile("/tmp/asan_f.vtu", f, "f");
        index_vector_type bc = m2.get_BCType();
        nodes.buildBCHash(bc);
    }
    MeshManager box;
    box.buildBoxMesh(37, 23, -1, 1, -1, 1, 12345);
    box.partitionMesh(8);
    TriangleNodesProvisioner bn(4, box);
    Nodes1DProvisioner n1(4, 100, -1.0, 4.0);
    n1.buildNodes();
    n1.computeJacobian();
    index_type steps = 0;
    const real_type err = advec1d::run(4, 30, -1.0, 4.0, 0.1, 0.8, 2.0, &steps);
    std::printf("host check ok: K=%d, advec1d err %.3e in %d steps\n", box.get_NumElements(), err, steps);
    std::remove("/tmp/asan_box.msh");
    std::remove("/tmp/asan_f.vtu");
    return 0;
}
