// Sanitizer run of the host-side setup path (no GPU): mesh reader/writer, connectivity, partition,
// provisioners, filter, bed slopes, split elements, vtu writer, advec1d, Gauss-face / cubature contexts, and the
// reshape helpers on the cases of the reference's BlitzHelpersTests.cpp:38-96 (exit code 1 on a mismatch).
#include "blitzdg/Advec1d.hpp"
#include "blitzdg/BlitzHelpers.hpp"
#include "blitzdg/MeshManager.hpp"
#include "blitzdg/Nodes1DProvisioner.hpp"
#include "blitzdg/TriangleNodesProvisioner.hpp"
#include "blitzdg/VtkOutputter.hpp"
#include <cmath>
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
    {   // binary cache: round trip, then a truncated and a corrupted copy (both must throw, under the sanitizers)
        m.writeCache("/tmp/asan_box.bdgmesh");
        MeshManager c;
        c.readCache("/tmp/asan_box.bdgmesh");
        if (c.get_NumElements() != m.get_NumElements() || c.get_EToE()[5] != m.get_EToE()[5] ||
            c.get_ElementPartitionMap().size() != m.get_ElementPartitionMap().size())
            return 3;
        std::FILE* f = std::fopen("/tmp/asan_box.bdgmesh", "rb");
        std::vector<char> raw(1 << 20);
        const size_t n = std::fread(raw.data(), 1, raw.size(), f);
        std::fclose(f);
        int refused = 0;
        for (int variant = 0; variant < 2; ++variant) {
            std::vector<char> bad(raw.begin(), raw.begin() + (variant == 0 ? n / 2 : n));
            if (variant == 1) bad[n / 3] ^= 0x21;
            f = std::fopen("/tmp/asan_bad.bdgmesh", "wb");
            std::fwrite(bad.data(), 1, bad.size(), f);
            std::fclose(f);
            try { c.readCache("/tmp/asan_bad.bdgmesh"); } catch (const std::exception&) { ++refused; }
        }
        if (refused != 2) return 4;
    }
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
    {   // reshapeMatTo1D / reshape1DToMat: Should_Convert_Full_Matrix_To_POD, Should_Convert_POD_To_Full_Matrix
        const double rowwise[25] = {2, 3, 0, 0, 0, 3, 0, 4, 0, 6, 0, -1, -3, 2, 0, 0, 0, 1, 0, 0, 0, 4, 2, 0, 1};
        const double colwise[25] = {2, 3, 0, 0, 0, 3, 0, -1, 0, 4, 0, 4, -3, 1, 2, 0, 0, 2, 0, 0, 0, 6, 0, 0, 1};
        real_matrix_type mat(5, 5), back(5, 5);
        reshape1DToMat(rowwise, mat);
        real_vector_type pod(25);
        reshapeMatTo1D(mat, pod.data());
        for (int i = 0; i < 25; ++i) if (pod(i) != rowwise[i]) return 1;
        reshapeMatTo1D(mat, pod.data(), false);
        for (int i = 0; i < 25; ++i) if (pod(i) != colwise[i]) return 1;
        reshape1DToMat(pod.data(), back, false);
        for (int i = 0; i < 5; ++i) for (int jj = 0; jj < 5; ++jj) if (back(i, jj) != mat(i, jj)) return 1;
        reshape1DToMat(rowwise, back, false);                       // column-wise read of the row-wise array = transpose
        for (int i = 0; i < 5; ++i) for (int jj = 0; jj < 5; ++jj) if (back(i, jj) != mat(jj, i)) return 1;
    }
    {   // Gauss-face and cubature contexts on a deformed copy of the mesh
        TriangleNodesProvisioner nodes(4, m2);
        const DGContext2D ctx = nodes.get_DGContext();
        real_matrix_type x = ctx.x(), y = ctx.y();
        for (index_type i = 0; i < x.rows(); ++i)
            for (index_type k = 0; k < x.cols(); ++k) y(i, k) += 0.03 * (1 - x(i, k) * x(i, k)) * (1 - y(i, k) * y(i, k));
        nodes.setCoordinates(x.data(), y.data());
        const GaussFaceContext2D g = nodes.buildGaussFaceNodes(10);
        const CubatureContext2D c = nodes.buildCubatureVolumeMesh(15);
        double area = 0;
        for (index_type i = 0; i < c.W().rows(); ++i)
            for (index_type k = 0; k < c.W().cols(); ++k) area += c.W()(i, k);
        if (g.NGauss() != 10 || c.NumCubaturePoints() != 54 || std::fabs(area - 4.0) > 1e-10) return 1; // (degree 15: the reference's 54-point rule)
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
