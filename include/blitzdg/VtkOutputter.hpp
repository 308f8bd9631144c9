// VtkOutputter: writes nodal DG fields as VTK XML unstructured grids (*.vtu) for Paraview.
// Same public names as the reference's include/VtkOutputter.hpp:30-99 (generateFileName,
// writeFieldToFile, writeFieldsToFiles), but the file is produced directly -- the VTK library is not
// needed: XML header + raw appended binary blocks (Float64 points / point data, Int64
// connectivity / offsets, UInt8 cell types). As in the reference, elements of order > 1 are first
// cut into N^2 linear triangles on the equispaced lattice (TriangleNodesProvisioner::splitElements)
// and every small triangle carries its own three points.
#pragma once
#include "TriangleNodesProvisioner.hpp"
#include "Types.hpp"
#include <map>
#include <string>

namespace blitzdg {

class VtkOutputter {
public:
    explicit VtkOutputter(const TriangleNodesProvisioner& nodesProvisioner) : NodesProvisioner{nodesProvisioner} {}

    /// fieldName + 7-digit zero-padded fileNumber + ".vtu" (reference src/VtkOutputter.cpp:52-56).
    std::string generateFileName(const std::string& fieldName, index_type fileNumber) const;
    void writeFieldToFile(const std::string& fileName, const real_matrix_type& field,
                          const std::string& fieldName) const;
    void writeFieldsToFiles(const std::map<std::string, real_matrix_type>& fields, index_type tstep) const;

    /// The writer proper: `x, y, field` are (3, numTriangles) -- one column per linear triangle.
    static void writeTriangles(const std::string& fileName, const real_matrix_type& x, const real_matrix_type& y,
                               const real_matrix_type& field, const std::string& fieldName);

private:
    const TriangleNodesProvisioner& NodesProvisioner;
};

} // namespace blitzdg
