// VTK-free *.vtu writer (see include/blitzdg/VtkOutputter.hpp).
#include "blitzdg/VtkOutputter.hpp"
#include <cstdint>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <stdexcept>
#include <vector>

namespace blitzdg {

std::string VtkOutputter::generateFileName(const std::string& fieldName, index_type fileNumber) const {
    std::stringstream name;
    name << fieldName << std::setfill('0') << std::setw(7) << fileNumber << ".vtu";
    return name.str();
}

namespace {
template <typename T>
void appendBlock(std::ofstream& out, const std::vector<T>& data) {
    const std::uint64_t bytes = static_cast<std::uint64_t>(data.size()) * sizeof(T);
    out.write(reinterpret_cast<const char*>(&bytes), sizeof(bytes));
    out.write(reinterpret_cast<const char*>(data.data()), static_cast<std::streamsize>(bytes));
}
} // namespace

void VtkOutputter::writeTriangles(const std::string& fileName, const real_matrix_type& x, const real_matrix_type& y,
                                  const real_matrix_type& field, const std::string& fieldName) {
    const index_type nv = field.rows(), nc = field.cols();
    if (nv != 3 || x.rows() != 3 || y.rows() != 3 || x.cols() != nc || y.cols() != nc)
        throw std::runtime_error("VtkOutputter: expected (3, numTriangles) arrays");
    const std::uint64_t numPoints = static_cast<std::uint64_t>(3) * nc, numCells = static_cast<std::uint64_t>(nc);
    std::vector<double> values(numPoints), points(3 * numPoints);
    std::vector<std::int64_t> conn(numPoints), offsets(numCells);
    std::vector<std::uint8_t> types(numCells, 5); // VTK_TRIANGLE
    for (index_type k = 0; k < nc; ++k) {
        for (index_type n = 0; n < 3; ++n) {
            const std::uint64_t id = static_cast<std::uint64_t>(3) * k + n;
            values[id] = field(n, k);
            points[3 * id] = x(n, k);
            points[3 * id + 1] = y(n, k);
            points[3 * id + 2] = 0.0;
            conn[id] = static_cast<std::int64_t>(id);
        }
        offsets[k] = static_cast<std::int64_t>(3) * (k + 1);
    }
    std::ofstream out(fileName, std::ios::binary);
    if (!out) throw std::runtime_error("VtkOutputter: cannot open " + fileName);
    std::uint64_t off = 0;
    auto next = [&off](std::uint64_t bytes) { const std::uint64_t o = off; off += 8 + bytes; return o; };
    const std::uint64_t oVal = next(values.size() * 8), oPts = next(points.size() * 8), oConn = next(conn.size() * 8),
                        oOff = next(offsets.size() * 8), oTyp = next(types.size());
    out << "<?xml version=\"1.0\"?>\n"
        << "<VTKFile type=\"UnstructuredGrid\" version=\"1.0\" byte_order=\"LittleEndian\" header_type=\"UInt64\">\n"
        << "  <UnstructuredGrid>\n"
        << "    <Piece NumberOfPoints=\"" << numPoints << "\" NumberOfCells=\"" << numCells << "\">\n"
        << "      <PointData Scalars=\"" << fieldName << "\">\n"
        << "        <DataArray type=\"Float64\" Name=\"" << fieldName << "\" format=\"appended\" offset=\"" << oVal << "\"/>\n"
        << "      </PointData>\n"
        << "      <Points>\n"
        << "        <DataArray type=\"Float64\" NumberOfComponents=\"3\" format=\"appended\" offset=\"" << oPts << "\"/>\n"
        << "      </Points>\n"
        << "      <Cells>\n"
        << "        <DataArray type=\"Int64\" Name=\"connectivity\" format=\"appended\" offset=\"" << oConn << "\"/>\n"
        << "        <DataArray type=\"Int64\" Name=\"offsets\" format=\"appended\" offset=\"" << oOff << "\"/>\n"
        << "        <DataArray type=\"UInt8\" Name=\"types\" format=\"appended\" offset=\"" << oTyp << "\"/>\n"
        << "      </Cells>\n"
        << "    </Piece>\n"
        << "  </UnstructuredGrid>\n"
        << "  <AppendedData encoding=\"raw\">\n   _";
    appendBlock(out, values);
    appendBlock(out, points);
    appendBlock(out, conn);
    appendBlock(out, offsets);
    appendBlock(out, types);
    out << "\n  </AppendedData>\n</VTKFile>\n";
    if (!out) throw std::runtime_error("VtkOutputter: write failed for " + fileName);
}

void VtkOutputter::writeFieldToFile(const std::string& fileName, const real_matrix_type& field,
                                    const std::string& fieldName) const {
    const real_matrix_type& x = NodesProvisioner.get_xGrid();
    const real_matrix_type& y = NodesProvisioner.get_yGrid();
    if (NodesProvisioner.get_NOrder() > 1) { // higher order than linear: break up the triangles
        real_matrix_type xnew, ynew, fieldnew;
        NodesProvisioner.splitElements(x, y, field, xnew, ynew, fieldnew);
        writeTriangles(fileName, xnew, ynew, fieldnew, fieldName);
    } else {
        writeTriangles(fileName, x, y, field, fieldName);
    }
}

void VtkOutputter::writeFieldsToFiles(const std::map<std::string, real_matrix_type>& fields, index_type tstep) const {
    for (const auto& kv : fields) writeFieldToFile(generateFileName(kv.first, tstep), kv.second, kv.first);
}

} // namespace blitzdg
