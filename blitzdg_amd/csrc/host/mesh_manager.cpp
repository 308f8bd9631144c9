// MeshManager implementation (setup path; CPU only).
// Behaviour follows the reference's src/MeshManager.cpp: Gmsh 2.2 ASCII parsing
// (:130-251), CCW enforcement (:292-306), shared-edge connectivity with
// self-connected boundary faces (:383-489), default Wall BC table (:315-329).
#include "blitzdg/MeshManager.hpp"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <numeric>
#include <sstream>
#include <stdexcept>

namespace blitzdg {

const real_type MeshManager::NodeTol = 1.e-5;

MeshManager::MeshManager() = default;

namespace {

bool nextLine(std::istream& in, std::string& line) {
    if (!std::getline(in, line)) return false;
    while (!line.empty() && (line.back() == '\r' || line.back() == ' ' || line.back() == '\t')) line.pop_back();
    return true;
}

void requireLine(std::istream& in, std::string& line, const char* what) {
    if (!nextLine(in, line)) throw std::runtime_error(std::string("Unexpected end of .msh file while reading ") + what);
}

// Splits on blanks, tabs and commas (the reference's CSV reader delimiters).
void tokenize(const std::string& line, std::vector<std::string>& out) {
    out.clear();
    std::size_t i = 0, n = line.size();
    while (i < n) {
        while (i < n && (line[i] == ' ' || line[i] == '\t' || line[i] == ',')) ++i;
        std::size_t j = i;
        while (j < n && !(line[j] == ' ' || line[j] == '\t' || line[j] == ',')) ++j;
        if (j > i) out.emplace_back(line, i, j - i);
        i = j;
    }
}

template <typename T>
std::vector<T> readTable(const std::string& path, index_type& rows, index_type& cols) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("Unable to open file: " + path);
    std::vector<T> vals;
    std::string line;
    std::vector<std::string> tok;
    rows = 0; cols = 0;
    while (nextLine(in, line)) {
        tokenize(line, tok);
        if (tok.empty()) continue;
        if (cols == 0) cols = static_cast<index_type>(tok.size());
        if (static_cast<index_type>(tok.size()) != cols)
            throw std::runtime_error("Inconsistent number of columns in file: " + path);
        for (const auto& s : tok) vals.push_back(static_cast<T>(std::stod(s)));
        ++rows;
    }
    return vals;
}

} // namespace

void MeshManager::readMesh(const std::string& gmshInputFile) {
    std::ifstream in(gmshInputFile);
    if (!in) throw std::runtime_error("Unable to open mesh file: " + gmshInputFile);

    std::string line;
    std::vector<std::string> tok;

    requireLine(in, line, "header");
    if (line != "$MeshFormat") throw std::runtime_error("Missing $MeshFormat header in .msh file!");
    requireLine(in, line, "format line");
    tokenize(line, tok);
    if (tok.size() < 3) throw std::runtime_error("Malformed $MeshFormat line in .msh file!");
    const float vers = std::stof(tok[0]);
    const int fileType = std::stoi(tok[1]), floatSize = std::stoi(tok[2]);
    if (vers < 2.0 || vers >= 3.0)
        throw std::runtime_error("Unsupported Gmsh version. Only 2.x is currently supported.");
    if (fileType != 0) throw std::runtime_error("Only ASCII-type Gmsh formats are supported.");
    if (floatSize != 8) throw std::runtime_error("Only 8-byte reals in Gmsh files are supported!");
    requireLine(in, line, "$EndMeshFormat");

    requireLine(in, line, "$Nodes");
    if (line != "$Nodes")
        throw std::runtime_error("Unexpected line marker in .msh file! Expected '$Nodes' but was:" + line + ".");
    requireLine(in, line, "node count");
    NumVerts = std::stoi(line);
    Dim = 3;
    Vert.resize(NumVerts * Dim);
    for (index_type i = 0; i < NumVerts; ++i) {
        requireLine(in, line, "nodes");
        tokenize(line, tok);
        if (tok.size() < 4) throw std::runtime_error("Malformed node row in .msh file!");
        const index_type id = std::stoi(tok[0]);
        if (id < 1 || id > NumVerts) throw std::runtime_error("Node number out of range in .msh file!");
        Vert((id - 1) * Dim) = std::stod(tok[1]);
        Vert((id - 1) * Dim + 1) = std::stod(tok[2]);
        Vert((id - 1) * Dim + 2) = std::stod(tok[3]);
    }
    requireLine(in, line, "$EndNodes");

    requireLine(in, line, "$Elements");
    if (line != "$Elements")
        throw std::runtime_error("Unexpected line marker in .msh file! Expected '$Elements' but was:" + line + ".");
    requireLine(in, line, "element count");
    const index_type numRows = std::stoi(line);

    std::vector<index_type> tris;
    tris.reserve(static_cast<std::size_t>(numRows) * 3);
    bool sawQuads = false;
    for (index_type i = 0; i < numRows; ++i) {
        requireLine(in, line, "elements");
        tokenize(line, tok);
        if (tok.size() < 4) throw std::runtime_error("Malformed element row in .msh file!");
        const int elemType = std::stoi(tok[1]), numTags = std::stoi(tok[2]);
        const int nLocal = static_cast<int>(tok.size()) - numTags - 3;
        if (nLocal == 1 && elemType != 15) throw std::runtime_error("Incorrect Element Type for point element!");
        if (nLocal == 2 && elemType != 1) throw std::runtime_error("Incorrect Element Type for line element!");
        if (nLocal == 3) {
            if (elemType != 2) throw std::runtime_error("Incorrect Element Type for triangle element!");
            // The reference reads node ids at token positions 5,6,7 (two tags assumed).
            if (tok.size() < 8) throw std::runtime_error("Triangle row with fewer than two tags is not supported!");
            for (int v = 5; v <= 7; ++v) tris.push_back(std::stoi(tok[v]) - 1);
        }
        if (nLocal == 4) {
            if (elemType != 3) throw std::runtime_error("Incorrect Element Type for quadrangle element!");
            sawQuads = true;
        }
    }
    if (sawQuads)
        throw std::runtime_error("Quadrangle meshes are outside the scope of the MI355X sw2d path (triangles only).");

    NumFaces = 3;
    NumElements = static_cast<index_type>(tris.size() / 3);
    EToV.resize(NumElements * NumFaces);
    std::copy(tris.begin(), tris.end(), EToV.begin());
    for (index_type i = 0; i < EToV.size(); ++i)
        if (EToV(i) < 0 || EToV(i) >= NumVerts) throw std::runtime_error("Element references a vertex out of range!");

    enforceCounterClockwise();
    buildConnectivity();
    buildBCTable(BCTag::Wall);
}

void MeshManager::writeMesh(const std::string& gmshOutputFile) const {
    std::FILE* f = std::fopen(gmshOutputFile.c_str(), "w");
    if (!f) throw std::runtime_error("Unable to open mesh file for writing: " + gmshOutputFile);
    std::fprintf(f, "$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n", NumVerts);
    for (index_type v = 0; v < NumVerts; ++v)
        std::fprintf(f, "%d %.17g %.17g %.17g\n", v + 1, Vert(v * Dim), Vert(v * Dim + 1), Dim > 2 ? Vert(v * Dim + 2) : 0.0);
    std::fprintf(f, "$EndNodes\n$Elements\n%d\n", NumElements);
    for (index_type k = 0; k < NumElements; ++k)
        std::fprintf(f, "%d 2 2 1 1 %d %d %d\n", k + 1, EToV(3 * k) + 1, EToV(3 * k + 1) + 1, EToV(3 * k + 2) + 1);
    std::fprintf(f, "$EndElements\n");
    if (std::fclose(f) != 0) throw std::runtime_error("Write failed for mesh file: " + gmshOutputFile);
}

namespace {

constexpr char kCacheMagic[8] = {'B', 'D', 'G', 'M', 'E', 'S', 'H', '1'};
constexpr std::uint32_t kCacheVersion = 1;

struct CacheHeader {
    char magic[8];
    std::uint32_t version, dim, numFaces, reserved;
    std::uint64_t numVerts, numElements;
    std::uint64_t nVert, nEToV, nEToE, nEToF, nBC, nEPart, nVPart; // entries of each table
};

std::uint64_t fnv1a(std::uint64_t h, const void* data, std::size_t bytes) {
    const unsigned char* p = static_cast<const unsigned char*>(data);
    for (std::size_t i = 0; i < bytes; ++i) { h ^= p[i]; h *= 1099511628211ULL; }
    return h;
}

template <typename V>
void putTable(std::FILE* f, const V& v, std::uint64_t& sum) {
    const std::size_t bytes = static_cast<std::size_t>(v.size()) * sizeof(v[0]);
    if (bytes && std::fwrite(&v[0], 1, bytes, f) != bytes) throw std::runtime_error("mesh cache: short write");
    if (bytes) sum = fnv1a(sum, &v[0], bytes);
}

template <typename V>
void getTable(std::FILE* f, V& v, std::uint64_t n, std::uint64_t& sum) {
    if (n > 2000000000ULL) throw std::runtime_error("mesh cache: table size out of range");
    v.resizeUninitialized(static_cast<index_type>(n));
    const std::size_t bytes = static_cast<std::size_t>(n) * sizeof(v[0]);
    if (bytes && std::fread(&v[0], 1, bytes, f) != bytes) throw std::runtime_error("mesh cache: file is truncated");
    if (bytes) sum = fnv1a(sum, &v[0], bytes);
}

} // namespace

void MeshManager::writeCache(const std::string& cacheFile) const {
    std::FILE* f = std::fopen(cacheFile.c_str(), "wb");
    if (!f) throw std::runtime_error("Unable to open mesh cache for writing: " + cacheFile);
    CacheHeader h{};
    std::memcpy(h.magic, kCacheMagic, 8);
    h.version = kCacheVersion; h.dim = static_cast<std::uint32_t>(Dim); h.numFaces = static_cast<std::uint32_t>(NumFaces);
    h.numVerts = static_cast<std::uint64_t>(NumVerts); h.numElements = static_cast<std::uint64_t>(NumElements);
    h.nVert = Vert.size(); h.nEToV = EToV.size(); h.nEToE = EToE.size(); h.nEToF = EToF.size(); h.nBC = BCType.size();
    h.nEPart = ElementPartitionMap.size(); h.nVPart = VertexPartitionMap.size();
    std::uint64_t sum = fnv1a(14695981039346656037ULL, &h, sizeof h);
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1;
    try {
        if (!ok) throw std::runtime_error("mesh cache: short write");
        putTable(f, Vert, sum); putTable(f, EToV, sum); putTable(f, EToE, sum); putTable(f, EToF, sum);
        putTable(f, BCType, sum); putTable(f, ElementPartitionMap, sum); putTable(f, VertexPartitionMap, sum);
        if (std::fwrite(&sum, sizeof sum, 1, f) != 1) throw std::runtime_error("mesh cache: short write");
    } catch (...) {
        std::fclose(f);
        throw;
    }
    if (std::fclose(f) != 0) throw std::runtime_error("Write failed for mesh cache: " + cacheFile);
}

void MeshManager::readCache(const std::string& cacheFile) {
    std::FILE* f = std::fopen(cacheFile.c_str(), "rb");
    if (!f) throw std::runtime_error("Unable to open mesh cache: " + cacheFile);
    MeshManager m; // filled completely and checked before this object is touched
    try {
        CacheHeader h{};
        if (std::fread(&h, sizeof h, 1, f) != 1 || std::memcmp(h.magic, kCacheMagic, 8) != 0)
            throw std::runtime_error("not a blitzdg mesh cache: " + cacheFile);
        if (h.version != kCacheVersion) throw std::runtime_error("mesh cache: unsupported version");
        if (h.numFaces != 3 || (h.dim != 2 && h.dim != 3) || h.numVerts < 3 || h.numVerts > 2000000000ULL || h.numElements < 1 ||
            h.numElements > 600000000ULL)
            throw std::runtime_error("mesh cache: header out of range");
        const std::uint64_t K = h.numElements, faces = 3 * K;
        if (h.nVert != h.numVerts * h.dim || h.nEToV != faces || h.nEToE != faces || h.nEToF != faces || h.nBC != faces ||
            (h.nEPart != 0 && h.nEPart != K) || (h.nVPart != 0 && h.nVPart != h.numVerts))
            throw std::runtime_error("mesh cache: table sizes do not fit the header");
        std::uint64_t sum = fnv1a(14695981039346656037ULL, &h, sizeof h), stored = 0;
        m.Dim = static_cast<index_type>(h.dim); m.NumFaces = 3;
        m.NumVerts = static_cast<index_type>(h.numVerts); m.NumElements = static_cast<index_type>(K);
        getTable(f, m.Vert, h.nVert, sum); getTable(f, m.EToV, h.nEToV, sum); getTable(f, m.EToE, h.nEToE, sum);
        getTable(f, m.EToF, h.nEToF, sum); getTable(f, m.BCType, h.nBC, sum);
        getTable(f, m.ElementPartitionMap, h.nEPart, sum); getTable(f, m.VertexPartitionMap, h.nVPart, sum);
        if (std::fread(&stored, sizeof stored, 1, f) != 1) throw std::runtime_error("mesh cache: file is truncated");
        if (stored != sum) throw std::runtime_error("mesh cache: checksum mismatch (corrupted file): " + cacheFile);
        for (std::uint64_t i = 0; i < faces; ++i) {
            const index_type v = m.EToV[static_cast<index_type>(i)], e = m.EToE[static_cast<index_type>(i)], ff = m.EToF[static_cast<index_type>(i)];
            if (v < 0 || v >= m.NumVerts || e < 0 || e >= m.NumElements || ff < 0 || ff > 2)
                throw std::runtime_error("mesh cache: index out of range");
        }
    } catch (...) {
        std::fclose(f);
        throw;
    }
    std::fclose(f);
    *this = std::move(m);
}

void MeshManager::readVertices(const std::string& vertFile) {
    index_type rows, cols;
    auto vals = readTable<real_type>(vertFile, rows, cols);
    NumVerts = rows; Dim = cols;
    Vert.resize(rows * cols);
    std::copy(vals.begin(), vals.end(), Vert.begin());
}

void MeshManager::readElements(const std::string& E2VFile) {
    index_type rows, cols;
    auto vals = readTable<index_type>(E2VFile, rows, cols);
    NumElements = rows; NumFaces = cols;
    EToV.resize(rows * cols);
    std::copy(vals.begin(), vals.end(), EToV.begin());
    BCType.resize(NumElements * NumFaces);
    EToE.resize(NumElements * NumFaces);
    EToF.resize(NumElements * NumFaces);
    if (NumFaces == 3) {
        buildConnectivity();
        buildBCTable(BCTag::Wall);
    }
}

void MeshManager::buildMesh(const index_type* e2v, index_type K, const real_type* vert, index_type Nv, index_type dim) {
    if (dim != 2 && dim != 3) throw std::runtime_error("buildMesh: vertices must have 2 or 3 coordinates");
    // Vertices are always stored (x,y,z)-interleaved: the nodes provisioner
    // indexes them with stride 3 (reference src/TriangleNodesProvisioner.cpp:755-760).
    Dim = 3; NumVerts = Nv; NumFaces = 3; NumElements = K;
    Vert.resize(Nv * 3);
    for (index_type i = 0; i < Nv; ++i) {
        Vert(3 * i) = vert[dim * i];
        Vert(3 * i + 1) = vert[dim * i + 1];
        Vert(3 * i + 2) = dim == 3 ? vert[dim * i + 2] : 0.0;
    }
    EToV.resize(K * 3);
    for (index_type i = 0; i < K * 3; ++i) {
        if (e2v[i] < 0 || e2v[i] >= Nv) throw std::runtime_error("buildMesh: vertex index out of range");
        EToV(i) = e2v[i];
    }
    enforceCounterClockwise();
    buildConnectivity();
    buildBCTable(BCTag::Wall);
}

void MeshManager::buildBoxMesh(index_type nx, index_type ny, real_type x0, real_type x1, real_type y0, real_type y1,
                               unsigned long long shuffleSeed) {
    if (nx < 1 || ny < 1) throw std::runtime_error("buildBoxMesh: need at least one cell per direction");
    const long long Kll = 2LL * nx * ny;
    if (Kll > 100000000LL) throw std::runtime_error("buildBoxMesh: mesh too large for int32 indexing");
    Dim = 3; NumFaces = 3;
    NumVerts = (nx + 1) * (ny + 1);
    NumElements = static_cast<index_type>(Kll);
    Vert.resize(NumVerts * 3);
    for (index_type j = 0; j <= ny; ++j)
        for (index_type i = 0; i <= nx; ++i) {
            const index_type v = j * (nx + 1) + i;
            Vert(3 * v) = x0 + (x1 - x0) * i / nx;
            Vert(3 * v + 1) = y0 + (y1 - y0) * j / ny;
            Vert(3 * v + 2) = 0.0;
        }
    std::vector<index_type> order(NumElements);
    std::iota(order.begin(), order.end(), 0);
    if (shuffleSeed != 0) {
        std::uint64_t s = shuffleSeed;
        auto next = [&s]() { // splitmix64
            std::uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            return z ^ (z >> 31);
        };
        for (index_type i = NumElements - 1; i > 0; --i) {
            const index_type j = static_cast<index_type>(next() % static_cast<std::uint64_t>(i + 1));
            std::swap(order[i], order[j]);
        }
    }
    EToV.resize(NumElements * 3);
    for (index_type slot = 0; slot < NumElements; ++slot) {
        const index_type e = order[slot];
        const index_type cell = e / 2, i = cell % nx, j = cell / nx;
        const index_type v00 = j * (nx + 1) + i, v10 = v00 + 1, v01 = v00 + (nx + 1), v11 = v01 + 1;
        if (e % 2 == 0) { EToV(3 * slot) = v00; EToV(3 * slot + 1) = v10; EToV(3 * slot + 2) = v11; }
        else            { EToV(3 * slot) = v00; EToV(3 * slot + 1) = v11; EToV(3 * slot + 2) = v01; }
    }
    enforceCounterClockwise();
    buildConnectivity();
    buildBCTable(BCTag::Wall);
}

void MeshManager::enforceCounterClockwise() {
    for (index_type k = 0; k < NumElements; ++k) {
        const index_type a = EToV(NumFaces * k), b = EToV(NumFaces * k + 1), c = EToV(NumFaces * k + 2);
        const real_type ax = Vert(a * Dim), ay = Vert(a * Dim + 1);
        const real_type bx = Vert(b * Dim), by = Vert(b * Dim + 1);
        const real_type cx = Vert(c * Dim), cy = Vert(c * Dim + 1);
        const real_type det = (ax - cx) * (by - cy) - (bx - cx) * (ay - cy);
        if (det < 0) std::swap(EToV(NumFaces * k + 1), EToV(NumFaces * k + 2));
    }
}

void MeshManager::buildConnectivity() {
    if (NumFaces != 3) throw std::runtime_error("buildConnectivity: triangles only");
    const index_type totalFaces = NumFaces * NumElements;
    EToE.resize(totalFaces);
    EToF.resize(totalFaces);
    // Face f of element k joins local vertices (f, (f+1)%3); two faces are
    // connected iff they share both vertices. Boundary faces stay self-connected.
    std::vector<std::pair<std::uint64_t, index_type>> keys(static_cast<std::size_t>(totalFaces));
    for (index_type k = 0; k < NumElements; ++k)
        for (index_type f = 0; f < NumFaces; ++f) {
            const index_type g = k * NumFaces + f;
            EToE(g) = k;
            EToF(g) = f;
            const std::uint64_t v1 = static_cast<std::uint64_t>(EToV(k * NumFaces + f));
            const std::uint64_t v2 = static_cast<std::uint64_t>(EToV(k * NumFaces + (f + 1) % NumFaces));
            keys[g] = { (std::min(v1, v2) << 32) | std::max(v1, v2), g };
        }
    std::sort(keys.begin(), keys.end());
    for (std::size_t i = 0; i + 1 < keys.size();) {
        std::size_t j = i + 1;
        while (j < keys.size() && keys[j].first == keys[i].first) ++j;
        // Conforming meshes have runs of length 1 (boundary) or 2 (interior).
        for (std::size_t a = i; a < j; ++a)
            for (std::size_t b = i; b < j; ++b)
                if (a != b) {
                    const index_type ga = keys[a].second, gb = keys[b].second;
                    EToE(ga) = gb / NumFaces;
                    EToF(ga) = gb % NumFaces;
                }
        i = j;
    }
}

void MeshManager::buildBCTable(index_type tagNumber) {
    BCType.resize(NumFaces * NumElements);
    for (index_type f = 0; f < NumFaces * NumElements; ++f)
        BCType(f) = (EToE(f) == f / NumFaces) ? tagNumber : 0;
}

void MeshManager::set_BCType(const index_type* bcType, index_type n) {
    if (n != NumFaces * NumElements) throw std::runtime_error("set_BCType: expected NumElements*NumFaces entries");
    for (index_type i = 0; i < n; ++i) BCType(i) = bcType[i];
}

namespace {
// Recursive coordinate bisection over element centroids: splits `ids` into
// `parts` pieces with sizes proportional to floor/ceil halves.
void rcb(std::vector<index_type>& ids, std::size_t lo, std::size_t hi, index_type firstPart, index_type parts,
         const std::vector<real_type>& cx, const std::vector<real_type>& cy, index_vector_type& epart) {
    if (parts <= 1) {
        for (std::size_t i = lo; i < hi; ++i) epart(ids[i]) = firstPart;
        return;
    }
    real_type xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
    for (std::size_t i = lo; i < hi; ++i) {
        xmin = std::min(xmin, cx[ids[i]]); xmax = std::max(xmax, cx[ids[i]]);
        ymin = std::min(ymin, cy[ids[i]]); ymax = std::max(ymax, cy[ids[i]]);
    }
    const bool splitX = (xmax - xmin) >= (ymax - ymin);
    const index_type leftParts = parts / 2;
    const std::size_t mid = lo + (hi - lo) * static_cast<std::size_t>(leftParts) / static_cast<std::size_t>(parts);
    const auto& c = splitX ? cx : cy;
    const auto& d = splitX ? cy : cx;
    std::nth_element(ids.begin() + lo, ids.begin() + mid, ids.begin() + hi, [&](index_type a, index_type b) {
        if (c[a] != c[b]) return c[a] < c[b];
        if (d[a] != d[b]) return d[a] < d[b];
        return a < b;
    });
    rcb(ids, lo, mid, firstPart, leftParts, cx, cy, epart);
    rcb(ids, mid, hi, firstPart + leftParts, parts - leftParts, cx, cy, epart);
}
} // namespace

void MeshManager::partitionMesh(index_type numPartitions) {
    if (numPartitions < 1) throw std::runtime_error("partitionMesh: numPartitions must be >= 1");
    ElementPartitionMap.resize(NumElements);
    VertexPartitionMap.resize(NumVerts);
    std::vector<real_type> cx(NumElements), cy(NumElements);
    for (index_type k = 0; k < NumElements; ++k) {
        real_type sx = 0, sy = 0;
        for (index_type v = 0; v < NumFaces; ++v) {
            sx += Vert(EToV(k * NumFaces + v) * Dim);
            sy += Vert(EToV(k * NumFaces + v) * Dim + 1);
        }
        cx[k] = sx / NumFaces; cy[k] = sy / NumFaces;
    }
    std::vector<index_type> ids(NumElements);
    std::iota(ids.begin(), ids.end(), 0);
    rcb(ids, 0, ids.size(), 0, numPartitions, cx, cy, ElementPartitionMap);
    // A vertex belongs to the part of the lowest-numbered element touching it.
    VertexPartitionMap.fill(-1);
    for (index_type k = 0; k < NumElements; ++k)
        for (index_type v = 0; v < NumFaces; ++v) {
            index_type& p = VertexPartitionMap(EToV(k * NumFaces + v));
            if (p < 0) p = ElementPartitionMap(k);
        }
    for (index_type v = 0; v < NumVerts; ++v)
        if (VertexPartitionMap(v) < 0) VertexPartitionMap(v) = 0;
}

void MeshManager::printVertices() const {
    for (index_type i = 0; i < NumVerts; ++i) {
        for (index_type d = 0; d < Dim; ++d) std::cout << Vert(i * Dim + d) << " ";
        std::cout << "\n";
    }
}

void MeshManager::printElements() const {
    for (index_type k = 0; k < NumElements; ++k) {
        for (index_type f = 0; f < NumFaces; ++f) std::cout << EToV(k * NumFaces + f) << " ";
        std::cout << "\n";
    }
}

} // namespace blitzdg
