// MeshManager: reads/holds a 2-D triangle mesh and its connectivity tables.
//
// Keeps the public surface of the reference's include/MeshManager.hpp:23-232
// (readMesh, buildMesh, buildConnectivity, partitionMesh, get_* accessors) so a
// driver written against blitzdg compiles against this. Differences, all on the
// setup side of the hot path:
//  * shared-edge matching uses a sort of (vmin,vmax) edge keys instead of the
//    CXSparse product V2F^T*V2F (src/MeshManager.cpp:383-489): O(K log K), same
//    EToE/EToF on conforming meshes, works at 10^6-10^7 elements;
//  * partitionMesh uses a built-in recursive coordinate bisection (METIS is not
//    available): same outputs (0-based epart[K], npart[Nv]);
//  * buildBoxMesh builds the synthetic structured box used by the benchmark
//    configurations without going through a 60 MB ASCII file.
#pragma once
#include "Types.hpp"
#include <string>
#include <vector>

namespace blitzdg {

enum BCTag { In = 1, Out = 2, Wall = 3, Far = 4, Cyl = 5, Dirichlet = 6, Neuman = 7, Slip = 8 };

class MeshManager {
public:
    static const real_type NodeTol;

    MeshManager();
    MeshManager(const MeshManager&) = delete;
    MeshManager& operator=(const MeshManager&) = delete;
    MeshManager(MeshManager&&) = default;
    MeshManager& operator=(MeshManager&&) = default;

    /// Reads a Gmsh 2.2 ASCII .msh file (triangles; quads are rejected here).
    void readMesh(const std::string& gmshInputFile);
    /// Writes the triangles as Gmsh 2.2 ASCII (two tags per element, coordinates with 17 significant
    /// digits) -- the format readMesh takes; not in the reference, which only reads.
    void writeMesh(const std::string& gmshOutputFile) const;
    /// Binary cache of everything this class holds (vertices, EToV, EToE, EToF, BCType, partition maps): one file,
    /// little-endian, checksummed. readCache restores the object without reading ASCII or rebuilding connectivity (the
    /// step before the hot path at 10^6-10^7 elements); it refuses a file whose magic, version, sizes,
    /// index ranges or checksum do not fit. Not in the reference, which re-reads the .msh file every run.
    void writeCache(const std::string& cacheFile) const;
    void readCache(const std::string& cacheFile);
    /// Reads whitespace/comma separated vertex table (rows of Dim reals).
    void readVertices(const std::string& vertFile);
    /// Reads element-to-vertex table (rows of NumFaces 0-based vertex ids).
    void readElements(const std::string& E2VFile);
    /// Builds a mesh from in-memory tables: EToV (K x 3, 0-based), Vert (Nv x dim, dim = 2 or 3).
    /// Enforces CCW ordering, builds connectivity and the default Wall BC table
    /// (as the reference's numpy buildMesh, src/MeshManager.cpp:74-122).
    void buildMesh(const index_type* EToV, index_type K, const real_type* Vert, index_type Nv, index_type dim);
    /// Structured box [x0,x1]x[y0,y1], nx*ny cells, each split into 2 CCW triangles
    /// (K = 2*nx*ny). shuffleSeed != 0 applies a Fisher-Yates element shuffle.
    void buildBoxMesh(index_type nx, index_type ny, real_type x0, real_type x1, real_type y0, real_type y1,
                      unsigned long long shuffleSeed = 0);

    void buildConnectivity();
    void buildBCTable(index_type tagNumber);
    void partitionMesh(index_type numPartitions);

    index_type get_Dim() const { return Dim; }
    index_type get_NumVerts() const { return NumVerts; }
    index_type get_NumFaces() const { return NumFaces; }
    index_type get_NumElements() const { return NumElements; }
    const real_vector_type& get_Vertices() const { return Vert; }
    const index_vector_type& get_Elements() const { return EToV; }
    const index_vector_type& get_EToE() const { return EToE; }
    const index_vector_type& get_EToF() const { return EToF; }
    const index_vector_type& get_BCType() const { return BCType; }
    void set_BCType(const index_type* bcType, index_type n);
    const index_vector_type& get_ElementPartitionMap() const { return ElementPartitionMap; }
    const index_vector_type& get_VertexPartitionMap() const { return VertexPartitionMap; }

    void printVertices() const;
    void printElements() const;

    static index_type get_Index(index_type row, index_type col, index_type numCols) {
        return col + row * numCols;
    }

private:
    void enforceCounterClockwise();

    index_type Dim = 3, NumVerts = 0, NumFaces = 3, NumElements = 0;
    real_vector_type Vert;
    index_vector_type EToV, EToE, EToF, BCType;
    index_vector_type ElementPartitionMap, VertexPartitionMap;
};

} // namespace blitzdg
