// TriangleNodesProvisioner: builds every table of the 2-D triangular nodal DG
// discretisation (nodes, V/Dr/Ds/Lift/Filter, physical grid and metric terms,
// normals/Fscale, vmapM/vmapP/BC maps) and hands them out as a DGContext2D.
//
// Public surface follows the reference's include/TriangleNodesProvisioner.hpp
// :32-427; construction order follows src/TriangleNodesProvisioner.cpp:39-79
// (buildNodes -> buildLift -> buildPhysicalGrid -> buildMaps). The object is
// movable, not copyable, and owns its tables; the RHS evaluators only read them.
#pragma once
#include "CubatureContext2D.hpp"
#include "DGContext2D.hpp"
#include "DenseLinAlg.hpp"
#include "GaussFaceContext2D.hpp"
#include "JacobiBuilders.hpp"
#include "MeshManager.hpp"
#include "Types.hpp"
#include <memory>
#include <vector>

namespace blitzdg {

class TriangleNodesProvisioner {
public:
    static const index_type NumFaces;
    static const real_type NodeTol;

    TriangleNodesProvisioner(index_type NOrder, const MeshManager& meshManager);
    TriangleNodesProvisioner(const TriangleNodesProvisioner&) = delete;
    TriangleNodesProvisioner& operator=(const TriangleNodesProvisioner&) = delete;
    TriangleNodesProvisioner(TriangleNodesProvisioner&&) = default;

    // ---- reference-element helpers (public in the reference, used by its tests)
    void evaluateSimplexPolynomial(const real_vector_type& a, const real_vector_type& b, index_type i,
                                   index_type j, real_vector_type& p) const;
    void evaluateGradSimplex(const real_vector_type& a, const real_vector_type& b, index_type id,
                             index_type jd, real_vector_type& dpdr, real_vector_type& dpds) const;
    void rsToab(const real_vector_type& r, const real_vector_type& s, real_vector_type& a,
                real_vector_type& b) const;
    void xyTors(const real_vector_type& x, const real_vector_type& y, real_vector_type& r,
                real_vector_type& s) const;
    void computeVandermondeMatrix(index_type N, const real_vector_type& r, const real_vector_type& s,
                                  real_matrix_type& V) const;
    void computeGradVandermondeMatrix(index_type N, const real_vector_type& r, const real_vector_type& s,
                                      real_matrix_type& V2Dr, real_matrix_type& V2Ds) const;
    void computeDifferentiationMatrices(const real_matrix_type& V2Dr, const real_matrix_type& V2Ds,
                                        const real_matrix_type& V, const real_matrix_type& Vc,
                                        real_matrix_type& Dr, real_matrix_type& Ds, real_matrix_type& Drw,
                                        real_matrix_type& Dsw) const;
    void computeEquilateralNodes(real_vector_type& x, real_vector_type& y) const;
    void computeWarpFactor(const real_vector_type& r, real_vector_type& warpFactor) const;
    void computeInterpMatrix(const real_vector_type& rout, const real_vector_type& sout,
                             real_matrix_type& IM) const;
    /// Output step (reference src/TriangleNodesProvisioner.cpp:1154-1264): interpolate a nodal
    /// field to the equispaced lattice of its element and cut the element into N^2 linear
    /// triangles; xnew, ynew, fieldnew become (3, N^2*K), one column per small triangle.
    void splitElements(const real_matrix_type& x, const real_matrix_type& y, const real_matrix_type& field,
                       real_matrix_type& xnew, real_matrix_type& ynew, real_matrix_type& fieldnew) const;
    /// The pieces of splitElements: equispaced-lattice interpolation matrix (Np, Np) and the local
    /// connectivity of the N^2 small triangles (lattice point indices, 3 per triangle).
    void splitOperators(real_matrix_type& IM, std::vector<index_type>& localE2V) const;

    // ---- build steps
    void buildNodes();
    void buildLift();
    void buildPhysicalGrid();
    void buildMaps();
    /// Appends the face-node lists of the mesh's BC table to BCmap.
    void buildBCHash();
    /// As the reference (src/TriangleNodesProvisioner.cpp:1028-1057) this APPENDS to
    /// the existing map: earlier entries are never cleared.
    void buildBCHash(const index_vector_type& bcType);
    void buildFilter(real_type Nc, index_type s);
    /// Overwrites the physical node coordinates (curved meshes); geometry is NOT
    /// rebuilt, as in the reference's setCoordinates_numpy (:1266-1272).
    void setCoordinates(const real_type* x, const real_type* y);
    /// Gauss-Legendre quadrature mesh of order NGauss on every face (reference
    /// src/TriangleNodesProvisioner.cpp:207-381), from the CURRENT node coordinates.
    GaussFaceContext2D buildGaussFaceNodes(index_type NGauss);
    /// Volume cubature mesh exact to degree NCubature (reference :81-205); as there, the nodal J, rx,
    /// ry, sx, sy of this provisioner are recomputed from the current node coordinates as a side effect.
    CubatureContext2D buildCubatureVolumeMesh(index_type NCubature);

    // ---- accessors
    const real_matrix_type& get_xGrid() const { return xGrid; }
    const real_matrix_type& get_yGrid() const { return yGrid; }
    const real_vector_type& get_rGrid() const { return rGrid; }
    const real_vector_type& get_sGrid() const { return sGrid; }
    const real_matrix_type& get_Dr() const { return Dr; }
    const real_matrix_type& get_Ds() const { return Ds; }
    const real_matrix_type& get_Drw() const { return Drw; }
    const real_matrix_type& get_Dsw() const { return Dsw; }
    const real_matrix_type& get_V() const { return V; }
    const real_matrix_type& get_Vinv() const { return Vinv; }
    const real_matrix_type& get_Filter() const { return Filter; }
    const real_matrix_type& get_J() const { return J; }
    const real_matrix_type& get_rx() const { return rx; }
    const real_matrix_type& get_ry() const { return ry; }
    const real_matrix_type& get_sx() const { return sx; }
    const real_matrix_type& get_sy() const { return sy; }
    const real_matrix_type& get_nx() const { return nx; }
    const real_matrix_type& get_ny() const { return ny; }
    const index_matrix_type& get_Fmask() const { return Fmask; }
    const real_matrix_type& get_Fx() const { return Fx; }
    const real_matrix_type& get_Fy() const { return Fy; }
    const real_matrix_type& get_Fscale() const { return Fscale; }
    const real_matrix_type& get_Lift() const { return Lift; }
    const MeshManager& get_MeshManager() const { return *Mesh2D; }
    const index_vector_type& get_vmapM() const { return vmapM; }
    const index_vector_type& get_vmapP() const { return vmapP; }
    const index_vector_type& get_mapP() const { return mapP; }
    const index_vector_type& get_vmapB() const { return vmapB; }
    const index_vector_type& get_mapB() const { return mapB; }
    const index_hashmap& get_bcMap() const { return BCmap; }
    /// Unique-node gather/scatter maps (reference buildMaps :1009-1019); built on
    /// first use because the sort over all Np*K nodes is not needed by the RHS path.
    const std::vector<index_type>& get_gather() const;
    const std::vector<index_type>& get_scatter() const;
    DGContext2D get_DGContext() const;

    index_type get_NumLocalPoints() const { return NumLocalPoints; }
    index_type get_NumFacePoints() const { return NumFacePoints; }
    index_type get_NumElements() const { return NumElements; }
    index_type get_NOrder() const { return NOrder; }

private:
    void buildGatherScatter() const;

    index_type NumElements, NOrder, NumLocalPoints, NumFacePoints;
    real_matrix_type xGrid, yGrid;
    real_vector_type rGrid, sGrid;
    real_matrix_type V, Dr, Ds, Drw, Dsw, Lift, J, rx, sx, ry, sy, nx, ny, Vinv, Filter;
    index_matrix_type Fmask;
    real_matrix_type Fscale, Fx, Fy;
    index_vector_type vmapM, vmapP, mapP, vmapB, mapB;
    index_hashmap BCmap;
    mutable std::vector<index_type> gatherVec, scatterVec;
    mutable bool gatherBuilt = false;
    const MeshManager* Mesh2D;
    JacobiBuilders Jacobi;
    VandermondeBuilders Vandermonde;
    DirectSolver LinSolver;
    DenseMatrixInverter Inverter;
};

} // namespace blitzdg
