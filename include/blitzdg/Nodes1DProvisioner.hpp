// Nodes1DProvisioner: 1-D nodal DG discretisation tables (LGL nodes, V, Dr,
// Lift, grid, connectivity, vmapM/vmapP, normals, Jacobian).
// Public surface follows the reference's include/Nodes1DProvisioner.hpp:25-302;
// construction follows src/Nodes1DProvisioner.cpp:33-307.
#pragma once
#include "DenseLinAlg.hpp"
#include "JacobiBuilders.hpp"
#include "Types.hpp"

namespace blitzdg {

class Nodes1DProvisioner {
public:
    static const index_type NumFacePoints;
    static const index_type NumFaces;
    static const real_type NodeTol;

    Nodes1DProvisioner(index_type NOrder, index_type NumElements, real_type xmin, real_type xmax);
    Nodes1DProvisioner(const Nodes1DProvisioner&) = delete;
    Nodes1DProvisioner& operator=(const Nodes1DProvisioner&) = delete;
    Nodes1DProvisioner(Nodes1DProvisioner&&) = default;

    void buildNodes();
    void computeJacobian();

    void buildDr();
    void buildLift();
    void buildConnectivityMatrices();
    void buildFaceMask();
    void buildMaps();
    void buildNormals();

    index_type get_NumElements() const { return NumElements; }
    index_type get_NumLocalPoints() const { return NumLocalPoints; }
    index_type get_Order() const { return NOrder; }
    const real_matrix_type& get_xGrid() const { return xGrid; }
    const real_vector_type& get_rGrid() const { return rGrid; }
    const real_matrix_type& get_V() const { return V; }
    const real_matrix_type& get_Vinv() const { return Vinv; }
    const real_matrix_type& get_Dr() const { return Dr; }
    const real_matrix_type& get_Lift() const { return Lift; }
    const real_matrix_type& get_J() const { return J; }
    const real_matrix_type& get_rx() const { return rx; }
    const real_matrix_type& get_nx() const { return nx; }
    const index_vector_type& get_Fmask() const { return Fmask; }
    const real_matrix_type& get_Fx() const { return Fx; }
    const real_matrix_type& get_Fscale() const { return Fscale; }
    const index_matrix_type& get_EToV() const { return EToV; }
    const index_matrix_type& get_EToE() const { return EToE; }
    const index_matrix_type& get_EToF() const { return EToF; }
    const index_vector_type& get_vmapM() const { return vmapM; }
    const index_vector_type& get_vmapP() const { return vmapP; }
    index_type get_mapI() const { return mapI; }
    index_type get_mapO() const { return mapO; }
    index_type get_vmapI() const { return vmapI; }
    index_type get_vmapO() const { return vmapO; }

private:
    real_type Min_x, Max_x;
    index_type NumElements, NOrder, NumLocalPoints;
    index_type mapI, mapO, vmapI, vmapO;
    real_matrix_type xGrid;
    real_vector_type rGrid;
    real_matrix_type V, Dr, Lift, J, rx, nx, Vinv;
    index_vector_type Fmask;
    real_matrix_type Fx, Fscale;
    index_matrix_type EToV, EToE, EToF;
    index_vector_type vmapM, vmapP;
    DirectSolver LinSolver;
    JacobiBuilders Jacobi;
    VandermondeBuilders Vandermonde;
};

} // namespace blitzdg
