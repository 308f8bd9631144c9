// GaussFaceContext2D: Gauss-Legendre quadrature mesh on the faces of every triangle -- normals,
// surface Jacobians, metric terms, weights, the volume-to-Gauss interpolation matrix and the
// interior/exterior Gauss-node maps. Same accessor names as the reference's
// include/GaussFaceContext2D.hpp:68-84; built by TriangleNodesProvisioner::buildGaussFaceNodes
// (reference src/TriangleNodesProvisioner.cpp:207-381). Unlike DGContext2D this context OWNS
// its tables (the reference copies them into shared_ptrs the same way), so it stays valid after
// the provisioner's coordinates change.
#pragma once
#include "Types.hpp"
#include <memory>

namespace blitzdg {

class GaussFaceContext2D {
public:
    struct Tables {
        index_type NGauss = 0;
        real_matrix_type nx, ny, sJ, Jac, rx, ry, sx, sy, x, y, W, Interp; // (3*NGauss, K); Interp (3*NGauss, Np)
        index_hashmap bcMap;                                                // BC tag -> flat Gauss-node ids
        index_vector_type mapM, mapP;                                       // (3*NGauss*K) flat ids g + 3*NGauss*k
    };
    GaussFaceContext2D() = default;
    explicit GaussFaceContext2D(Tables&& t) : t_{std::make_shared<Tables>(std::move(t))} {}

    index_type NGauss() const { return t_->NGauss; }
    const real_matrix_type& nx() const { return t_->nx; }
    const real_matrix_type& ny() const { return t_->ny; }
    const real_matrix_type& sJ() const { return t_->sJ; }
    const real_matrix_type& Jac() const { return t_->Jac; }
    const real_matrix_type& rx() const { return t_->rx; }
    const real_matrix_type& ry() const { return t_->ry; }
    const real_matrix_type& sx() const { return t_->sx; }
    const real_matrix_type& sy() const { return t_->sy; }
    const index_hashmap& bcMap() const { return t_->bcMap; }
    const real_matrix_type& x() const { return t_->x; }
    const real_matrix_type& y() const { return t_->y; }
    const real_matrix_type& W() const { return t_->W; }
    const real_matrix_type& Interp() const { return t_->Interp; }
    const index_vector_type& mapM() const { return t_->mapM; }
    const index_vector_type& mapP() const { return t_->mapP; }

private:
    std::shared_ptr<const Tables> t_;
};

} // namespace blitzdg
