// CubatureContext2D: volume cubature mesh of every triangle -- cubature points and weights on
// the reference element, interpolation / differentiation matrices from the nodal set to the
// cubature points, metric terms and Jacobian at the cubature points, the per-element cubature mass
// matrix and its upper Cholesky factor. Same accessor names as the reference's
// include/CubatureContext2D.hpp:75-96; built by TriangleNodesProvisioner::buildCubatureVolumeMesh
// (reference src/TriangleNodesProvisioner.cpp:81-205). Owns its tables.
#pragma once
#include "Types.hpp"
#include <memory>

namespace blitzdg {

/// Dense 3-D array (n0, n1, n2), last index contiguous -- the reference's real_tensor3_type as it is
/// used for MM / MMChol: (Np, Np, K), element index fastest.
template <typename T>
class tensor3_type {
public:
    tensor3_type() = default;
    tensor3_type(index_type n0, index_type n1, index_type n2)
        : n0_(n0), n1_(n1), n2_(n2), d_(static_cast<std::size_t>(n0) * n1 * n2, T{}) {}
    T& operator()(index_type i, index_type j, index_type k) {
        return d_[(static_cast<std::size_t>(i) * n1_ + j) * n2_ + k];
    }
    const T& operator()(index_type i, index_type j, index_type k) const {
        return d_[(static_cast<std::size_t>(i) * n1_ + j) * n2_ + k];
    }
    index_type length(int dim) const { return dim == 0 ? n0_ : (dim == 1 ? n1_ : n2_); }
    std::size_t numElements() const { return d_.size(); }
    T* data() { return d_.data(); }
    const T* data() const { return d_.data(); }
private:
    index_type n0_ = 0, n1_ = 0, n2_ = 0;
    detail::storage<T> d_;
};
using real_tensor3_type = tensor3_type<real_type>;

class CubatureContext2D {
public:
    struct Tables {
        index_type NCubature = 0, NumCubaturePoints = 0;
        real_vector_type r, s, w;                       // (Ncub)
        real_matrix_type V, Dr, Ds;                     // (Ncub, Np)
        real_matrix_type rx, sx, ry, sy, J, x, y, W;    // (Ncub, K)
        real_tensor3_type MM, MMChol;                   // (Np, Np, K)
    };
    CubatureContext2D() = default;
    explicit CubatureContext2D(Tables&& t) : t_{std::make_shared<Tables>(std::move(t))} {}

    index_type NCubature() const { return t_->NCubature; }
    index_type NumCubaturePoints() const { return t_->NumCubaturePoints; }
    const real_vector_type& r() const { return t_->r; }
    const real_vector_type& s() const { return t_->s; }
    const real_vector_type& w() const { return t_->w; }
    const real_matrix_type& V() const { return t_->V; }
    const real_matrix_type& rx() const { return t_->rx; }
    const real_matrix_type& sx() const { return t_->sx; }
    const real_matrix_type& ry() const { return t_->ry; }
    const real_matrix_type& sy() const { return t_->sy; }
    const real_matrix_type& Jac() const { return t_->J; }
    const real_matrix_type& Dr() const { return t_->Dr; }
    const real_matrix_type& Ds() const { return t_->Ds; }
    const real_tensor3_type& MM() const { return t_->MM; }
    const real_tensor3_type& MMChol() const { return t_->MMChol; }
    const real_matrix_type& x() const { return t_->x; }
    const real_matrix_type& y() const { return t_->y; }
    const real_matrix_type& W() const { return t_->W; }

private:
    std::shared_ptr<const Tables> t_;
};

} // namespace blitzdg
