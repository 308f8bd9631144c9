// DGContext2D: non-owning bundle of the 2-D DG tables the RHS evaluator reads.
// Same accessor names as the reference's include/DGContext2D.hpp:82-196. The
// view borrows from the TriangleNodesProvisioner that created it and must not
// outlive it (as in the reference, which hands out unique_ptr::get() pointers,
// src/TriangleNodesProvisioner.cpp:1364-1396).
#pragma once
#include "Types.hpp"
#include <vector>

namespace blitzdg {

class DGContext2D {
public:
    DGContext2D() = default;
    DGContext2D(index_type order, index_type numLocalPoints, index_type numFacePoints, index_type numElems,
                index_type numFaces, const real_matrix_type* filter, const real_vector_type* rgrid,
                const real_vector_type* sgrid, const real_matrix_type* xgrid, const real_matrix_type* ygrid,
                const real_matrix_type* fscale, const index_matrix_type* fmask,
                const std::vector<index_type>* gather, const std::vector<index_type>* scatter,
                const real_matrix_type* vandermonde2d, const real_matrix_type* vandermonde2dinv,
                const real_matrix_type* jacobian, const real_matrix_type* rx, const real_matrix_type* ry,
                const real_matrix_type* sx, const real_matrix_type* sy, const real_matrix_type* nx,
                const real_matrix_type* ny, const real_matrix_type* Dr, const real_matrix_type* Ds,
                const real_matrix_type* lift, const index_vector_type* vmapM, const index_vector_type* vmapP,
                const index_hashmap* bcmap)
        : N_{order}, Np_{numLocalPoints}, Nfp_{numFacePoints}, K_{numElems}, NumFaces_{numFaces},
          Filt_{filter}, r_{rgrid}, s_{sgrid}, xGrid_{xgrid}, yGrid_{ygrid}, Fscale_{fscale}, Fmask_{fmask},
          Gather_{gather}, Scatter_{scatter}, V_{vandermonde2d}, Vinv_{vandermonde2dinv}, J_{jacobian},
          rx_{rx}, ry_{ry}, sx_{sx}, sy_{sy}, nx_{nx}, ny_{ny}, Dr_{Dr}, Ds_{Ds}, Lift_{lift},
          vmapM_{vmapM}, vmapP_{vmapP}, bcHash_{bcmap} {}

    index_type order() const { return N_; }
    index_type numLocalPoints() const { return Np_; }
    index_type numFacePoints() const { return Nfp_; }
    index_type numElements() const { return K_; }
    index_type numFaces() const { return NumFaces_; }
    const real_matrix_type& filter() const { return *Filt_; }
    const real_vector_type& r() const { return *r_; }
    const real_vector_type& s() const { return *s_; }
    const real_matrix_type& x() const { return *xGrid_; }
    const real_matrix_type& y() const { return *yGrid_; }
    const real_matrix_type& fscale() const { return *Fscale_; }
    const index_matrix_type& fmask() const { return *Fmask_; }
    const std::vector<index_type>& gather() const { return *Gather_; }
    const std::vector<index_type>& scatter() const { return *Scatter_; }
    const real_matrix_type& V() const { return *V_; }
    const real_matrix_type& Vinv() const { return *Vinv_; }
    const real_matrix_type& jacobian() const { return *J_; }
    const real_matrix_type& rx() const { return *rx_; }
    const real_matrix_type& sx() const { return *sx_; }
    const real_matrix_type& ry() const { return *ry_; }
    const real_matrix_type& sy() const { return *sy_; }
    const real_matrix_type& nx() const { return *nx_; }
    const real_matrix_type& ny() const { return *ny_; }
    const real_matrix_type& Dr() const { return *Dr_; }
    const real_matrix_type& Ds() const { return *Ds_; }
    const real_matrix_type& lift() const { return *Lift_; }
    const index_vector_type& vmapM() const { return *vmapM_; }
    const index_vector_type& vmapP() const { return *vmapP_; }
    const index_hashmap& bcmap() const { return *bcHash_; }

    /// Physical-space derivative operators of one (possibly curved) element from
    /// its nodal coordinates (reference include/DGContext2D.hpp:222-257).
    void computeDifferentiationMatrices(const real_vector_type& x, const real_vector_type& y,
                                        const real_matrix_type& V, real_matrix_type& Dx,
                                        real_matrix_type& Dy) const {
        const real_matrix_type& D_r = *Dr_;
        const real_matrix_type& D_s = *Ds_;
        const index_type Nout = V.rows();
        for (index_type i = 0; i < Nout && i < Np_; ++i) {
            real_type xr = 0, xs = 0, yr = 0, ys = 0;
            for (index_type k = 0; k < Np_; ++k) {
                xr += D_r(i, k) * x(k); xs += D_s(i, k) * x(k);
                yr += D_r(i, k) * y(k); ys += D_s(i, k) * y(k);
            }
            const real_type J = -xs * yr + xr * ys;
            const real_type rxi = ys / J, sxi = -yr / J, ryi = -xs / J, syi = xr / J;
            for (index_type j = 0; j < Np_; ++j) {
                Dx(i, j) = rxi * D_r(i, j) + sxi * D_s(i, j);
                Dy(i, j) = ryi * D_r(i, j) + syi * D_s(i, j);
            }
        }
    }

private:
    index_type N_ = 0, Np_ = 0, Nfp_ = 0, K_ = 0, NumFaces_ = 0;
    const real_matrix_type* Filt_ = nullptr;
    const real_vector_type* r_ = nullptr;
    const real_vector_type* s_ = nullptr;
    const real_matrix_type* xGrid_ = nullptr;
    const real_matrix_type* yGrid_ = nullptr;
    const real_matrix_type* Fscale_ = nullptr;
    const index_matrix_type* Fmask_ = nullptr;
    const std::vector<index_type>* Gather_ = nullptr;
    const std::vector<index_type>* Scatter_ = nullptr;
    const real_matrix_type* V_ = nullptr;
    const real_matrix_type* Vinv_ = nullptr;
    const real_matrix_type* J_ = nullptr;
    const real_matrix_type* rx_ = nullptr;
    const real_matrix_type* ry_ = nullptr;
    const real_matrix_type* sx_ = nullptr;
    const real_matrix_type* sy_ = nullptr;
    const real_matrix_type* nx_ = nullptr;
    const real_matrix_type* ny_ = nullptr;
    const real_matrix_type* Dr_ = nullptr;
    const real_matrix_type* Ds_ = nullptr;
    const real_matrix_type* Lift_ = nullptr;
    const index_vector_type* vmapM_ = nullptr;
    const index_vector_type* vmapP_ = nullptr;
    const index_hashmap* bcHash_ = nullptr;
};

} // namespace blitzdg
