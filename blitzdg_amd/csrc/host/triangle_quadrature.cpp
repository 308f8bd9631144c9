// Face-quadrature and volume-cubature meshes of the triangle provisioner (curved / over-integrated
// right-hand sides): TriangleNodesProvisioner::buildGaussFaceNodes and ::buildCubatureVolumeMesh,
// plus TriangleCubatureRules (the reference's tabulated rules; a computed rule beyond its table).
//
// Restates the reference's src/TriangleNodesProvisioner.cpp:207-381 (Gauss face nodes) and :81-205
// (cubature volume mesh) with plain loops, element-parallel (parallel_for.hpp). The Gauss-node maps mapM / mapP
// and the BC lists must come out identical to the reference's construction (face-major BC order,
// neighbour face traversed backwards); real tables agree to round-off.
#include "blitzdg/TriangleCubatureRules.hpp"
#include "blitzdg/TriangleNodesProvisioner.hpp"
#include "parallel_for.hpp"
#include <atomic>
#include <cmath>
#include <stdexcept>
#include <vector>

namespace blitzdg {

namespace {
#include "triangle_cubature_table.inc"
} // namespace

TriangleCubatureRules::TriangleCubatureRules(index_type NCubature) : NCubature_{NCubature} {
    if (NCubature < 1) throw std::runtime_error("TriangleCubatureRules: degree must be >= 1");
    if (NCubature > NumPreComputed) {
        *this = conical(NCubature);
        return;
    }
    const index_type n = kCubatureCount[NCubature - 1];
    const double* p = kCubaturePoints + 3 * static_cast<std::size_t>(kCubatureFirst[NCubature - 1]);
    r_.resize(n); s_.resize(n); w_.resize(n);
    for (index_type c = 0; c < n; ++c) {
        r_(c) = p[3 * c];
        s_(c) = p[3 * c + 1];
        w_(c) = p[3 * c + 2];
    }
}

TriangleCubatureRules TriangleCubatureRules::conical(index_type NCubature) {
    if (NCubature < 1) throw std::runtime_error("TriangleCubatureRules: degree must be >= 1");
    TriangleCubatureRules rule;
    rule.NCubature_ = NCubature;
    const index_type n = (NCubature + 2) / 2; // 2n - 1 >= NCubature
    JacobiBuilders jac;
    real_vector_type a(n), wa(n), b(n), wb(n);
    jac.computeJacobiQuadWeights(0.0, 0.0, n - 1, a, wa); // Legendre
    jac.computeJacobiQuadWeights(1.0, 0.0, n - 1, b, wb); // weight (1 - b): the collapse Jacobian
    rule.r_.resize(n * n); rule.s_.resize(n * n); rule.w_.resize(n * n);
    index_type c = 0;
    for (index_type j = 0; j < n; ++j)       // b outer: points come out row by row in s
        for (index_type i = 0; i < n; ++i, ++c) {
            rule.r_(c) = 0.5 * (1.0 + a(i)) * (1.0 - b(j)) - 1.0;
            rule.s_(c) = b(j);
            rule.w_(c) = 0.5 * wa(i) * wb(j);      // dr ds = (1 - b)/2 da db
        }
    return rule;
}

namespace {
real_matrix_type product(const real_matrix_type& A, const real_matrix_type& B) {
    real_matrix_type C(A.rows(), B.cols());
    for (index_type i = 0; i < A.rows(); ++i)
        for (index_type j = 0; j < B.cols(); ++j) {
            real_type s = 0;
            for (index_type k = 0; k < A.cols(); ++k) s += A(i, k) * B(k, j);
            C(i, j) = s;
        }
    return C;
}
} // namespace

// ------------------------------------------------------------------ Gauss face nodes (:207-381)

GaussFaceContext2D TriangleNodesProvisioner::buildGaussFaceNodes(index_type NGauss) {
    if (NGauss < 1) throw std::runtime_error("buildGaussFaceNodes: NGauss must be >= 1");
    const index_type Np = NumLocalPoints, K = NumElements, NF = NumFaces, Ng = NGauss, Nfp = NF * Ng;
    if (static_cast<long long>(Nfp) * K > 2147483647LL)
        throw std::runtime_error("buildGaussFaceNodes: 3*NGauss*K exceeds 32-bit node numbering");

    real_vector_type z(Ng), w(Ng);
    Jacobi.computeJacobiQuadWeights(0, 0, Ng - 1, z, w);

    // face 1: s = -1, r = z;  face 2: r = -z, s = z;  face 3: r = -1, s = -z   (:218-219)
    GaussFaceContext2D::Tables t;
    t.NGauss = Ng;
    t.Interp.resize(Nfp, Np);
    std::vector<real_matrix_type> dVdr(NF), dVds(NF);
    for (index_type f = 0; f < NF; ++f) {
        real_vector_type fr(Ng), fs(Ng);
        for (index_type i = 0; i < Ng; ++i) {
            fr(i) = f == 0 ? z(i) : (f == 1 ? -z(i) : -1.0);
            fs(i) = f == 0 ? -1.0 : (f == 1 ? z(i) : -z(i));
        }
        real_matrix_type Vf(Ng, Np);
        computeVandermondeMatrix(NOrder, fr, fs, Vf);
        const real_matrix_type If = product(Vf, Vinv);
        for (index_type i = 0; i < Ng; ++i)
            for (index_type j = 0; j < Np; ++j) t.Interp(f * Ng + i, j) = If(i, j);
        dVdr[f] = product(If, Dr);
        dVds[f] = product(If, Ds);
    }

    for (real_matrix_type* m : {&t.nx, &t.ny, &t.sJ, &t.Jac, &t.rx, &t.ry, &t.sx, &t.sy, &t.x, &t.y, &t.W})
        m->resizeUninitialized(Nfp, K);
    t.mapM.resizeUninitialized(Nfp * K);
    t.mapP.resizeUninitialized(Nfp * K);

    const index_vector_type& E2E = Mesh2D->get_EToE();
    const index_vector_type& E2F = Mesh2D->get_EToF();
    const index_vector_type& bcVec = Mesh2D->get_BCType();

    detail::parallelFor(K, [&](index_type k) {
        for (index_type f = 0; f < NF; ++f) {
            for (index_type ig = 0; ig < Ng; ++ig) {
                real_type xr = 0, yr = 0, xs = 0, ys = 0, gx = 0, gy = 0;
                for (index_type m = 0; m < Np; ++m) {
                    const real_type xm = xGrid(m, k), ym = yGrid(m, k);
                    xr += dVdr[f](ig, m) * xm; yr += dVdr[f](ig, m) * ym;
                    xs += dVds[f](ig, m) * xm; ys += dVds[f](ig, m) * ym;
                    gx += t.Interp(f * Ng + ig, m) * xm; gy += t.Interp(f * Ng + ig, m) * ym;
                }
                const real_type jac = xr * ys - xs * yr;
                const real_type grx = ys / jac, gry = -xs / jac, gsx = -yr / jac, gsy = xr / jac;
                real_type gnx, gny;
                if (f == 0) { gnx = -gsx; gny = -gsy; }
                else if (f == 1) { gnx = grx + gsx; gny = gry + gsy; }
                else { gnx = -grx; gny = -gry; }
                real_type sj = std::sqrt(gnx * gnx + gny * gny);
                gnx = gnx / sj; gny = gny / sj;
                sj = sj * jac;
                const index_type row = f * Ng + ig;
                t.nx(row, k) = gnx; t.ny(row, k) = gny; t.sJ(row, k) = sj; t.Jac(row, k) = jac;
                t.rx(row, k) = grx; t.ry(row, k) = gry; t.sx(row, k) = gsx; t.sy(row, k) = gsy;
                t.x(row, k) = gx; t.y(row, k) = gy;
                t.W(row, k) = w(ig) * sj;
                // flat ids: node (row, k) -> row + Nfp*k; the neighbour's face is traversed backwards (:318-326)
                const index_type k2 = E2E(NF * k + f), f2 = E2F(NF * k + f);
                t.mapM(row + k * Nfp) = row + k * Nfp;
                t.mapP(row + k * Nfp) = (k != k2) ? (Ng * (f2 + 1) - ig - 1) + k2 * Nfp : row + k * Nfp;
            }
        }
    });
    // boundary lists in the reference's order: faces outermost, then elements, then Gauss points (:253, :328-335)
    for (index_type tag : {3, 6, 7, 1, 2, 4, 5, 8}) t.bcMap[tag];   // Wall, Dirichlet, Neuman, In, Out, Cyl, Far, Slip
    for (index_type f = 0; f < NF; ++f)
        for (index_type k = 0; k < K; ++k)
            if (E2E(NF * k + f) == k) {
                std::vector<index_type>& list = t.bcMap[bcVec(NF * k + f)];
                for (index_type ig = 0; ig < Ng; ++ig) list.push_back(f * Ng + ig + k * Nfp);
            }
    return GaussFaceContext2D(std::move(t));
}

// ------------------------------------------------------------------ cubature volume mesh (:81-205)

CubatureContext2D TriangleNodesProvisioner::buildCubatureVolumeMesh(index_type NCubature) {
    const TriangleCubatureRules cubature(NCubature);
    const index_type Ncub = cubature.NumCubaturePoints(), Np = NumLocalPoints, K = NumElements;

    CubatureContext2D::Tables t;
    t.NCubature = NCubature;
    t.NumCubaturePoints = Ncub;
    t.r = cubature.rCoord();
    t.s = cubature.sCoord();
    t.w = cubature.weights();
    t.V.resize(Ncub, Np);
    computeInterpMatrix(t.r, t.s, t.V);
    {   // Drcub = V2Dr V^{-1} (the weak operators the reference also asks for are never used, :94-99)
        real_matrix_type V2Dr(Ncub, Np), V2Ds(Ncub, Np);
        computeGradVandermondeMatrix(NOrder, t.r, t.s, V2Dr, V2Ds);
        t.Dr = product(V2Dr, Vinv);
        t.Ds = product(V2Ds, Vinv);
    }
    for (real_matrix_type* m : {&t.rx, &t.sx, &t.ry, &t.sy, &t.J, &t.x, &t.y, &t.W}) m->resizeUninitialized(Ncub, K);
    t.MM = real_tensor3_type(Np, Np, K);
    t.MMChol = real_tensor3_type(Np, Np, K);

    std::atomic<bool> notPositive{false};
    detail::parallelChunks(K, [&](index_type kBegin, index_type kEnd) {
        std::vector<real_type> xe(Np), ye(Np), jw(Ncub);
        real_matrix_type MMk(Np, Np), R(Np, Np);
        DenseCholeskyFactorizer chol;
        for (index_type k = kBegin; k < kEnd; ++k) {
            for (index_type m = 0; m < Np; ++m) { xe[m] = xGrid(m, k); ye[m] = yGrid(m, k); }
            for (index_type i = 0; i < Ncub; ++i) {
                real_type xr = 0, xs = 0, yr = 0, ys = 0, xc = 0, yc = 0;
                for (index_type m = 0; m < Np; ++m) {
                    xr += t.Dr(i, m) * xe[m]; xs += t.Ds(i, m) * xe[m];
                    yr += t.Dr(i, m) * ye[m]; ys += t.Ds(i, m) * ye[m];
                    xc += t.V(i, m) * xe[m];  yc += t.V(i, m) * ye[m];
                }
                const real_type jac = -xs * yr + xr * ys;
                t.J(i, k) = jac;
                t.rx(i, k) = ys / jac; t.sx(i, k) = -yr / jac; t.ry(i, k) = -xs / jac; t.sy(i, k) = xr / jac;
                t.x(i, k) = xc; t.y(i, k) = yc;
                t.W(i, k) = t.w(i) * jac;
                jw[i] = jac * t.w(i);
            }
            // the nodal metric terms are repaired from the current coordinates as well (:129-152)
            for (index_type i = 0; i < Np; ++i) {
                real_type xr = 0, xs = 0, yr = 0, ys = 0;
                for (index_type m = 0; m < Np; ++m) {
                    xr += Dr(i, m) * xe[m]; yr += Dr(i, m) * ye[m];
                    xs += Ds(i, m) * xe[m]; ys += Ds(i, m) * ye[m];
                }
                const real_type jac = xr * ys - xs * yr;
                J(i, k) = jac;
                rx(i, k) = ys / jac; ry(i, k) = -xs / jac; sx(i, k) = -yr / jac; sy(i, k) = xr / jac;
            }
            // cubature mass matrix V^T diag(J w) V and its upper Cholesky factor (:175-187)
            for (index_type i = 0; i < Np; ++i)
                for (index_type j = 0; j < Np; ++j) {
                    real_type s = 0;
                    for (index_type c = 0; c < Ncub; ++c) s += t.V(c, i) * (jw[c] * t.V(c, j));
                    MMk(i, j) = s;
                }
            try {
                chol.computeCholesky(MMk, R);
            } catch (const std::runtime_error&) {
                notPositive = true;
                continue;
            }
            for (index_type i = 0; i < Np; ++i)
                for (index_type j = 0; j < Np; ++j) {
                    t.MM(i, j, k) = MMk(i, j);
                    t.MMChol(i, j, k) = R(i, j);
                }
        }
    }, 16);
    if (notPositive)
        throw std::runtime_error("buildCubatureVolumeMesh: an element's cubature mass matrix is not positive definite "
                                 "(inverted element?)");
    return CubatureContext2D(std::move(t));
}

} // namespace blitzdg
