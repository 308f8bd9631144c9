// TriangleNodesProvisioner implementation (setup path; CPU, element loops on std::thread workers).
//
// Restates the table construction of the reference's
// src/TriangleNodesProvisioner.cpp with plain loops:
//   simplex basis + gradients :383-393, :642-676      nodes (warp & blend) :549-640
//   V / gradV :418-454     Dr, Ds (and weak Drw, Dsw) :456-513     filter :515-547
//   Fmask :678-728    physical grid, metric terms, normals, Fscale :730-893
//   vmapM / vmapP / mapP / vmapB / mapB :895-1005    BC hash :1022-1057    Lift :1060-1138
// The index maps must come out bit-identical to the reference's (its tests pin
// them on input/coarse_box.msh); real-valued tables agree to round-off.
#include "blitzdg/TriangleNodesProvisioner.hpp"
#include "parallel_for.hpp"
#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>
#include <stdexcept>

namespace blitzdg {

const index_type TriangleNodesProvisioner::NumFaces = 3;
const real_type TriangleNodesProvisioner::NodeTol = 1.e-5;

namespace {
const real_type pi = 3.141592653589793238462643383279502884;

// C = A * B for small dense matrices.
real_matrix_type matmul(const real_matrix_type& A, const real_matrix_type& B) {
    real_matrix_type C(A.rows(), B.cols());
    for (index_type i = 0; i < A.rows(); ++i)
        for (index_type j = 0; j < B.cols(); ++j) {
            real_type s = 0;
            for (index_type k = 0; k < A.cols(); ++k) s += A(i, k) * B(k, j);
            C(i, j) = s;
        }
    return C;
}

real_matrix_type transposed(const real_matrix_type& A) {
    real_matrix_type T(A.cols(), A.rows());
    for (index_type i = 0; i < A.rows(); ++i)
        for (index_type j = 0; j < A.cols(); ++j) T(j, i) = A(i, j);
    return T;
}
} // namespace

TriangleNodesProvisioner::TriangleNodesProvisioner(index_type N, const MeshManager& mesh)
    : NumElements{mesh.get_NumElements()}, NOrder{N}, NumLocalPoints{(N + 2) * (N + 1) / 2},
      NumFacePoints{N + 1}, Mesh2D{&mesh} {
    if (N < 1) throw std::runtime_error("TriangleNodesProvisioner: polynomial order must be >= 1");
    if (mesh.get_NumFaces() != NumFaces)
        throw std::runtime_error("TriangleNodesProvisioner: mesh is not a triangle mesh");
    const long long total = static_cast<long long>(NumLocalPoints) * NumElements;
    if (total > std::numeric_limits<index_type>::max())
        throw std::runtime_error("TriangleNodesProvisioner: Np*K exceeds 32-bit node numbering");
    const index_type Np = NumLocalPoints, K = NumElements, Nfaces = NumFaces * NumFacePoints;
    rGrid.resize(Np); sGrid.resize(Np);
    V.resize(Np, Np); Vinv.resize(Np, Np); Dr.resize(Np, Np); Ds.resize(Np, Np);
    Drw.resize(Np, Np); Dsw.resize(Np, Np); Filter.resize(Np, Np);
    Lift.resize(Np, Nfaces);
    Fmask.resize(NumFacePoints, NumFaces);
    // The (rows, K) tables are written entry by entry by buildPhysicalGrid / buildMaps: leave their
    // first touch to those element-parallel loops instead of zero-filling gigabytes serially here.
    for (real_matrix_type* m : {&xGrid, &yGrid, &J, &rx, &sx, &ry, &sy}) m->resizeUninitialized(Np, K);
    for (real_matrix_type* m : {&nx, &ny, &Fscale, &Fx, &Fy}) m->resizeUninitialized(Nfaces, K);
    for (index_vector_type* v : {&vmapM, &vmapP, &mapP}) v->resizeUninitialized(Nfaces * K);

    buildNodes();
    buildLift();
    buildPhysicalGrid();
    buildMaps();
}

// ---------------------------------------------------------------- basis functions

void TriangleNodesProvisioner::evaluateSimplexPolynomial(const real_vector_type& a, const real_vector_type& b,
                                                         index_type i, index_type j, real_vector_type& p) const {
    const index_type n = a.size();
    real_vector_type h1(n), h2(n);
    Jacobi.computeJacobiPolynomial(a, 0.0, 0.0, i, h1);
    Jacobi.computeJacobiPolynomial(b, 2.0 * i + 1.0, 0.0, j, h2);
    if (p.size() != n) p.resize(n);
    for (index_type k = 0; k < n; ++k) p(k) = std::sqrt(2.0) * h1(k) * h2(k) * std::pow(1. - b(k), i);
}

void TriangleNodesProvisioner::evaluateGradSimplex(const real_vector_type& a, const real_vector_type& b,
                                                   index_type id, index_type jd, real_vector_type& dpdr,
                                                   real_vector_type& dpds) const {
    const index_type n = a.size();
    real_vector_type fa(n), gb(n), dfa(n), dgb(n);
    Jacobi.computeJacobiPolynomial(a, 0., 0., id, fa);
    Jacobi.computeJacobiPolynomial(b, 2. * id + 1., 0., jd, gb);
    Jacobi.computeGradJacobi(a, 0., 0., id, dfa);
    Jacobi.computeGradJacobi(b, 2. * id + 1., 0., jd, dgb);
    if (dpdr.size() != n) dpdr.resize(n);
    if (dpds.size() != n) dpds.resize(n);
    const real_type norm = std::pow(2., id + 0.5);
    for (index_type k = 0; k < n; ++k) {
        const real_type hb = 0.5 * (1. - b(k));
        // d/dr = (2/(1-b)) d/da
        real_type dr = dfa(k) * gb(k);
        if (id > 1) dr *= std::pow(hb, id - 1);
        // d/ds = ((1+a)/2)/((1-b)/2) d/da + d/db
        real_type ds = dfa(k) * (gb(k) * (0.5 * (1 + a(k))));
        if (id > 1) ds *= std::pow(hb, id - 1);
        real_type tmp = dgb(k) * std::pow(hb, id);
        if (id > 0) tmp -= 0.5 * id * gb(k) * std::pow(hb, id - 1);
        ds += fa(k) * tmp;
        dpdr(k) = norm * dr;
        dpds(k) = norm * ds;
    }
}

void TriangleNodesProvisioner::rsToab(const real_vector_type& r, const real_vector_type& s, real_vector_type& a,
                                      real_vector_type& b) const {
    const index_type n = r.size();
    if (a.size() != n) a.resize(n);
    if (b.size() != n) b.resize(n);
    for (index_type i = 0; i < n; ++i) {
        a(i) = (s(i) != 1.0) ? 2.0 * (1.0 + r(i)) / (1.0 - s(i)) - 1.0 : -1.0;
        b(i) = s(i);
    }
}

void TriangleNodesProvisioner::xyTors(const real_vector_type& x, const real_vector_type& y, real_vector_type& r,
                                      real_vector_type& s) const {
    const index_type n = x.size();
    if (r.size() != n) r.resize(n);
    if (s.size() != n) s.resize(n);
    const real_type rt3 = std::sqrt(3.0);
    for (index_type i = 0; i < n; ++i) {
        const real_type L1 = (rt3 * y(i) + 1.0) / 3.0;
        const real_type L2 = (-3.0 * x(i) - rt3 * y(i) + 2.0) / 6.0;
        const real_type L3 = (3.0 * x(i) - rt3 * y(i) + 2.0) / 6.0;
        r(i) = -L2 + L3 - L1;
        s(i) = -L2 - L3 + L1;
    }
}

void TriangleNodesProvisioner::computeVandermondeMatrix(index_type N, const real_vector_type& r,
                                                        const real_vector_type& s, real_matrix_type& Vout) const {
    const index_type nr = r.size();
    real_vector_type a(nr), b(nr), p(nr);
    rsToab(r, s, a, b);
    index_type col = 0;
    for (index_type i = 0; i <= N; ++i)
        for (index_type j = 0; j <= N - i; ++j) {
            evaluateSimplexPolynomial(a, b, i, j, p);
            for (index_type k = 0; k < nr; ++k) Vout(k, col) = p(k);
            ++col;
        }
}

void TriangleNodesProvisioner::computeGradVandermondeMatrix(index_type N, const real_vector_type& r,
                                                            const real_vector_type& s, real_matrix_type& V2Dr,
                                                            real_matrix_type& V2Ds) const {
    const index_type nr = r.size();
    real_vector_type a(nr), b(nr), dr(nr), ds(nr);
    rsToab(r, s, a, b);
    index_type col = 0;
    for (index_type i = 0; i <= N; ++i)
        for (index_type j = 0; j <= N - i; ++j) {
            evaluateGradSimplex(a, b, i, j, dr, ds);
            for (index_type k = 0; k < nr; ++k) {
                V2Dr(k, col) = dr(k);
                V2Ds(k, col) = ds(k);
            }
            ++col;
        }
}

void TriangleNodesProvisioner::computeDifferentiationMatrices(const real_matrix_type& V2Dr,
                                                              const real_matrix_type& V2Ds,
                                                              const real_matrix_type& Vmat,
                                                              const real_matrix_type& Vc, real_matrix_type& Dr_,
                                                              real_matrix_type& Ds_, real_matrix_type& Drw_,
                                                              real_matrix_type& Dsw_) const {
    // Dr = V2Dr V^{-1}  <=>  V^T Dr^T = V2Dr^T  (likewise Ds).
    const real_matrix_type Vt = transposed(Vmat), V2Drt = transposed(V2Dr), V2Dst = transposed(V2Ds);
    real_matrix_type Drt, Dst;
    LinSolver.solve(Vt, V2Drt, Drt);
    LinSolver.solve(Vt, V2Dst, Dst);
    Dr_ = transposed(Drt);
    Ds_ = transposed(Dst);

    // Weak operators: Drw = (Vc V2Dr^T) (Vc Vc^T)^{-1}.
    const real_matrix_type VVt = matmul(Vc, transposed(Vc));
    const real_matrix_type VVrt = matmul(Vc, V2Drt), VVst = matmul(Vc, V2Dst);
    real_matrix_type Drwt, Dswt;
    LinSolver.solve(transposed(VVt), transposed(VVrt), Drwt);
    LinSolver.solve(transposed(VVt), transposed(VVst), Dswt);
    Drw_ = transposed(Drwt);
    Dsw_ = transposed(Dswt);
}

void TriangleNodesProvisioner::buildFilter(real_type Nc, index_type s) {
    const real_type alpha = -std::log(std::numeric_limits<real_type>::epsilon());
    const index_type Np = NumLocalPoints;
    real_vector_type sigma(Np);
    index_type count = 0;
    for (index_type i = 0; i <= NOrder; ++i)
        for (index_type j = 0; j <= NOrder - i; ++j) {
            if ((i + j) >= Nc) {
                const real_type k = (static_cast<real_type>(i + j) - Nc) / (static_cast<real_type>(NOrder) - Nc);
                sigma(count) = std::exp(-alpha * std::pow(k, s));
            } else {
                sigma(count) = 1.0;
            }
            ++count;
        }
    // Filter = V diag(sigma) V^{-1}
    real_matrix_type tmp(Np, Np);
    for (index_type i = 0; i < Np; ++i)
        for (index_type j = 0; j < Np; ++j) tmp(i, j) = sigma(i) * Vinv(i, j);
    Filter = matmul(V, tmp);
}

// ---------------------------------------------------------------- nodes

void TriangleNodesProvisioner::computeEquilateralNodes(real_vector_type& x, real_vector_type& y) const {
    static const real_type alphaOptimal[15] = {0.0000, 0.0000, 1.4152, 0.1001, 0.2751, 0.9800, 1.0999, 1.2832,
                                               1.3648, 1.4773, 1.4959, 1.5743, 1.5770, 1.6223, 1.6258};
    const real_type alpha = (NOrder < 16) ? alphaOptimal[NOrder - 1] : 2.0 / 3.0;
    const index_type Np = (NOrder + 1) * (NOrder + 2) / 2;
    if (x.size() != Np) x.resize(Np);
    if (y.size() != Np) y.resize(Np);

    // Equidistributed barycentric lattice on the equilateral triangle.
    real_vector_type L1(Np), L2(Np), L3(Np);
    index_type count = 0;
    for (index_type n = 1; n <= NOrder + 1; ++n)
        for (index_type m = 1; m <= NOrder + 2 - n; ++m) {
            L1(count) = (n - 1.0) / NOrder;
            L3(count) = (m - 1.0) / NOrder;
            ++count;
        }
    real_vector_type t1(Np), t2(Np), t3(Np), w1(Np), w2(Np), w3(Np);
    for (index_type i = 0; i < Np; ++i) {
        L2(i) = 1.0 - L1(i) - L3(i);
        x(i) = -L2(i) + L3(i);
        y(i) = (-L2(i) - L3(i) + 2 * L1(i)) / std::sqrt(3.0);
        t1(i) = L3(i) - L2(i);
        t2(i) = L1(i) - L3(i);
        t3(i) = L2(i) - L1(i);
    }
    computeWarpFactor(t1, w1);
    computeWarpFactor(t2, w2);
    computeWarpFactor(t3, w3);
    const real_type a2 = alpha * alpha;
    for (index_type i = 0; i < Np; ++i) {
        // blend (4 Li Lj) * warp * (1 + (alpha Lk)^2), one term per edge
        const real_type warp1 = 4 * L2(i) * L3(i) * w1(i) * (1 + a2 * L1(i) * L1(i));
        const real_type warp2 = 4 * L1(i) * L3(i) * w2(i) * (1 + a2 * L2(i) * L2(i));
        const real_type warp3 = 4 * L1(i) * L2(i) * w3(i) * (1 + a2 * L3(i) * L3(i));
        x(i) += 1 * warp1 + std::cos(2 * pi / 3) * warp2 + std::cos(4 * pi / 3) * warp3;
        y(i) += 0 * warp1 + std::sin(2 * pi / 3) * warp2 + std::sin(4 * pi / 3) * warp3;
    }
}

void TriangleNodesProvisioner::computeWarpFactor(const real_vector_type& r, real_vector_type& warpFactor) const {
    const index_type Np1 = NOrder + 1, nr = r.size();
    real_vector_type req(Np1), rLGL(Np1);
    for (index_type i = 0; i < Np1; ++i) req(i) = -1.0 + 2 * i / (Np1 - 1.0);
    Jacobi.computeGaussLobottoPoints(0.0, 0.0, NOrder, rLGL);

    real_matrix_type Veq(Np1, Np1), Veqinv(Np1, Np1);
    Vandermonde.computeVandermondeMatrix(req, Veq, Veqinv);

    // Lagrange interpolants through the equispaced points, evaluated at r:
    // Veq^T L = P, P(i,:) = Legendre_i(r).
    real_matrix_type P(Np1, nr), L;
    real_vector_type p(nr);
    for (index_type i = 0; i < Np1; ++i) {
        Jacobi.computeJacobiPolynomial(r, 0.0, 0.0, i, p);
        for (index_type k = 0; k < nr; ++k) P(i, k) = p(k);
    }
    LinSolver.solve(transposed(Veq), P, L);

    if (warpFactor.size() != nr) warpFactor.resize(nr);
    for (index_type k = 0; k < nr; ++k) {
        real_type w = 0;
        for (index_type j = 0; j < Np1; ++j) w += L(j, k) * (rLGL(j) - req(j));
        // Scale by 1/(1-r^2) away from the end points.
        const real_type zf = (std::fabs(r(k)) < 1.0 - 1e-10) ? 1.0 : 0.0;
        const real_type sf = 1.0 - (zf * r(k)) * (zf * r(k));
        warpFactor(k) = w / sf + w * (zf - 1.0);
    }
}

void TriangleNodesProvisioner::buildNodes() {
    real_vector_type x(NumLocalPoints), y(NumLocalPoints);
    computeEquilateralNodes(x, y);
    xyTors(x, y, rGrid, sGrid);

    // Face masks: ascending node indices on s=-1, r+s=0, r=-1.
    for (index_type f = 0; f < NumFaces; ++f) {
        index_type count = 0;
        for (index_type n = 0; n < NumFacePoints; ++n) Fmask(n, f) = 0;
        for (index_type i = 0; i < NumLocalPoints; ++i) {
            const real_type t = (f == 0) ? sGrid(i) + 1 : (f == 1) ? rGrid(i) + sGrid(i) : rGrid(i) + 1;
            if (std::fabs(t) < NodeTol) {
                if (count >= NumFacePoints) throw std::runtime_error("buildNodes: too many nodes on a face");
                Fmask(count++, f) = i;
            }
        }
    }
}

void TriangleNodesProvisioner::buildLift() {
    const index_type Np = NumLocalPoints, Nfp = NumFacePoints;
    real_matrix_type E(Np, NumFaces * Nfp);
    real_vector_type faceCoord(Nfp);
    real_matrix_type V1D(Nfp, Nfp), V1Dinv(Nfp, Nfp), massEdge(Nfp, Nfp);
    for (index_type f = 0; f < NumFaces; ++f) {
        // Edge parametrised by r on faces 0,1 and by s on face 2.
        for (index_type i = 0; i < Nfp; ++i) faceCoord(i) = (f < 2) ? rGrid(Fmask(i, f)) : sGrid(Fmask(i, f));
        Vandermonde.computeVandermondeMatrix(faceCoord, V1D, V1Dinv);
        const real_matrix_type massEdgeInv = matmul(V1D, transposed(V1D));
        Inverter.computeInverse(massEdgeInv, massEdge);
        for (index_type i = 0; i < Nfp; ++i)
            for (index_type j = 0; j < Nfp; ++j) E(Fmask(i, f), f * Nfp + j) = massEdge(i, j);
    }
    computeVandermondeMatrix(NOrder, rGrid, sGrid, V);
    Inverter.computeInverse(V, Vinv);
    // Lift = M^{-1} E with M^{-1} = V V^T.
    Lift = matmul(matmul(V, transposed(V)), E);
}

// ---------------------------------------------------------------- physical grid

void TriangleNodesProvisioner::buildPhysicalGrid() {
    const index_vector_type& EToV = Mesh2D->get_Elements();
    const real_vector_type& Vert = Mesh2D->get_Vertices();
    NumElements = Mesh2D->get_NumElements();
    const index_type Np = NumLocalPoints, K = NumElements, Nfp = NumFacePoints;

    computeVandermondeMatrix(NOrder, rGrid, sGrid, V);
    Inverter.computeInverse(V, Vinv);
    real_matrix_type V2Dr(Np, Np), V2Ds(Np, Np);
    computeGradVandermondeMatrix(NOrder, rGrid, sGrid, V2Dr, V2Ds);
    computeDifferentiationMatrices(V2Dr, V2Ds, V, V, Dr, Ds, Drw, Dsw);

    detail::parallelChunks(K, [&](index_type kBegin, index_type kEnd) {
        std::vector<real_type> xr(Np), xs(Np), yr(Np), ys(Np), xe(Np), ye(Np);
        for (index_type k = kBegin; k < kEnd; ++k) {
            const index_type va = EToV(3 * k), vb = EToV(3 * k + 1), vc = EToV(3 * k + 2);
            const real_type xa = Vert(3 * va), xb = Vert(3 * vb), xc = Vert(3 * vc);
            const real_type ya = Vert(3 * va + 1), yb = Vert(3 * vb + 1), yc = Vert(3 * vc + 1);
            // Affine map of the reference triangle onto element k.
            for (index_type n = 0; n < Np; ++n) {
                const real_type r = rGrid(n), s = sGrid(n);
                xe[n] = 0.5 * (-(r + s) * xa + (1 + r) * xb + (1 + s) * xc);
                ye[n] = 0.5 * (-(r + s) * ya + (1 + r) * yb + (1 + s) * yc);
                xGrid(n, k) = xe[n];
                yGrid(n, k) = ye[n];
            }
            for (index_type i = 0; i < Np; ++i) {
                real_type a = 0, b = 0, c = 0, d = 0;
                for (index_type m = 0; m < Np; ++m) {
                    a += Dr(i, m) * xe[m]; b += Dr(i, m) * ye[m];
                    c += Ds(i, m) * xe[m]; d += Ds(i, m) * ye[m];
                }
                xr[i] = a; yr[i] = b; xs[i] = c; ys[i] = d;
                const real_type jac = a * d - c * b;
                J(i, k) = jac;
                rx(i, k) = d / jac;
                ry(i, k) = -c / jac;
                sx(i, k) = -b / jac;
                sy(i, k) = a / jac;
            }
            // Outward normals and surface Jacobian at the face nodes.
            for (index_type f = 0; f < NumFaces; ++f)
                for (index_type n = 0; n < Nfp; ++n) {
                    const index_type row = f * Nfp + n, v = Fmask(n, f);
                    Fx(row, k) = xe[v];
                    Fy(row, k) = ye[v];
                    real_type nxv, nyv;
                    if (f == 0)      { nxv = yr[v];          nyv = -xr[v]; }
                    else if (f == 1) { nxv = ys[v] - yr[v];  nyv = -xs[v] + xr[v]; }
                    else             { nxv = -ys[v];         nyv = xs[v]; }
                    const real_type norm = std::sqrt(nxv * nxv + nyv * nyv);
                    nx(row, k) = nxv / norm;
                    ny(row, k) = nyv / norm;
                    Fscale(row, k) = norm / J(v, k);
                }
        }
    });
}

void TriangleNodesProvisioner::setCoordinates(const real_type* x, const real_type* y) {
    std::copy(x, x + xGrid.numElements(), xGrid.data());
    std::copy(y, y + yGrid.numElements(), yGrid.data());
}

// ---------------------------------------------------------------- index maps

void TriangleNodesProvisioner::buildMaps() {
    const index_type Np = NumLocalPoints, K = NumElements, Nfp = NumFacePoints;
    const index_vector_type& E2E = Mesh2D->get_EToE();
    const index_vector_type& E2F = Mesh2D->get_EToF();
    const index_vector_type& E2V = Mesh2D->get_Elements();
    const real_vector_type& Vert = Mesh2D->get_Vertices();

    // Volume node (n,k) has id n + Np*k. Flat face-node index: n + Nfp*f + 3*Nfp*k.
    detail::parallelFor(K, [&](index_type k) {
        for (index_type f = 0; f < NumFaces; ++f) {
            const index_type k2 = E2E(NumFaces * k + f), f2 = E2F(NumFaces * k + f);
            // Reference length of the edge: distance between its two vertices.
            const index_type v1 = E2V(k * NumFaces + f), v2 = E2V(k * NumFaces + ((f + 1) % NumFaces));
            const real_type refd = std::hypot(Vert(3 * v1) - Vert(3 * v2), Vert(3 * v1 + 1) - Vert(3 * v2 + 1));
            for (index_type n = 0; n < Nfp; ++n) {
                const index_type flat = n + Nfp * f + NumFaces * Nfp * k;
                const index_type nodeM = Fmask(n, f);
                vmapM(flat) = nodeM + Np * k;
                const real_type x1 = xGrid(nodeM, k), y1 = yGrid(nodeM, k);
                // Scan the neighbour's face; the last node within tolerance wins and a
                // face node with no partner keeps 0 (reference :960-968, init :928).
                index_type vP = 0, mP = 0;
                for (index_type nP = 0; nP < Nfp; ++nP) {
                    const index_type nodeP = Fmask(nP, f2);
                    const real_type x2 = xGrid(nodeP, k2), y2 = yGrid(nodeP, k2);
                    if (std::hypot(x2 - x1, y2 - y1) < refd * NodeTol) {
                        vP = nodeP + Np * k2;
                        mP = nP + f2 * Nfp + k2 * NumFaces * Nfp;
                    }
                }
                vmapP(flat) = vP;
                mapP(flat) = mP;
            }
        }
    });

    // Boundary face nodes: vmapP == vmapM.
    const index_type total = K * NumFaces * Nfp;
    std::vector<index_type> bnd;
    for (index_type i = 0; i < total; ++i)
        if (vmapP(i) == vmapM(i)) bnd.push_back(i);
    mapB.resize(static_cast<index_type>(bnd.size()));
    vmapB.resize(static_cast<index_type>(bnd.size()));
    for (index_type i = 0; i < mapB.size(); ++i) {
        mapB(i) = bnd[i];
        vmapB(i) = vmapM(bnd[i]);
    }
    buildBCHash();
    gatherBuilt = false;
    gatherVec.clear();
    scatterVec.clear();
}

void TriangleNodesProvisioner::buildBCHash() { buildBCHash(Mesh2D->get_BCType()); }

void TriangleNodesProvisioner::buildBCHash(const index_vector_type& bcType) {
    // Every node of a face inherits the face's tag; nodes are listed in ascending
    // flat face-node order n + Nfp*(f + 3k).
    const index_type Nfp = NumFacePoints, faces = NumFaces * NumElements;
    if (bcType.size() != faces) throw std::runtime_error("buildBCHash: bcType must have NumFaces*NumElements entries");
    for (index_type g = 0; g < faces; ++g) {
        const index_type tag = bcType(g);
        if (tag == 0) continue;
        std::vector<index_type>& list = BCmap[tag];
        for (index_type n = 0; n < Nfp; ++n) list.push_back(g * Nfp + n);
    }
}

void TriangleNodesProvisioner::buildGatherScatter() const {
    // Unique physical nodes to tolerance 1e-9 (reference uniquetol call :1017):
    // gather[u] = one representative DG node id per unique point (column-wise
    // numbering), scatter[id] = index of its unique point. Two sorts: cluster the
    // x coordinates, then order by (x cluster, y) and merge neighbours in y.
    const index_type Np = NumLocalPoints, K = NumElements, total = Np * K;
    const real_type tol = 1.e-9;
    auto X = [&](index_type id) { return xGrid(id % Np, id / Np); };
    auto Y = [&](index_type id) { return yGrid(id % Np, id / Np); };
    std::vector<index_type> order(total), xcl(total);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](index_type a, index_type b) { return X(a) < X(b); });
    index_type cluster = 0;
    for (index_type i = 0; i < total; ++i) {
        if (i > 0 && X(order[i]) - X(order[i - 1]) > tol) ++cluster;
        xcl[order[i]] = cluster;
    }
    std::sort(order.begin(), order.end(), [&](index_type a, index_type b) {
        if (xcl[a] != xcl[b]) return xcl[a] < xcl[b];
        if (Y(a) != Y(b)) return Y(a) < Y(b);
        return a < b;
    });
    gatherVec.clear();
    scatterVec.assign(total, 0);
    for (index_type i = 0; i < total; ++i) {
        const index_type id = order[i];
        const bool same = i > 0 && xcl[id] == xcl[order[i - 1]] && Y(id) - Y(order[i - 1]) <= tol;
        if (!same) gatherVec.push_back(id);
        scatterVec[id] = static_cast<index_type>(gatherVec.size()) - 1;
    }
    gatherBuilt = true;
}

const std::vector<index_type>& TriangleNodesProvisioner::get_gather() const {
    if (!gatherBuilt) buildGatherScatter();
    return gatherVec;
}

const std::vector<index_type>& TriangleNodesProvisioner::get_scatter() const {
    if (!gatherBuilt) buildGatherScatter();
    return scatterVec;
}

void TriangleNodesProvisioner::computeInterpMatrix(const real_vector_type& rout, const real_vector_type& sout,
                                                   real_matrix_type& IM) const {
    real_matrix_type Vout(rout.size(), NumLocalPoints);
    computeVandermondeMatrix(NOrder, rout, sout, Vout);
    IM = matmul(Vout, Vinv);
}

void TriangleNodesProvisioner::splitOperators(real_matrix_type& IM, std::vector<index_type>& localE2V) const {
    const index_type N = NOrder, Np = NumLocalPoints;
    real_vector_type rout(Np), sout(Np);
    std::vector<index_type> counter(static_cast<std::size_t>(N + 1) * (N + 1), -1); // -1: no lattice point
    index_type count = 0;
    for (index_type n = 0; n < N + 1; ++n)
        for (index_type m = 0; m < N + 1 - n; ++m) {
            rout(count) = -1.0 + 2.0 * static_cast<real_type>(m) / static_cast<real_type>(N);
            sout(count) = -1.0 + 2.0 * static_cast<real_type>(n) / static_cast<real_type>(N);
            counter[n * (N + 1) + m] = count++;
        }
    IM.resize(Np, Np);
    computeInterpMatrix(rout, sout, IM);
    localE2V.clear();
    for (index_type n = 0; n < N; ++n)
        for (index_type m = 0; m < N - n; ++m) {
            const index_type v1 = counter[n * (N + 1) + m], v2 = counter[n * (N + 1) + m + 1],
                             v3 = counter[(n + 1) * (N + 1) + m], v4 = counter[(n + 1) * (N + 1) + m + 1];
            localE2V.insert(localE2V.end(), {v1, v2, v3});
            if (v4 >= 0) localE2V.insert(localE2V.end(), {v2, v4, v3});
        }
}

void TriangleNodesProvisioner::splitElements(const real_matrix_type& x, const real_matrix_type& y,
                                             const real_matrix_type& field, real_matrix_type& xnew,
                                             real_matrix_type& ynew, real_matrix_type& fieldnew) const {
    const index_type Np = field.rows(), K = field.cols();
    if (Np != NumLocalPoints || x.rows() != Np || y.rows() != Np || x.cols() != K || y.cols() != K)
        throw std::runtime_error("splitElements: x, y and field must be (Np, K)");
    real_matrix_type IM;
    std::vector<index_type> tri;
    splitOperators(IM, tri);
    const index_type nLocal = static_cast<index_type>(tri.size() / 3);
    auto interpolate = [&](const real_matrix_type& f) { // (Np, Np) x (Np, K), K contiguous
        real_matrix_type out(Np, K);
        detail::parallelFor(Np, [&](index_type i) {
            real_type* o = out.data() + static_cast<std::size_t>(i) * K;
            for (index_type m = 0; m < Np; ++m) {
                const real_type a = IM(i, m);
                const real_type* src = f.data() + static_cast<std::size_t>(m) * K;
                for (index_type k = 0; k < K; ++k) o[k] += a * src[k];
            }
        }, 1);
        return out;
    };
    const real_matrix_type rx_ = interpolate(x), ry_ = interpolate(y), rf = interpolate(field);
    xnew.resize(3, nLocal * K);
    ynew.resize(3, nLocal * K);
    fieldnew.resize(3, nLocal * K);
    for (index_type k = 0; k < K; ++k)
        for (index_type l = 0; l < nLocal; ++l)
            for (index_type c = 0; c < 3; ++c) {
                const index_type v = tri[3 * l + c], i = k * nLocal + l;
                xnew(c, i) = rx_(v, k);
                ynew(c, i) = ry_(v, k);
                fieldnew(c, i) = rf(v, k);
            }
}

DGContext2D TriangleNodesProvisioner::get_DGContext() const {
    // The gather/scatter maps need a sort over all Np*K nodes and are not read by the
    // RHS path: built eagerly only for small meshes, otherwise on get_gather().
    if (!gatherBuilt && static_cast<long long>(NumLocalPoints) * NumElements <= 2000000LL) buildGatherScatter();
    return DGContext2D(NOrder, NumLocalPoints, NumFacePoints, NumElements, NumFaces, &Filter, &rGrid, &sGrid,
                       &xGrid, &yGrid, &Fscale, &Fmask, &gatherVec, &scatterVec, &V, &Vinv, &J, &rx, &ry,
                       &sx, &sy, &nx, &ny, &Dr, &Ds, &Lift, &vmapM, &vmapP, &BCmap);
}

} // namespace blitzdg
