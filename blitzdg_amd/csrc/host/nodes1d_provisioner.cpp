// Nodes1DProvisioner implementation (setup path; CPU only).
// Semantics follow the reference's src/Nodes1DProvisioner.cpp:
// buildNodes :57-89, buildNormals :91-102, buildMaps :104-155,
// buildFaceMask :157-169, connectivity :171-238, buildLift :240-259,
// computeJacobian :261-281, buildDr :283-307.
#include "blitzdg/Nodes1DProvisioner.hpp"
#include <cmath>

namespace blitzdg {

const index_type Nodes1DProvisioner::NumFacePoints = 1;
const index_type Nodes1DProvisioner::NumFaces = 2;
const real_type Nodes1DProvisioner::NodeTol = 1.e-5;

Nodes1DProvisioner::Nodes1DProvisioner(index_type N, index_type K, real_type xmin, real_type xmax)
    : Min_x{xmin}, Max_x{xmax}, NumElements{K}, NOrder{N}, NumLocalPoints{N + 1},
      mapI{0}, mapO{NumFacePoints * NumFaces * K - 1}, vmapI{0}, vmapO{(N + 1) * K - 1},
      xGrid(N + 1, K), rGrid(N + 1), V(N + 1, N + 1), Dr(N + 1, N + 1),
      Lift(N + 1, NumFacePoints * NumFaces), J(N + 1, K), rx(N + 1, K),
      nx(NumFacePoints * NumFaces, K), Vinv(N + 1, N + 1), Fmask(NumFacePoints * NumFaces),
      Fx(NumFacePoints * NumFaces, K), Fscale(NumFacePoints * NumFaces, K),
      EToV(K, NumFaces), EToE(K, NumFaces), EToF(K, NumFaces),
      vmapM(NumFacePoints * NumFaces * K), vmapP(NumFacePoints * NumFaces * K) {}

void Nodes1DProvisioner::buildNodes() {
    Jacobi.computeGaussLobottoPoints(0.0, 0.0, NOrder, rGrid);
    Vandermonde.computeVandermondeMatrix(rGrid, V, Vinv);
    buildDr();
    buildLift();

    const real_type width = (Max_x - Min_x) / NumElements;
    for (index_type k = 0; k < NumElements; ++k)
        for (index_type n = 0; n < NumLocalPoints; ++n)
            xGrid(n, k) = Min_x + width * (k + 0.5 * (rGrid(n) + 1.));

    for (index_type k = 0; k < NumElements; ++k) {
        EToV(k, 0) = k;
        EToV(k, 1) = k + 1;
    }
    buildConnectivityMatrices();
    buildFaceMask();
    buildMaps();
    buildNormals();
}

void Nodes1DProvisioner::buildNormals() {
    for (index_type k = 0; k < NumElements; ++k) {
        nx(0, k) = -1.0; // left end: outward normal points to -x
        nx(1, k) = 1.0;
    }
}

void Nodes1DProvisioner::buildConnectivityMatrices() {
    // Vertex v is shared by the right face of element v-1 and the left face of
    // element v; the two domain ends stay self-connected.
    for (index_type k = 0; k < NumElements; ++k)
        for (index_type f = 0; f < NumFaces; ++f) {
            EToE(k, f) = k;
            EToF(k, f) = f;
        }
    for (index_type k = 0; k + 1 < NumElements; ++k) {
        EToE(k, 1) = k + 1; EToF(k, 1) = 0;
        EToE(k + 1, 0) = k; EToF(k + 1, 0) = 1;
    }
}

void Nodes1DProvisioner::buildFaceMask() {
    Fmask(0) = 0;
    Fmask(1) = NumLocalPoints - 1;
    for (index_type k = 0; k < NumElements; ++k)
        for (index_type f = 0; f < NumFacePoints * NumFaces; ++f) Fx(f, k) = xGrid(Fmask(f), k);
}

void Nodes1DProvisioner::buildMaps() {
    // Volume node id of node n of element k is n + Np*k (column-wise numbering).
    index_type count = 0;
    for (index_type k = 0; k < NumElements; ++k)
        for (index_type f = 0; f < NumFaces; ++f) vmapM(count++) = Fmask(f) + NumLocalPoints * k;

    vmapP.fill(0);
    count = 0;
    for (index_type k1 = 0; k1 < NumElements; ++k1)
        for (index_type f1 = 0; f1 < NumFaces; ++f1) {
            const index_type k2 = EToE(k1, f1), f2 = EToF(k1, f1);
            const index_type vidM = vmapM(k1 * NumFaces + f1), vidP = vmapM(k2 * NumFaces + f2);
            const real_type dx = xGrid(vidM % NumLocalPoints, vidM / NumLocalPoints) -
                                 xGrid(vidP % NumLocalPoints, vidP / NumLocalPoints);
            if (std::sqrt(dx * dx) < NodeTol) vmapP(count) = vidP;
            ++count;
        }
}

void Nodes1DProvisioner::buildLift() {
    // Lift = V (V^T E), E(0,0) = E(Np-1,1) = 1.
    const index_type Np = NumLocalPoints;
    real_matrix_type temp(Np, 2);
    for (index_type i = 0; i < Np; ++i) {
        temp(i, 0) = V(0, i);
        temp(i, 1) = V(Np - 1, i);
    }
    for (index_type i = 0; i < Np; ++i)
        for (index_type j = 0; j < 2; ++j) {
            real_type s = 0;
            for (index_type k = 0; k < Np; ++k) s += V(i, k) * temp(k, j);
            Lift(i, j) = s;
        }
}

void Nodes1DProvisioner::computeJacobian() {
    const index_type Np = NumLocalPoints;
    for (index_type i = 0; i < Np; ++i)
        for (index_type k = 0; k < NumElements; ++k) {
            real_type s = 0;
            for (index_type m = 0; m < Np; ++m) s += Dr(i, m) * xGrid(m, k);
            J(i, k) = s;
            rx(i, k) = 1 / s;
        }
    for (index_type f = 0; f < NumFaces; ++f)
        for (index_type k = 0; k < NumElements; ++k) Fscale(f, k) = 1 / J(Fmask(f), k);
}

void Nodes1DProvisioner::buildDr() {
    // Dr = DVr V^{-1}, obtained from V^T Dr^T = DVr^T.
    const index_type Np = NumLocalPoints;
    real_matrix_type DVr(Np, Np), Vt(Np, Np), DVrt(Np, Np), Drt(Np, Np);
    Vandermonde.computeGradVandermonde(rGrid, DVr);
    for (index_type i = 0; i < Np; ++i)
        for (index_type j = 0; j < Np; ++j) {
            Vt(i, j) = V(j, i);
            DVrt(i, j) = DVr(j, i);
        }
    LinSolver.solve(Vt, DVrt, Drt);
    for (index_type i = 0; i < Np; ++i)
        for (index_type j = 0; j < Np; ++j) Dr(i, j) = Drt(j, i);
}

} // namespace blitzdg
