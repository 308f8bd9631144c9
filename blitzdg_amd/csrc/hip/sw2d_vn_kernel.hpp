// sw2d_vn_kernel.hpp -- variants B and D of the sw2d right-hand side on PER-NODE geometry tables.
//
// The reference's functions take whatever rx, sx, ry, sy, nx, ny, Fscale the context holds
// (swhelpers/rhs.py:178-311, src/sw2d/main.cpp:279-484) -- e.g. the nodal tables a provisioner carries after
// setCoordinates + buildCubatureVolumeMesh (src/TriangleNodesProvisioner.cpp:129-152). The fast kernels of variants
// B / C / D compress straight-sided geometry to 13 numbers per element; tables that are not of that kind used to be
// refused for them. This is their general form:
//     RHS_c[i] = -(rx_i (Dr F_c)_i + sx_i (Ds F_c)_i + ry_i (Dr G_c)_i + sy_i (Ds G_c)_i) + (Lift (Fscale dFlux_c))_i + S_c[i]
// with the metric at the OUTPUT node and normals / scales per face node. One field per wavefront, one lane per
// element (the layout of sw2d_stage_vd_kernel / sw2d_stage_vb_kernel): the fluxes F_c, G_c at the Np nodes and the 3 Nfp
// scaled flux jumps are formed once and kept in registers; the output rows are then produced one node at a time in a
// rolled loop whose operator rows -- Dr[i][.], Ds[i][.], Lift[i][.] -- are wave-uniform scalar loads, and each row is
// updated and stored as soon as it exists. The filtered RHS (Filter (flux terms + sources), what the drivers apply)
// cannot ride in the operators here -- the metric multiplies after the differentiation -- so it is a second pass
// (sw2d_filter_rows_kernel) over the unfiltered rows. A correctness path: nothing of it is tuned; the throughput
// kernels for curved meshes are those of sw2d_curved_nt_kernel.hpp.
#pragma once
#include "sw2d_vb_kernel.hpp"

namespace bdg_dev {

// operator image: row i = {Dr[i][m], Ds[i][m]} for m < Np, then Lift[i][j] for j < 3 Nfp
template <int N>
struct VnOps {
    using E = Elem<N>;
    static constexpr int ROW = 2 * E::Np + E::NFN;
    static constexpr int DOUBLES = ROW * E::Np;
};

// PHYS 1: variants C / D (VdParams: tracer as field 3, sources of swhelpers/rhs.py:300-309, per-face speed);
// PHYS 2: variant B (VbParams: depth traces, star states, open boundary, ONE global speed in *bp.lam, sources of
// src/sw2d/main.cpp:461-483). FILT: rows go unfiltered to p.rhs whatever MODE says (the filter pass does the update).
template <int N, int MODE, int PHYS>
__global__ __launch_bounds__(256, 1) void sw2d_stage_vn_kernel(const StageParams p, const VdParams vp, const VbParams bp,
                                                               const double* __restrict__ ops) {
    using E = Elem<N>;
    constexpr int Np = E::Np, Nfp = E::Nfp, NFN = E::NFN;
    const int c = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)) + (PHYS == 1 ? vp.cbase : 0);
    const unsigned k = static_cast<unsigned>(p.kbegin) + blockIdx.x * 64u + (threadIdx.x & 63u);
    if (k >= static_cast<unsigned>(p.kend)) return;
    const unsigned k8 = k * 8u, k4 = k * 4u;
    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld, fplane = static_cast<long long>(NFN) * ld;
    const double* __restrict__ qin = p.qin;
    const double g = p.g, halfg = 0.5 * p.g;

    // ---- fluxes of this wave's field at the nodes
    double F[Np], G[Np];
#pragma unroll
    for (int m = 0; m < Np; ++m) {
        const double h = ld_row(qin + m * ld, k8), hu = ld_row(qin + plane + m * ld, k8), hv = ld_row(qin + 2 * plane + m * ld, k8);
        const double r = fast_rcp(h);
        const double u = hu * r, v = hv * r;
        const double pr = halfg * h * h;
        if (c == 0) { F[m] = hu; G[m] = hv; }
        else if (c == 1) { F[m] = hu * u + pr; G[m] = hu * v; }
        else if (c == 2) { F[m] = PHYS == 2 ? hu * v : hv * u; G[m] = hv * v + pr; } // F3: flux.py:14 / main.cpp:382
        else { const double hN = ld_row(qin + 3 * plane + m * ld, k8); F[m] = hN * u; G[m] = hN * v; }
    }

    // ---- scaled flux jumps at the face nodes: s_j = Fscale_j/2 ((F_M - F_P) nx_j + (G_M - G_P) ny_j - lam (q_M - q_P))
    double s[NFN];
    const int tags = PHYS == 2 ? ld_row(bp.obc, k4) : 0;
    const double lamGlobal = PHYS == 2 ? *bp.lam : 0.0;
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        double hM[Nfp], huM[Nfp], hvM[Nfp], hP[Nfp], huP[Nfp], hvP[Nfp], rM[Nfp], rP[Nfp], nM[Nfp], nP[Nfp], nxv[Nfp], nyv[Nfp];
        double lam = lamGlobal;
#pragma unroll
        for (int n = 0; n < Nfp; ++n) {
            const int j = f * Nfp + n, m = E::fmask(f, n);
            const int id = ld_row(p.vmapP + j * ld, k4);
            nxv[n] = ld_row(p.fgeo + j * ld, k8);
            nyv[n] = ld_row(p.fgeo + fplane + j * ld, k8);
            nM[n] = nP[n] = 0.0;
            if constexpr (PHYS == 2) {
                const VbTrace t = vb_trace<N>(qin, bp.H, ld, plane, k8, m, id, (tags >> j) & 1, nxv[n], nyv[n], bp.tide);
                hM[n] = t.hM; huM[n] = t.huM; hvM[n] = t.hvM; hP[n] = t.hP; huP[n] = t.huP; hvP[n] = t.hvP; rM[n] = t.rM; rP[n] = t.rP;
            } else {
                hM[n] = ld_row(qin + m * ld, k8); huM[n] = ld_row(qin + plane + m * ld, k8); hvM[n] = ld_row(qin + 2 * plane + m * ld, k8);
                const unsigned o8 = static_cast<unsigned>(id < 0 ? -(id + 1) : id) * 8u;
                hP[n] = ld_row(qin, o8); huP[n] = ld_row(qin + plane, o8); hvP[n] = ld_row(qin + 2 * plane, o8);
                if (c == 3) { nM[n] = ld_row(qin + 3 * plane + m * ld, k8); nP[n] = ld_row(qin + 3 * plane, o8); }
                if (id < 0) { // reflective wall: no normal flow (rhs.py:236-237)
                    const double un = huM[n] * nxv[n] + hvM[n] * nyv[n];
                    huP[n] = huM[n] - 2 * nxv[n] * un;
                    hvP[n] = hvM[n] - 2 * nyv[n] * un;
                }
                rM[n] = fast_rcp(hM[n]);
                rP[n] = fast_rcp(hP[n]);
                const double uM = huM[n] * rM[n], vM = hvM[n] * rM[n], uP = huP[n] * rP[n], vP = hvP[n] * rP[n];
                const double spdM = fast_sqrt(uM * uM + vM * vM) + fast_sqrt(g * hM[n]);
                const double spdP = fast_sqrt(uP * uP + vP * vP) + fast_sqrt(g * hP[n]);
                lam = fmax(lam, fmax(spdM, spdP)); // the face's own maximum (rhs.py:262-265)
            }
        }
#pragma unroll
        for (int n = 0; n < Nfp; ++n) {
            const int j = f * Nfp + n;
            const double uM = huM[n] * rM[n], vM = hvM[n] * rM[n], uP = huP[n] * rP[n], vP = hvP[n] * rP[n];
            double dF, dG, dq;
            if (c == 0) { dF = huM[n] - huP[n]; dG = hvM[n] - hvP[n]; dq = hM[n] - hP[n]; }
            else if (c == 1) {
                dF = (huM[n] * uM + halfg * hM[n] * hM[n]) - (huP[n] * uP + halfg * hP[n] * hP[n]);
                dG = huM[n] * vM - huP[n] * vP;
                dq = huM[n] - huP[n];
            } else if (c == 2) {
                dF = PHYS == 2 ? huM[n] * vM - huP[n] * vP : hvM[n] * uM - hvP[n] * uP;
                dG = (hvM[n] * vM + halfg * hM[n] * hM[n]) - (hvP[n] * vP + halfg * hP[n] * hP[n]);
                dq = hvM[n] - hvP[n];
            } else {
                dF = nM[n] * uM - nP[n] * uP;
                dG = nM[n] * vM - nP[n] * vP;
                dq = nM[n] - nP[n];
            }
            s[j] = 0.5 * ld_row(p.fgeo + 2 * fplane + j * ld, k8) * (dF * nxv[n] + dG * nyv[n] - lam * dq);
        }
    }

    // ---- output rows, one node at a time
    const long long fo = static_cast<long long>(c) * plane;
    const bool src = PHYS == 2 ? c != 0 : (vp.sources != 0 && (c == 1 || c == 2));
#pragma unroll 1
    for (int i = 0; i < Np; ++i) {
        const double* __restrict__ row = ops + static_cast<size_t>(i) * VnOps<N>::ROW;
        const long long io = static_cast<long long>(i) * ld;
        const double rx = ld_row(p.geo + io, k8), sx = ld_row(p.geo + plane + io, k8), ry = ld_row(p.geo + 2 * plane + io, k8),
                     sy = ld_row(p.geo + 3 * plane + io, k8);
        double a = 0.0, b = 0.0, cG = 0.0, d = 0.0;
#pragma unroll
        for (int m = 0; m < Np; ++m) {
            a = fma(row[2 * m], F[m], a);
            b = fma(row[2 * m + 1], F[m], b);
            cG = fma(row[2 * m], G[m], cG);
            d = fma(row[2 * m + 1], G[m], d);
        }
        double R = -(rx * a + sx * b) - (ry * cG + sy * d);
        double lift = 0.0;
#pragma unroll
        for (int j = 0; j < NFN; ++j) lift = fma(row[2 * Np + j], s[j], lift);
        R += lift;
        if (src) {
            const double h = ld_row(qin + io, k8), hu = ld_row(qin + plane + io, k8), hv = ld_row(qin + 2 * plane + io, k8);
            const double r = fast_rcp(h);
            const double u = hu * r, v = hv * r, nrm = fast_sqrt(u * u + v * v);
            if constexpr (PHYS == 2) { // src/sw2d/main.cpp:461-483
                R += c == 1 ? g * h * ld_row(bp.Hx + io, k8) - bp.cd * u * nrm + bp.fcor * hv
                            : g * h * ld_row(bp.Hy + io, k8) - bp.cd * v * nrm - bp.fcor * hu;
            } else { // swhelpers/rhs.py:300-309 (the drag of RHS3 enters with the reference's sign)
                const double fc = vp.fcor ? ld_row(vp.fcor + io, k8) : vp.fconst;
                const double cdn = vp.cd * nrm;
                if (c == 1) R += (fc * hv - cdn * u) - (vp.zx ? g * h * ld_row(vp.zx + io, k8) : 0.0);
                else R += -(fc * hu - cdn * v) - (vp.zy ? g * h * ld_row(vp.zy + io, k8) : 0.0);
            }
        }
        if constexpr (MODE == MODE_RHS) {
            st_row(p.rhs + fo + io, k8, R);
        } else if constexpr (MODE == MODE_LSERK) {
            const double n1 = p.ca * ld_row(p.res + fo + io, k8) + p.cc * R;
            st_row(p.res + fo + io, k8, n1);
            st_row(p.qout + fo + io, k8, ld_row(qin + fo + io, k8) + p.cb * n1);
        } else {
            const double val = p.ca * ld_row(p.qbase + fo + io, k8) + p.cb * ld_row(qin + fo + io, k8) + p.cc * R;
            double sp = 0.0;
            if (PHYS == 2 ? c != 0 : (c == 1 || c == 2)) sp = (PHYS == 2 && bp.sponge) ? ld_row(bp.sponge + io, k8) : p.sponge;
            st_row(p.qout + fo + io, k8, sponge_relax(val, sp));
        }
    }
}

// ---- second pass of a FILTERED evaluation on per-node geometry: rows R of `raw` (nf fields) -> Filter R, then the
// update of MODE (same rules as above). filt: (Np, Np) row-major. One field per wavefront, one lane per element.
template <int N, int MODE, int PHYS>
__global__ __launch_bounds__(256, 1) void sw2d_filter_rows_kernel(const StageParams p, const double* __restrict__ raw,
                                                                  const double* __restrict__ filt, const double* __restrict__ spongeField,
                                                                  int cbase) {
    using E = Elem<N>;
    constexpr int Np = E::Np;
    const int c = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)) + cbase;
    const unsigned k = static_cast<unsigned>(p.kbegin) + blockIdx.x * 64u + (threadIdx.x & 63u);
    if (k >= static_cast<unsigned>(p.kend)) return;
    const unsigned k8 = k * 8u;
    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld, fo = static_cast<long long>(c) * plane;
    double R[Np];
#pragma unroll
    for (int m = 0; m < Np; ++m) R[m] = ld_row(raw + fo + m * ld, k8);
#pragma unroll 1
    for (int i = 0; i < Np; ++i) {
        const double* __restrict__ row = filt + static_cast<size_t>(i) * Np;
        const long long io = static_cast<long long>(i) * ld;
        double r = 0.0;
#pragma unroll
        for (int m = 0; m < Np; ++m) r = fma(row[m], R[m], r);
        if constexpr (MODE == MODE_RHS) {
            st_row(p.rhs + fo + io, k8, r);
        } else if constexpr (MODE == MODE_LSERK) {
            const double n1 = p.ca * ld_row(p.res + fo + io, k8) + p.cc * r;
            st_row(p.res + fo + io, k8, n1);
            st_row(p.qout + fo + io, k8, ld_row(p.qin + fo + io, k8) + p.cb * n1);
        } else {
            const double val = p.ca * ld_row(p.qbase + fo + io, k8) + p.cb * ld_row(p.qin + fo + io, k8) + p.cc * r;
            double sp = 0.0;
            if (PHYS == 2 ? c != 0 : (c == 1 || c == 2)) sp = (PHYS == 2 && spongeField) ? ld_row(spongeField + io, k8) : p.sponge;
            st_row(p.qout + fo + io, k8, sponge_relax(val, sp));
        }
    }
}

// ---- variant B's global speed on per-node normals (the reduction kernel is sw2d_vb_speed_reduce_kernel)
template <int N>
__global__ __launch_bounds__(256) void sw2d_vn_speed_kernel(const StageParams p, const VbParams vp, double* __restrict__ out) {
    using E = Elem<N>;
    constexpr int Np = E::Np, Nfp = E::Nfp, NFN = E::NFN;
    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld, fplane = static_cast<long long>(NFN) * ld;
    const unsigned k = static_cast<unsigned>(p.kbegin) + blockIdx.x * 256u + threadIdx.x;
    double best = 0.0;
    bool bad = false;
    if (k < static_cast<unsigned>(p.kend)) {
        const unsigned k8 = k * 8u, k4 = k * 4u;
        const int tags = ld_row(vp.obc, k4);
#pragma unroll 1
        for (int j = 0; j < NFN; ++j) {
            const int m = fmask_rt<N>(j / Nfp, j % Nfp);
            const int id = ld_row(p.vmapP + j * ld, k4);
            const VbTrace t = vb_trace<N>(p.qin, vp.H, ld, plane, k8, m, id, (tags >> j) & 1, ld_row(p.fgeo + j * ld, k8),
                                          ld_row(p.fgeo + fplane + j * ld, k8), vp.tide);
            const double uM = t.huM * t.rM, vM = t.hvM * t.rM, uP = t.huP * t.rP, vP = t.hvP * t.rP;
            const double spdM = sqrt(uM * uM + vM * vM) + sqrt(p.g * t.hM);
            const double spdP = sqrt(uP * uP + vP * vP) + sqrt(p.g * t.hP);
            if (spdM != spdM || spdP != spdP) bad = true;
            best = fmax(best, fmax(spdM, spdP));
        }
    }
    __shared__ double sA[256];
    __shared__ int sBad;
    if (threadIdx.x == 0) sBad = 0;
    __syncthreads();
    if (bad) sBad = 1;
    sA[threadIdx.x] = best;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (static_cast<int>(threadIdx.x) < st) sA[threadIdx.x] = fmax(sA[threadIdx.x], sA[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sBad ? __builtin_nan("") : sA[0];
}

} // namespace bdg_dev
