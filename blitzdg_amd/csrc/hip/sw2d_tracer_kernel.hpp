// sw2d_tracer_kernel.hpp -- the passive tracer equation (hN)_t + (hN u)_x + (hN v)_y = 0 of the
// reference's Python RHS (swhelpers/rhs.py:178-311, swhelpers/flux.py:17-19; sw2d.py:29-31), fourth
// field of a four-field solver, as its own fused RHS + stage pass. It runs after the three-field pass
// of sw2d_affine_kernel.hpp on the same input state and shares its structure (one lane per
// straight-sided element, everything unrolled, operators through scalar loads): the flow state and
// its neighbour traces are read again (from L2) because the Lax-Friedrichs speed of a face is the
// maximum over the flow's wave speeds there.
#pragma once
#include "sw2d_affine_kernel.hpp"

namespace bdg_dev {

template <int N, int MODE>
__global__ __launch_bounds__(256) void sw2d_stage_tracer_kernel(const StageParams p) {
    using E = Elem<N>;
    constexpr int Np = E::Np, Nfp = E::Nfp, NFN = E::NFN;

    const unsigned nwg = gridDim.x, xcd = blockIdx.x % 8u, q8 = nwg / 8u, r8 = nwg % 8u;
    const unsigned tile = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + blockIdx.x / 8u;
    const unsigned k = static_cast<unsigned>(p.kbegin) + tile * blockDim.x + threadIdx.x;
    if (k >= static_cast<unsigned>(p.kend)) return;
    const unsigned k8 = k * 8u, k4 = k * 4u;

    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld;
    const double* __restrict__ ops = p.opsAffine;
    const double* __restrict__ qin = p.qin;

    int idx[NFN];
#pragma unroll
    for (int j = 0; j < NFN; ++j) idx[j] = ld_row(p.vmapP + j * ld, k4);
    double h[Np], hu[Np], hv[Np], hN[Np];
#pragma unroll
    for (int n = 0; n < Np; ++n) {
        h[n] = ld_row(qin + n * ld, k8);
        hu[n] = ld_row(qin + plane + n * ld, k8);
        hv[n] = ld_row(qin + 2 * plane + n * ld, k8);
        hN[n] = ld_row(qin + 3 * plane + n * ld, k8);
    }
    const double* __restrict__ ag = p.ageo;
    const double rx = ld_row(ag, k8), sx = ld_row(ag + ld, k8), ry = ld_row(ag + 2 * ld, k8), sy = ld_row(ag + 3 * ld, k8);
    double fnx[3], fny[3], fsc[3];
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        fnx[f] = ld_row(ag + (4 + f) * ld, k8);
        fny[f] = ld_row(ag + (7 + f) * ld, k8);
        fsc[f] = ld_row(ag + (10 + f) * ld, k8);
    }
    __builtin_amdgcn_sched_barrier(0);
    double hP[NFN], huP[NFN], hvP[NFN], hNP[NFN];
#pragma unroll
    for (int j = 0; j < NFN; ++j) {
        const unsigned o8 = static_cast<unsigned>(idx[j] < 0 ? -(idx[j] + 1) : idx[j]) * 8u;
        hP[j] = ld_row(qin, o8);
        huP[j] = ld_row(qin + plane, o8);
        hvP[j] = ld_row(qin + 2 * plane, o8);
        hNP[j] = ld_row(qin + 3 * plane, o8);
    }
    __builtin_amdgcn_sched_barrier(0);

    const double g = p.g;
    double R[Np];
#pragma unroll
    for (int i = 0; i < Np; ++i) R[i] = 0.0;

    // ---- surface term
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        const double nxf = fnx[f], nyf = fny[f];
        double lam = 0.0;
        double uM[Nfp], vM[Nfp], uP[Nfp], vP[Nfp];
#pragma unroll
        for (int n = 0; n < Nfp; ++n) {
            const int j = f * Nfp + n, m = E::fmask(f, n);
            const double rM = fast_rcp(h[m]);
            uM[n] = hu[m] * rM;
            vM[n] = hv[m] * rM;
            const double spdM = fast_sqrt(uM[n] * uM[n] + vM[n] * vM[n]) + fast_sqrt(g * h[m]);
            double huq = huP[j], hvq = hvP[j];
            if (idx[j] < 0) { // reflective wall: no normal flow (the tracer trace is the element's own)
                const double un = hu[m] * nxf + hv[m] * nyf;
                huq = hu[m] - 2 * nxf * un;
                hvq = hv[m] - 2 * nyf * un;
            }
            const double r = fast_rcp(hP[j]);
            uP[n] = huq * r;
            vP[n] = hvq * r;
            const double spdP = fast_sqrt(uP[n] * uP[n] + vP[n] * vP[n]) + fast_sqrt(g * hP[j]);
            lam = fmax(lam, fmax(spdM, spdP));
        }
        const double half_fs = 0.5 * fsc[f];
#pragma unroll
        for (int n = 0; n < Nfp; ++n) {
            const int j = f * Nfp + n, m = E::fmask(f, n);
            const double dF = hN[m] * uM[n] - hNP[j] * uP[n], dG = hN[m] * vM[n] - hNP[j] * vP[n];
            const double s4 = half_fs * (dF * nxf + dG * nyf - lam * (hN[m] - hNP[j]));
#pragma unroll
            for (int i = 0; i < Np; ++i) R[i] = fma(ops[AffineOps<N>::OFF_LIFT + j * Np + i], s4, R[i]);
        }
    }

    const long long fo = 3 * plane;
    double oldv[Np];
    if constexpr (MODE != MODE_RHS) {
        const double* __restrict__ base2 = ((MODE == MODE_LSERK) ? p.res : p.qbase) + fo;
#pragma unroll
        for (int i = 0; i < Np; ++i) oldv[i] = ld_row(base2 + i * ld, k8);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- volume term
#pragma unroll
    for (int m = 0; m < Np; ++m) {
        const double r = fast_rcp(h[m] * p.one);
        const double F4 = hN[m] * (hu[m] * r), G4 = hN[m] * (hv[m] * r);
        const double a = -(rx * F4 + ry * G4), b = -(sx * F4 + sy * G4);
#pragma unroll
        for (int i = 0; i < Np; ++i) R[i] = fma(ops[AffineOps<N>::OFF_D + 2 * (m * Np + i)], a, R[i]);
#pragma unroll
        for (int i = 0; i < Np; ++i) R[i] = fma(ops[AffineOps<N>::OFF_D + 2 * (m * Np + i) + 1], b, R[i]);
    }

    // ---- stage update / output of the tracer field
    if constexpr (MODE == MODE_RHS) {
#pragma unroll
        for (int i = 0; i < Np; ++i) st_row(p.rhs + fo + i * ld, k8, R[i]);
    } else if constexpr (MODE == MODE_LSERK) {
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            const double n1 = p.ca * oldv[i] + p.cc * R[i];
            st_row(p.res + fo + i * ld, k8, n1);
            st_row(p.qout + fo + i * ld, k8, hN[i] + p.cb * n1);
        }
    } else {
#pragma unroll
        for (int i = 0; i < Np; ++i) st_row(p.qout + fo + i * ld, k8, p.ca * oldv[i] + p.cb * hN[i] + p.cc * R[i]);
    }
}

} // namespace bdg_dev
