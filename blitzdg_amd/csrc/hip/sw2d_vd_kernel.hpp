// sw2d_vd_kernel.hpp -- "variant D" of the sw2d right-hand side on the device: the 3 conserved
// fields plus a passive tracer hN, with Coriolis, quadratic drag and bed-slope sources, as the
// reference's Python implementation evaluates it:
//   swhelpers/rhs.py:178-311  sw2dComputeRHS(h,hu,hv,hN,zx,zy,g,H,f,CD,ctx,vmapM,vmapP)
//   swhelpers/flux.py:1-22    F4 = hN u, G4 = hN v
//   sources (rhs.py:300-309)  RHS2 += f hv - CD|u| u;  RHS3 -= f hu - CD|u| v  (sic: +CD|u| v);
//                             RHS2 -= g h zx;  RHS3 -= g h zy
// (sw2d.py:36-146, "variant C", is the same without drag / bed slope.)
//
// Structure: the rolled one-field-per-wave kernel of sw2d_affine_kernel.hpp with nf = 3 or 4
// wavefronts per 64 elements (wave c owns field c; c = 3 is the tracer), straight-sided elements.
// Sources enter through a third operator row in the volume loop, R_c[i] += F'[i][m] S_c[m], with
// F' = Filter when the caller asks for the filtered RHS (the reference drivers filter the whole
// RHS, sources included: sw2d.py:222-225) and F' = I otherwise.
#pragma once
#include "sw2d_affine_kernel.hpp"

namespace bdg_dev {

template <int N>
struct VdOps {
    using E = Elem<N>;
    // [m][i]{Dr'[i][m], Ds'[i][m], F'[i][m]} then [j][i] Lift'[i][j]
    static constexpr int OFF_D = 0;
    static constexpr int OFF_LIFT = 3 * E::Np * E::Np;
    static constexpr int DOUBLES = OFF_LIFT + E::NFN * E::Np;
};

struct VdParams {
    const double* zx;      // (Np, ld) planes or nullptr
    const double* zy;
    const double* fcor;    // (Np, ld) Coriolis parameter or nullptr -> fconst
    double fconst;
    double cd;             // drag coefficient CD
    int nf;                // 3 or 4 fields
    int sources;           // 0: none (variant A physics, optionally with tracer)
    int cbase;             // first field this launch computes (wave w owns field cbase + w); 3 = tracer only
};

template <int N, int MODE>
__global__ __launch_bounds__(256, 2) void sw2d_stage_vd_kernel(const StageParams p, const VdParams vp) {
    using E = Elem<N>;
    constexpr int Np = E::Np, Nfp = E::Nfp;

    const unsigned nwg = gridDim.x, xcd = blockIdx.x % 8u, q8 = nwg / 8u, r8 = nwg % 8u;
    const unsigned tile = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + blockIdx.x / 8u;
    const int c = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) + vp.cbase; // field of this wave (0..nf-1)
    const unsigned k = static_cast<unsigned>(p.kbegin) + tile * 64u + (threadIdx.x & 63u);
    if (k >= static_cast<unsigned>(p.kend)) return;
    const unsigned k8 = k * 8u, k4 = k * 4u;

    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld;
    const double* __restrict__ ops = p.opsAffine; // VdOps<N> image
    const double* __restrict__ qin = p.qin;
    const double* __restrict__ ag = p.ageo;
    const double g = p.g, halfg = 0.5 * p.g;
    const bool tracer = c == 3;
    const bool src = vp.sources != 0 && (c == 1 || c == 2);

    double R[Np];
#pragma unroll
    for (int i = 0; i < Np; ++i) R[i] = 0.0;

    // ---- volume term + sources, one input node per iteration
    {
        const double rx = ld_row(ag, k8), sx = ld_row(ag + ld, k8), ry = ld_row(ag + 2 * ld, k8),
                     sy = ld_row(ag + 3 * ld, k8);
#pragma unroll 1
        for (int m = 0; m < Np; ++m) {
            const double h = ld_row(qin + m * ld, k8), hu = ld_row(qin + plane + m * ld, k8),
                         hv = ld_row(qin + 2 * plane + m * ld, k8);
            const double r = fast_rcp(h);
            const double u = hu * r, v = hv * r;
            const double pr = halfg * h * h;
            double F, G;
            if (c == 0) { F = hu; G = hv; }
            else if (c == 1) { F = hu * u + pr; G = hu * v; }
            else if (c == 2) { F = hv * u; G = hv * v + pr; }   // flux.py: F3 = hv*u
            else { const double hN = ld_row(qin + 3 * plane + m * ld, k8); F = hN * u; G = hN * v; }
            const double a = -(rx * F + ry * G), b = -(sx * F + sy * G);
            const double* __restrict__ row = ops + VdOps<N>::OFF_D + 3 * m * Np;
            if (src) {
                const double fc = vp.fcor ? ld_row(vp.fcor + m * ld, k8) : vp.fconst;
                const double cdn = vp.cd * fast_sqrt(u * u + v * v);
                double S;
                if (c == 1) {
                    S = fc * hv - cdn * u;
                    if (vp.zx) S -= g * h * ld_row(vp.zx + m * ld, k8);
                } else {
                    S = -(fc * hu - cdn * v);
                    if (vp.zy) S -= g * h * ld_row(vp.zy + m * ld, k8);
                }
#pragma unroll
                for (int i = 0; i < Np; ++i)
                    R[i] = fma(row[3 * i + 2], S, fma(row[3 * i + 1], b, fma(row[3 * i], a, R[i])));
            } else {
#pragma unroll
                for (int i = 0; i < Np; ++i) R[i] = fma(row[3 * i + 1], b, fma(row[3 * i], a, R[i]));
            }
        }
    }

    // ---- surface term
#pragma unroll 1
    for (int f = 0; f < 3; ++f) {
        const double nxf = ld_row(ag + (4 + f) * ld, k8), nyf = ld_row(ag + (7 + f) * ld, k8);
        const double half_fs = 0.5 * ld_row(ag + (10 + f) * ld, k8);
        double lam = 0.0;
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll 1
            for (int n = 0; n < Nfp; ++n) {
                const int j = f * Nfp + n, m = fmask_rt<N>(f, n);
                const int id = ld_row(p.vmapP + j * ld, k4);
                const double hM = ld_row(qin + m * ld, k8), huM = ld_row(qin + plane + m * ld, k8),
                             hvM = ld_row(qin + 2 * plane + m * ld, k8);
                const unsigned o8 = static_cast<unsigned>(id < 0 ? -(id + 1) : id) * 8u;
                const double hq = ld_row(qin, o8);
                double huq = ld_row(qin + plane, o8), hvq = ld_row(qin + 2 * plane, o8);
                if (id < 0) { // reflective wall: no normal flow
                    const double un = huM * nxf + hvM * nyf;
                    huq = huM - 2 * nxf * un;
                    hvq = hvM - 2 * nyf * un;
                }
                const double rM = fast_rcp(hM), rP = fast_rcp(hq);
                const double uM = huM * rM, vM = hvM * rM, uP = huq * rP, vP = hvq * rP;
                if (pass == 0) {
                    const double spdM = fast_sqrt(uM * uM + vM * vM) + fast_sqrt(g * hM);
                    const double spdP = fast_sqrt(uP * uP + vP * vP) + fast_sqrt(g * hq);
                    lam = fmax(lam, fmax(spdM, spdP));
                } else {
                    double dF, dG, dq;
                    if (c == 0) { dF = huM - huq; dG = hvM - hvq; dq = hM - hq; }
                    else if (c == 1) {
                        dF = (huM * uM + halfg * hM * hM) - (huq * uP + halfg * hq * hq);
                        dG = huM * vM - huq * vP;
                        dq = huM - huq;
                    } else if (c == 2) {
                        dF = hvM * uM - hvq * uP;
                        dG = (hvM * vM + halfg * hM * hM) - (hvq * vP + halfg * hq * hq);
                        dq = hvM - hvq;
                    } else {
                        const double nM = ld_row(qin + 3 * plane + m * ld, k8), nP = ld_row(qin + 3 * plane, o8);
                        dF = nM * uM - nP * uP;
                        dG = nM * vM - nP * vP;
                        dq = nM - nP;
                    }
                    const double s = half_fs * (dF * nxf + dG * nyf - lam * dq);
                    const double* __restrict__ row = ops + VdOps<N>::OFF_LIFT + j * Np;
#pragma unroll
                    for (int i = 0; i < Np; ++i) R[i] = fma(row[i], s, R[i]);
                }
            }
        }
    }
    (void)tracer;

    // ---- stage update / output of this wave's field
    const long long fo = static_cast<long long>(c) * plane;
    if constexpr (MODE == MODE_RHS) {
#pragma unroll
        for (int i = 0; i < Np; ++i) st_row(p.rhs + fo + i * ld, k8, R[i]);
    } else {
        constexpr int CH = 7;
        const double* __restrict__ base2 = ((MODE == MODE_LSERK) ? p.res : p.qbase) + fo;
        const double a = p.ca, b = p.cb, cc = p.cc;
#pragma unroll
        for (int i0 = 0; i0 < Np; i0 += CH) {
            double q1[CH], o1[CH];
#pragma unroll
            for (int t = 0; t < CH; ++t)
                if (i0 + t < Np) {
                    q1[t] = ld_row(qin + fo + (i0 + t) * ld, k8);
                    o1[t] = ld_row(base2 + (i0 + t) * ld, k8);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < CH; ++t) {
                const int i = i0 + t;
                if (i < Np) {
                    if constexpr (MODE == MODE_LSERK) {
                        const double n1 = a * o1[t] + cc * R[i];
                        st_row(p.res + fo + i * ld, k8, n1);
                        st_row(p.qout + fo + i * ld, k8, q1[t] + b * n1);
                    } else {
                        const double val = a * o1[t] + b * q1[t] + cc * R[i];
                        st_row(p.qout + fo + i * ld, k8, (c == 1 || c == 2) ? sponge_relax(val, p.sponge) : val);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

} // namespace bdg_dev
