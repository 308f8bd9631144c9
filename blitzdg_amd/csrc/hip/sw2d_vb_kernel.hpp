// sw2d_vb_kernel.hpp -- "variant B" of the sw2d right-hand side on the device: the physics of the
// reference's C++ `sw2d` driver,
//   src/sw2d/main.cpp:279-484  computeRHS(fields, numParams, physParams, DGContext2D, t)
// on top of variant A: still-water depth H with well-balanced star states at the faces (:356-368),
// open-boundary nodes driven by a tidal elevation (:348-353), ONE Lax-Friedrichs speed for the whole
// mesh (:414) and bed-slope / drag / Coriolis sources (:461-478).
//
// Two kernels per RHS evaluation:
//   sw2d_vb_speed_kernel   max over all face nodes of max(spd-, spd+) -> per-block partial maxima
//                          (finished by sw2d_vb_speed_reduce_kernel into one device scalar)
//   sw2d_stage_vb_kernel   the fused RHS (+ stage update), one field per wavefront as in
//                          sw2d_vd_kernel.hpp, reading that scalar
// Quirks kept: hM is overwritten by hMstar before the momentum rescale, so the rescale is
// hMstar*(huM/hMstar) (NaN for a dry star state) and the hydrostatic correction of :420-421 is
// identically zero (it is dropped here); the open-boundary assignment wins over the wall
// assignment where a node is in both lists (buildBCHash appends).
#pragma once
#include "sw2d_affine_kernel.hpp"
#include "sw2d_vd_kernel.hpp"

namespace bdg_dev {

struct VbParams {
    const double* H;       // (Np, ld) still-water depth
    const double* Hx;      // (Np, ld) bed slopes as the driver builds them (main.cpp:128-133)
    const double* Hy;
    const int* obc;        // (1, ld): bit j = face node j (0 .. 3Nfp-1) is an open-boundary node
    const double* lam;     // device scalar written by the speed pass
    const double* sponge;  // (Np, ld) sponge coefficient or nullptr (-> StageParams::sponge)
    double tide;           // open-boundary elevation of this evaluation
    double fcor, cd;
};

// '-' and '+' traces of one face node after boundary conditions and star states.
struct VbTrace {
    double hM, huM, hvM, hP, huP, hvP, rM, rP; // rM = 1/hM, rP = 1/hP
};

template <int N>
__device__ __forceinline__ VbTrace vb_trace(const double* __restrict__ qin, const double* __restrict__ H,
                                            long long ld, long long plane, unsigned k8, int m, int id, bool open,
                                            double nxf, double nyf, double tide) {
    VbTrace t;
    double hM = ld_row(qin + m * ld, k8);
    double huM = ld_row(qin + plane + m * ld, k8), hvM = ld_row(qin + 2 * plane + m * ld, k8);
    const double HM = ld_row(H + m * ld, k8);
    const unsigned o8 = static_cast<unsigned>(id < 0 ? -(id + 1) : id) * 8u;
    double hP = ld_row(qin, o8);
    double huP = ld_row(qin + plane, o8), hvP = ld_row(qin + 2 * plane, o8);
    const double HP = ld_row(H, o8);
    if (open) {                    // free surface follows the tide, momentum copied (:348-353)
        huP = huM;
        hvP = hvM;
        hP = HM + tide;
    } else if (id < 0) {           // reflective wall (:340-345)
        const double un = huM * nxf + hvM * nyf;
        hP = hM;
        huP = huM - 2 * nxf * un;
        hvP = hvM - 2 * nyf * un;
    }
    const double bM = -HM, bP = -HP, mx = fmax(bP, bM);
    const double hMs = fmax(0.0, hM + bM - mx), hPs = fmax(0.0, hP + bP - mx);
    t.rM = fast_rcp(hMs);
    t.rP = fast_rcp(hPs);
    t.hM = hMs;
    t.hP = hPs;
    t.huM = hMs * (huM * t.rM);    // hMstar*(huM/hM) with hM already = hMstar
    t.hvM = hMs * (hvM * t.rM);
    t.huP = hPs * (huP * t.rP);
    t.hvP = hPs * (hvP * t.rP);
    return t;
}

// ---- pass 1: global Lax-Friedrichs speed. One lane per element; out[b] = block maximum
//      (NaN if any speed is NaN).
template <int N>
__global__ __launch_bounds__(256) void sw2d_vb_speed_kernel(const StageParams p, const VbParams vp,
                                                            double* __restrict__ out) {
    using E = Elem<N>;
    constexpr int Np = E::Np, Nfp = E::Nfp;
    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld;
    const unsigned k = static_cast<unsigned>(p.kbegin) + blockIdx.x * 256u + threadIdx.x;
    double best = 0.0;
    bool bad = false;
    if (k < static_cast<unsigned>(p.kend)) {
        const unsigned k8 = k * 8u, k4 = k * 4u;
        const int tags = ld_row(vp.obc, k4);
#pragma unroll 1
        for (int f = 0; f < 3; ++f) {
            const double nxf = ld_row(p.ageo + (4 + f) * ld, k8), nyf = ld_row(p.ageo + (7 + f) * ld, k8);
#pragma unroll 1
            for (int n = 0; n < Nfp; ++n) {
                const int j = f * Nfp + n, m = fmask_rt<N>(f, n);
                const int id = ld_row(p.vmapP + j * ld, k4);
                const VbTrace t = vb_trace<N>(p.qin, vp.H, ld, plane, k8, m, id, (tags >> j) & 1, nxf, nyf, vp.tide);
                const double uM = t.huM * t.rM, vM = t.hvM * t.rM, uP = t.huP * t.rP, vP = t.hvP * t.rP;
                const double spdM = sqrt(uM * uM + vM * vM) + sqrt(p.g * t.hM);
                const double spdP = sqrt(uP * uP + vP * vP) + sqrt(p.g * t.hP);
                if (spdM != spdM || spdP != spdP) bad = true;
                best = fmax(best, fmax(spdM, spdP));
            }
        }
    }
    __shared__ double sA[256];
    __shared__ int sBad;
    if (threadIdx.x == 0) sBad = 0;
    __syncthreads();
    if (bad) sBad = 1;
    sA[threadIdx.x] = best;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (static_cast<int>(threadIdx.x) < s) sA[threadIdx.x] = fmax(sA[threadIdx.x], sA[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sBad ? __builtin_nan("") : sA[0];
}

template <int N> // (a template only so that every per-order translation unit may define it)
__global__ __launch_bounds__(256) void sw2d_vb_speed_reduce_kernel(const double* __restrict__ partials, int n,
                                                                   double* __restrict__ out) {
    __shared__ double sA[256];
    __shared__ int sBad;
    if (threadIdx.x == 0) sBad = 0;
    __syncthreads();
    double best = 0.0;
    bool bad = false;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double v = partials[i];
        if (v != v) bad = true;
        best = fmax(best, v);
    }
    if (bad) sBad = 1;
    sA[threadIdx.x] = best;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (static_cast<int>(threadIdx.x) < s) sA[threadIdx.x] = fmax(sA[threadIdx.x], sA[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sBad ? __builtin_nan("") : sA[0];
}

// ---- pass 2: fused RHS (+ stage update); wave c of a workgroup owns field c of 64 elements.
template <int N, int MODE>
__global__ __launch_bounds__(192, 2) void sw2d_stage_vb_kernel(const StageParams p, const VbParams vp) {
    using E = Elem<N>;
    constexpr int Np = E::Np, Nfp = E::Nfp;

    const unsigned nwg = gridDim.x, xcd = blockIdx.x % 8u, q8 = nwg / 8u, r8 = nwg % 8u;
    const unsigned tile = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + blockIdx.x / 8u;
    const int c = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // field of this wave (0..2)
    const unsigned k = static_cast<unsigned>(p.kbegin) + tile * 64u + (threadIdx.x & 63u);
    if (k >= static_cast<unsigned>(p.kend)) return;
    const unsigned k8 = k * 8u, k4 = k * 4u;

    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld;
    const double* __restrict__ ops = p.opsAffine; // VdOps<N> image
    const double* __restrict__ qin = p.qin;
    const double* __restrict__ ag = p.ageo;
    const double g = p.g, halfg = 0.5 * p.g;
    const double lam = *vp.lam;

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
            else { F = hu * v; G = hv * v + pr; }              // F3 is the same array as G2 (:382)
            const double a = -(rx * F + ry * G), b = -(sx * F + sy * G);
            const double* __restrict__ row = ops + VdOps<N>::OFF_D + 3 * m * Np;
            if (c != 0) {
                const double nrm = fast_sqrt(u * u + v * v);
                double S;
                if (c == 1) S = g * h * ld_row(vp.Hx + m * ld, k8) - vp.cd * u * nrm + vp.fcor * hv;
                else S = g * h * ld_row(vp.Hy + m * ld, k8) - vp.cd * v * nrm - vp.fcor * hu;
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
    {
        const int tags = ld_row(vp.obc, k4);
#pragma unroll 1
        for (int f = 0; f < 3; ++f) {
            const double nxf = ld_row(ag + (4 + f) * ld, k8), nyf = ld_row(ag + (7 + f) * ld, k8);
            const double half_fs = 0.5 * ld_row(ag + (10 + f) * ld, k8);
#pragma unroll 1
            for (int n = 0; n < Nfp; ++n) {
                const int j = f * Nfp + n, m = fmask_rt<N>(f, n);
                const int id = ld_row(p.vmapP + j * ld, k4);
                const VbTrace t = vb_trace<N>(qin, vp.H, ld, plane, k8, m, id, (tags >> j) & 1, nxf, nyf, vp.tide);
                const double uM = t.huM * t.rM, vM = t.hvM * t.rM, uP = t.huP * t.rP, vP = t.hvP * t.rP;
                double dF, dG, dq;
                if (c == 0) { dF = t.huM - t.huP; dG = t.hvM - t.hvP; dq = t.hM - t.hP; }
                else if (c == 1) {
                    dF = (t.huM * uM + halfg * t.hM * t.hM) - (t.huP * uP + halfg * t.hP * t.hP);
                    dG = t.huM * vM - t.huP * vP;
                    dq = t.huM - t.huP;
                } else {
                    dF = t.huM * vM - t.huP * vP;
                    dG = (t.hvM * vM + halfg * t.hM * t.hM) - (t.hvP * vP + halfg * t.hP * t.hP);
                    dq = t.hvM - t.hvP;
                }
                const double s = half_fs * (dF * nxf + dG * nyf - lam * dq);
                const double* __restrict__ row = ops + VdOps<N>::OFF_LIFT + j * Np;
#pragma unroll
                for (int i = 0; i < Np; ++i) R[i] = fma(row[i], s, R[i]);
            }
        }
    }

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
            double q1[CH], o1[CH], sp[CH];
#pragma unroll
            for (int t = 0; t < CH; ++t)
                if (i0 + t < Np) {
                    q1[t] = ld_row(qin + fo + (i0 + t) * ld, k8);
                    o1[t] = ld_row(base2 + (i0 + t) * ld, k8);
                    if constexpr (MODE == MODE_COMBINE)
                        sp[t] = (vp.sponge && c != 0) ? ld_row(vp.sponge + (i0 + t) * ld, k8) : p.sponge;
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
                        st_row(p.qout + fo + i * ld, k8, c != 0 ? sponge_relax(val, sp[t]) : val);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

} // namespace bdg_dev
