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
    // unrolled stage kernel only: the global speed of the NEXT evaluation, accumulated from the state this
    // launch writes (bit pattern of a non-negative double; atomic max; NaN bit patterns win). The '+' trace of an
    // interior face is the neighbour's '-' trace, so the maximum over all '-' star states plus the boundary '+'
    // states is the maximum the speed pass would find. nullptr: do not accumulate.
    unsigned long long* lamNext;
    double tideNext;       // open-boundary elevation assumed for that next evaluation
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
            // a face's nodes unrolled: its Nfp indices, then its gathers, are in flight together (rolled, every node was two
            // dependent round trips on its own: 0.147 ms at N = 6 on 5 10^5 elements)
            int ids[Nfp];
#pragma unroll
            for (int n = 0; n < Nfp; ++n) ids[n] = ld_row(p.vmapP + (f * Nfp + n) * ld, k4);
#pragma unroll
            for (int n = 0; n < Nfp; ++n) {
                const int j = f * Nfp + n, m = fmask_rt<N>(f, n);
                const int id = ids[n];
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

// ---- pass 2, unrolled form (N <= 5): the structure of sw2d_stage_affine_kernel -- one lane per element, all
// three fields, every load in a few large batches, operators through scalar loads -- with variant B's surface
// term (star states, open boundary, one global speed) and its sources seeded into R2, R3 node by node.
// FILTERED: the filtered RHS is Filter * (flux terms + sources); the operators are then the plain ones and the
// filter (vf) multiplies the finished R_c.
template <int N, int MODE, bool FILTERED>
__global__ __launch_bounds__(256) void sw2d_stage_vb_unrolled_kernel(const StageParams p, const VbParams vp,
                                                                      const double* __restrict__ vf) {
    using E = Elem<N>;
    constexpr int Np = E::Np, Nfp = E::Nfp, NFN = E::NFN;

    const unsigned nwg = gridDim.x, xcd = blockIdx.x % 8u, q8 = nwg / 8u, r8 = nwg % 8u;
    const unsigned tile = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + blockIdx.x / 8u;
    // Lanes past the end redo the last element (identical loads, identical stores) instead of leaving: the
    // wave-wide maximum at the end of the kernel then never reads a retired lane.
    const unsigned k = min(static_cast<unsigned>(p.kbegin) + tile * blockDim.x + threadIdx.x, static_cast<unsigned>(p.kend) - 1u);
    const unsigned k8 = k * 8u, k4 = k * 4u;

    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld;
    const double* __restrict__ ops = p.opsAffine; // AffineOps<N> image, plain operators
    const double* __restrict__ qin = p.qin;
    const double lam = *vp.lam;

    // ---- first batch: gather indices, own state, geometry, depth at the face nodes, bed slopes
    int idx[NFN];
#pragma unroll
    for (int j = 0; j < NFN; ++j) idx[j] = ld_row(p.vmapP + j * ld, k4);
    const int tags = ld_row(vp.obc, k4);
    double h[Np], hu[Np], hv[Np];
#pragma unroll
    for (int n = 0; n < Np; ++n) {
        h[n] = ld_row(qin + n * ld, k8);
        hu[n] = ld_row(qin + plane + n * ld, k8);
        hv[n] = ld_row(qin + 2 * plane + n * ld, k8);
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
    double HM[NFN], bx[Np], by[Np];
#pragma unroll
    for (int f = 0; f < 3; ++f)
#pragma unroll
        for (int n = 0; n < Nfp; ++n) HM[f * Nfp + n] = ld_row(vp.H + E::fmask(f, n) * ld, k8);
#pragma unroll
    for (int n = 0; n < Np; ++n) {
        bx[n] = ld_row(vp.Hx + n * ld, k8);
        by[n] = ld_row(vp.Hy + n * ld, k8);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- second batch: neighbour traces and neighbour depth
    double hP[NFN], huP[NFN], hvP[NFN], HP[NFN];
#pragma unroll
    for (int j = 0; j < NFN; ++j) {
        const unsigned o8 = static_cast<unsigned>(idx[j] < 0 ? -(idx[j] + 1) : idx[j]) * 8u;
        hP[j] = ld_row(qin, o8);
        huP[j] = ld_row(qin + plane, o8);
        hvP[j] = ld_row(qin + 2 * plane, o8);
        HP[j] = ld_row(vp.H, o8);
    }
    __builtin_amdgcn_sched_barrier(0);

    const double g = p.g, halfg = 0.5 * p.g;
    double R1[Np], R2[Np], R3[Np];
    // ---- sources (main.cpp:461-478): RHS2 += g h Hx - CD u|u| + f hv;  RHS3 += g h Hy - CD v|u| - f hu
#pragma unroll
    for (int m = 0; m < Np; ++m) {
        const double r = fast_rcp(h[m]);
        const double u = hu[m] * r, v = hv[m] * r;
        const double cdn = vp.cd * fast_sqrt(u * u + v * v);
        const double gh = g * h[m];
        R1[m] = 0.0;
        R2[m] = fma(gh, bx[m], fma(vp.fcor, hv[m], -(cdn * u)));
        R3[m] = fma(gh, by[m], -fma(vp.fcor, hu[m], cdn * v));
    }

    // ---- surface term with star states, face by face
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        const double nxf = fnx[f], nyf = fny[f], half_fs = 0.5 * fsc[f];
#pragma unroll
        for (int n = 0; n < Nfp; ++n) {
            const int j = f * Nfp + n, m = E::fmask(f, n);
            // open boundary: the surface follows the tide (:348-353); reflective wall (:340-345); else the neighbour's trace -- by
            // SELECTS: as an if / else-if per face node these were fifteen branches in the unrolled body (the scheduler lost its one
            // region: 398 scalar registers spilled to vector lanes, 22 branches in the RHS instance against 3 in variant A's kernel)
            double hq = hP[j], huq = huP[j], hvq = hvP[j];
            if constexpr (MODE == MODE_RHS) { // (the RHS-only instance keeps the tests: with the selects it spilled 126-134 registers, 42 so)
                if ((tags >> j) & 1) {
                    huq = hu[m];
                    hvq = hv[m];
                    hq = HM[j] + vp.tide;
                } else if (idx[j] < 0) {
                    const double un = hu[m] * nxf + hv[m] * nyf;
                    hq = h[m];
                    huq = hu[m] - 2 * nxf * un;
                    hvq = hv[m] - 2 * nyf * un;
                }
            } else {
                const bool open = ((tags >> j) & 1) != 0, wall = (idx[j] < 0) & !open;
                const double un2 = (open ? 0.0 : 2.0) * (hu[m] * nxf + hv[m] * nyf);   // open: the own momentum, wall: its mirror image
                const bool own = open | wall;
                hq = open ? HM[j] + vp.tide : (wall ? h[m] : hP[j]);
                huq = own ? hu[m] - nxf * un2 : huP[j];
                hvq = own ? hv[m] - nyf * un2 : hvP[j];
            }
            const double bM = -HM[j], bP = -HP[j], mx = fmax(bP, bM);
            const double hMs = fmax(0.0, h[m] + bM - mx), hPs = fmax(0.0, hq + bP - mx);
            const double rM = fast_rcp(hMs), rP = fast_rcp(hPs);
            const double huMs = hMs * (hu[m] * rM), hvMs = hMs * (hv[m] * rM);   // hMstar*(huM/hM), hM = hMstar
            const double huPs = hPs * (huq * rP), hvPs = hPs * (hvq * rP);
            const double uM = huMs * rM, vM = hvMs * rM, uP = huPs * rP, vP = hvPs * rP;
            const double prM = halfg * hMs * hMs, prP = halfg * hPs * hPs;
            const double F2M = huMs * uM + prM, G2M = huMs * vM, G3M = hvMs * vM + prM;
            const double F2P = huPs * uP + prP, G2P = huPs * vP, G3P = hvPs * vP + prP;
            const double dh = hMs - hPs, dhu = huMs - huPs, dhv = hvMs - hvPs;
            const double s1 = half_fs * (dhu * nxf + dhv * nyf - lam * dh);
            const double s2 = half_fs * ((F2M - F2P) * nxf + (G2M - G2P) * nyf - lam * dhu);
            const double s3 = half_fs * ((G2M - G2P) * nxf + (G3M - G3P) * nyf - lam * dhv);
#pragma unroll
            for (int i = 0; i < Np; ++i) {
                const double lj = ops[AffineOps<N>::OFF_LIFT + j * Np + i];
                R1[i] = fma(lj, s1, R1[i]);
                R2[i] = fma(lj, s2, R2[i]);
                R3[i] = fma(lj, s3, R3[i]);
            }
        }
    }

    // ---- stage inputs needed at the very end: issue now, land during the volume loop
    double old1[Np], old2[Np], old3[Np];
    if constexpr (MODE != MODE_RHS) {
        const double* __restrict__ b2 = (MODE == MODE_LSERK) ? p.res : p.qbase;
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            old1[i] = ld_row(b2 + i * ld, k8);
            old2[i] = ld_row(b2 + plane + i * ld, k8);
            old3[i] = ld_row(b2 + 2 * plane + i * ld, k8);
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- volume term
#pragma unroll
    for (int m = 0; m < Np; ++m) {
        const double r = fast_rcp(h[m] * p.one);
        const double u = hu[m] * r, v = hv[m] * r;
        const double pr = halfg * h[m] * h[m];
        const double F2 = hu[m] * u + pr, G2 = hu[m] * v, G3 = hv[m] * v + pr;
        const double a1 = -(rx * hu[m] + ry * hv[m]), b1 = -(sx * hu[m] + sy * hv[m]);
        const double a2 = -(rx * F2 + ry * G2), b2 = -(sx * F2 + sy * G2);
        const double a3 = -(rx * G2 + ry * G3), b3 = -(sx * G2 + sy * G3);
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            const double dr = ops[AffineOps<N>::OFF_D + 2 * (m * Np + i)];
            R1[i] = fma(dr, a1, R1[i]);
            R2[i] = fma(dr, a2, R2[i]);
            R3[i] = fma(dr, a3, R3[i]);
        }
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            const double ds = ops[AffineOps<N>::OFF_D + 2 * (m * Np + i) + 1];
            R1[i] = fma(ds, b1, R1[i]);
            R2[i] = fma(ds, b2, R2[i]);
            R3[i] = fma(ds, b3, R3[i]);
        }
    }

    if constexpr (FILTERED) {
        auto applyFilter = [&](double (&R)[Np]) {
            double T[Np];
#pragma unroll
            for (int i = 0; i < Np; ++i) T[i] = 0.0;
#pragma unroll
            for (int m = 0; m < Np; ++m)
#pragma unroll
                for (int i = 0; i < Np; ++i) T[i] = fma(vf[m * Np + i], R[m], T[i]);
#pragma unroll
            for (int i = 0; i < Np; ++i) R[i] = T[i];
        };
        applyFilter(R1);
        applyFilter(R2);
        applyFilter(R3);
    }

    // ---- stage update / output
    if constexpr (MODE == MODE_RHS) {
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            st_row(p.rhs + i * ld, k8, R1[i]);
            st_row(p.rhs + plane + i * ld, k8, R2[i]);
            st_row(p.rhs + 2 * plane + i * ld, k8, R3[i]);
        }
    } else if constexpr (MODE == MODE_LSERK) {
        const double a = p.ca, b = p.cb, dt = p.cc;
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            const double n1 = a * old1[i] + dt * R1[i], n2 = a * old2[i] + dt * R2[i], n3 = a * old3[i] + dt * R3[i];
            st_row(p.res + i * ld, k8, n1);
            st_row(p.res + plane + i * ld, k8, n2);
            st_row(p.res + 2 * plane + i * ld, k8, n3);
            h[i] += b * n1;
            hu[i] += b * n2;
            hv[i] += b * n3;
            st_row(p.qout + i * ld, k8, h[i]);
            st_row(p.qout + plane + i * ld, k8, hu[i]);
            st_row(p.qout + 2 * plane + i * ld, k8, hv[i]);
        }
    } else {
        // Sponge coefficients: the nodal field if there is one, else the scalar, picked by a select and not by a branch
        // (the load is issued either way), and the relaxation written without a test (x / (1 + 0 x^2) is x): a branch per
        // node in this unrolled body -- even ONE branch around the loads -- cost the scheduler its one big region, every
        // operator entry was hoisted and spilled (1314-1770 scalar, 432-474 vector registers).
        const double a = p.ca, b = p.cb, c = p.cc;
        double spv[Np];
        const bool field = vp.sponge != nullptr;
        const double* __restrict__ spp = field ? vp.sponge : p.qin; // (no field: any valid plane, the value is not used)
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            const double v = ld_row(spp + i * ld, k8);
            spv[i] = field ? v : p.sponge;
        }
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            const double v2 = a * old2[i] + b * hu[i] + c * R2[i], v3 = a * old3[i] + b * hv[i] + c * R3[i];
            h[i] = a * old1[i] + b * h[i] + c * R1[i];
            hu[i] = v2 / (1.0 + spv[i] * v2 * v2);
            hv[i] = v3 / (1.0 + spv[i] * v3 * v3);
            st_row(p.qout + i * ld, k8, h[i]);
            st_row(p.qout + plane + i * ld, k8, hu[i]);
            st_row(p.qout + 2 * plane + i * ld, k8, hv[i]);
        }
    }

    // ---- global speed of the next evaluation from the state just written (h, hu, hv now hold it). Round 4: the '-' star
    //      states of all face nodes in one branch-free loop (an interior '+' star state is the neighbour's '-' star state, which that
    //      element's lane covers), then ONE branch for the lanes that have a boundary node at all -- the rebuilt '+' states of
    //      open-boundary and wall nodes, by selects inside. Round 3 tested every node (fifteen divergent branches with IEEE
    //      square roots behind them in the unrolled body): 395-813 spilled scalar registers, 0.509 ms per LSERK4 stage at C3.
    //      The arithmetic (IEEE sqrt, contraction as the speed pass has it) is unchanged: the value equals sw2d_vb_speed_kernel's.
    if constexpr (MODE != MODE_RHS) {
        if (vp.lamNext) {
            double best = 0.0;
            bool anyBoundary = tags != 0;
#pragma unroll
            for (int f = 0; f < 3; ++f) {
#pragma unroll
                for (int n = 0; n < Nfp; ++n) {
                    const int j = f * Nfp + n, m = E::fmask(f, n);
                    const double HMj = ld_row(vp.H + m * ld, k8);
                    const double HPj = ld_row(vp.H, static_cast<unsigned>(idx[j] < 0 ? -(idx[j] + 1) : idx[j]) * 8u);
                    const double bM = -HMj, bP = -HPj, mx = fmax(bP, bM);
                    const double hMs = fmax(0.0, h[m] + bM - mx);
                    const double rM = fast_rcp(hMs);
                    const double huMs = hMs * (hu[m] * rM), hvMs = hMs * (hv[m] * rM);
                    const double uM = huMs * rM, vM = hvMs * rM;
                    const double spd = sqrt(uM * uM + vM * vM) + sqrt(g * hMs);
                    anyBoundary = anyBoundary | (idx[j] < 0);
                    best = ((spd != spd) | (best != best)) ? __builtin_nan("") : fmax(best, spd); // a NaN stays
                }
            }
            if (anyBoundary) { // rare: elements on the domain boundary
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    const double nxf = fnx[f], nyf = fny[f];
#pragma unroll
                    for (int n = 0; n < Nfp; ++n) {
                        const int j = f * Nfp + n, m = E::fmask(f, n);
                        const bool open = ((tags >> j) & 1) != 0, wall = idx[j] < 0;
                        const double HMj = ld_row(vp.H + m * ld, k8);
                        const double HPj = ld_row(vp.H, static_cast<unsigned>(idx[j] < 0 ? -(idx[j] + 1) : idx[j]) * 8u);
                        const double bM = -HMj, bP = -HPj, mx = fmax(bP, bM);
                        const double un = hu[m] * nxf + hv[m] * nyf;
                        const double hq = open ? HMj + vp.tideNext : h[m];
                        const double huq = open ? hu[m] : hu[m] - 2 * nxf * un;
                        const double hvq = open ? hv[m] : hv[m] - 2 * nyf * un;
                        const double hPs = fmax(0.0, hq + bP - mx);
                        const double rP = fast_rcp(hPs);
                        const double huPs = hPs * (huq * rP), hvPs = hPs * (hvq * rP);
                        const double uP = huPs * rP, vP = hvPs * rP;
                        const double spdP = sqrt(uP * uP + vP * vP) + sqrt(g * hPs);
                        const bool here = open | wall;
                        best = (here & ((spdP != spdP) | (best != best))) ? __builtin_nan("") : (here ? fmax(best, spdP) : best);
                    }
                }
            }
            unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(best));
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long other = __shfl_xor(bits, off);
                bits = other > bits ? other : bits;
            }
            if ((threadIdx.x & 63u) == 0u) atomicMax(vp.lamNext, bits);
        }
    }
}

} // namespace bdg_dev
