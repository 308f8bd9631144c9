// sw2d_affine_lean_kernel.hpp -- A/B variant 8 of the straight-element stage kernel (round 4): the unrolled kernel's arithmetic
// in a register budget that lets TWO waves share a SIMD, without reading the state again.
//
// sw2d_stage_affine_kernel holds, per lane, its element's state (3 Np doubles), the three faces' neighbour traces (9 Nfp), the
// accumulators (3 Np) and the residual rows (3 Np) at once: 256 VGPR + 154 AGPR at N = 4, one wave per SIMD, nothing to cover a
// wave's memory round trips but the other SIMDs' traffic. The streamed variants (2, 3) fit two or three waves by reading the
// state three times. This one keeps state and accumulators resident (6 Np doubles = 180 registers at N = 4) and gives up the
// rest: neighbour traces are gathered one face at a time (requesting the next face's behind the current face's lift products
// spilled 178 registers), velocities are formed twice per face node instead of kept, and the residual rows arrive in batches
// of five nodes during the update -- latencies the partner wave is there to cover.
// Same operator image (AffineOps, plain or pre-filtered), same node-by-node arithmetic as variant 0, but the volume term is
// accumulated before the surface term (variant 0: after it): results equal variant 0's to round-off, not bit for bit.
// LSERK stages only (what the benchmark times); selected with BDG_SW2D_AFFINE_VARIANT=8.
#pragma once
#include "sw2d_affine_kernel.hpp"

namespace bdg_dev {

template <int N>
__global__ __launch_bounds__(64, 2) void sw2d_stage_affine_lean_kernel(const StageParams p) {
    using E = Elem<N>;
    constexpr int Np = E::Np, Nfp = E::Nfp;

    const unsigned nwg = gridDim.x, xcd = blockIdx.x % 8u, q8 = nwg / 8u, r8 = nwg % 8u;
    const unsigned tile = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + blockIdx.x / 8u;
    const unsigned k = static_cast<unsigned>(p.kbegin) + tile * blockDim.x + threadIdx.x;
    if (k >= static_cast<unsigned>(p.kend)) return;
    const unsigned k8 = k * 8u, k4 = k * 4u;

    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld;
    const double* __restrict__ ops = p.opsAffine;
    const double* __restrict__ qin = p.qin;
    const double* __restrict__ ag = p.ageo;
    const double g = p.g, halfg = 0.5 * p.g;

    // ---- first batch: own state and the volume metric
    double h[Np], hu[Np], hv[Np];
#pragma unroll
    for (int n = 0; n < Np; ++n) {
        h[n] = ld_row(qin + n * ld, k8);
        hu[n] = ld_row(qin + plane + n * ld, k8);
        hv[n] = ld_row(qin + 2 * plane + n * ld, k8);
    }
    __builtin_amdgcn_sched_barrier(0);

    double R1[Np], R2[Np], R3[Np];
#pragma unroll
    for (int i = 0; i < Np; ++i) R1[i] = R2[i] = R3[i] = 0.0;

    // ---- volume term (as variant 0)
    {
        const double rx = ld_row(ag, k8), sx = ld_row(ag + ld, k8), ry = ld_row(ag + 2 * ld, k8), sy = ld_row(ag + 3 * ld, k8);
#pragma unroll
        for (int m = 0; m < Np; ++m) {
            const double r = fast_rcp(h[m]);
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
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- surface term, face by face: only one face's neighbour traces are in registers at a time, and the velocities are formed
    //      again in the second pass instead of being kept (as the streamed variants do); '-' traces come from the resident state
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        const double nxf = ld_row(ag + (4 + f) * ld, k8), nyf = ld_row(ag + (7 + f) * ld, k8);
        const double half_fs = 0.5 * ld_row(ag + (10 + f) * ld, k8);
        int idx[Nfp];
        double hq[Nfp], huq[Nfp], hvq[Nfp];
#pragma unroll
        for (int n = 0; n < Nfp; ++n) idx[n] = ld_row(p.vmapP + (f * Nfp + n) * ld, k4);
#pragma unroll
        for (int n = 0; n < Nfp; ++n) {
            const unsigned o8 = static_cast<unsigned>(idx[n] < 0 ? -(idx[n] + 1) : idx[n]) * 8u;
            hq[n] = ld_row(qin, o8);
            huq[n] = ld_row(qin + plane, o8);
            hvq[n] = ld_row(qin + 2 * plane, o8);
        }
        __builtin_amdgcn_sched_barrier(0);
        double lam = 0.0;
#pragma unroll
        for (int n = 0; n < Nfp; ++n) {
            const int m = E::fmask(f, n);
            if (idx[n] < 0) { // reflective wall: no normal flow
                const double un = hu[m] * nxf + hv[m] * nyf;
                huq[n] = hu[m] - 2 * nxf * un;
                hvq[n] = hv[m] - 2 * nyf * un;
            }
            const double rM = fast_rcp(h[m]), rP = fast_rcp(hq[n]);
            const double uM = hu[m] * rM, vM = hv[m] * rM, uP = huq[n] * rP, vP = hvq[n] * rP;
            const double spdM = fast_sqrt(uM * uM + vM * vM) + fast_sqrt(g * h[m]);
            const double spdP = fast_sqrt(uP * uP + vP * vP) + fast_sqrt(g * hq[n]);
            lam = fmax(lam, fmax(spdM, spdP));
        }
#pragma unroll
        for (int n = 0; n < Nfp; ++n) {
            const int j = f * Nfp + n, m = E::fmask(f, n);
            const double rM = fast_rcp(h[m] * p.one), rP = fast_rcp(hq[n] * p.one);
            const double uM = hu[m] * rM, vM = hv[m] * rM, uP = huq[n] * rP, vP = hvq[n] * rP;
            const double prM = halfg * h[m] * h[m], prP = halfg * hq[n] * hq[n];
            const double F2M = hu[m] * uM + prM, G2M = hu[m] * vM, G3M = hv[m] * vM + prM;
            const double F2P = huq[n] * uP + prP, G2P = huq[n] * vP, G3P = hvq[n] * vP + prP;
            const double dh = h[m] - hq[n], dhu = hu[m] - huq[n], dhv = hv[m] - hvq[n];
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
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- LSERK4 stage update, residual rows five nodes at a time (own state from registers)
    constexpr int CH = 5;
    double* __restrict__ rs = p.res;
    double* __restrict__ o = p.qout;
    const double a = p.ca, b = p.cb, dt = p.cc;
#pragma unroll
    for (int i0 = 0; i0 < Np; i0 += CH) {
        double o1[CH], o2[CH], o3[CH];
#pragma unroll
        for (int t = 0; t < CH; ++t) {
            const int i = i0 + t;
            if (i < Np) {
                o1[t] = ld_row(rs + i * ld, k8);
                o2[t] = ld_row(rs + plane + i * ld, k8);
                o3[t] = ld_row(rs + 2 * plane + i * ld, k8);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < CH; ++t) {
            const int i = i0 + t;
            if (i < Np) {
                const double n1 = a * o1[t] + dt * R1[i], n2 = a * o2[t] + dt * R2[i], n3 = a * o3[t] + dt * R3[i];
                st_row(rs + i * ld, k8, n1);
                st_row(rs + plane + i * ld, k8, n2);
                st_row(rs + 2 * plane + i * ld, k8, n3);
                st_row(o + i * ld, k8, h[i] + b * n1);
                st_row(o + plane + i * ld, k8, hu[i] + b * n2);
                st_row(o + 2 * plane + i * ld, k8, hv[i] + b * n3);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

} // namespace bdg_dev
