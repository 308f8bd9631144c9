// sw2d_affine_kernel.hpp -- fused RHS + stage update for straight-sided
// (affine) elements, the case every table built by the reference's
// TriangleNodesProvisioner falls in: rx, sx, ry, sy are constant per element and
// nx, ny, Fscale constant per face (reference src/TriangleNodesProvisioner.cpp
// :807-892 computes them from an affine map and stores them per node).
//
// Same mathematics as sw2d_stage_kernel (reference src/sw2d-simple/main.cpp
// :181-356 + stage update), reorganised for the register file:
//   * "outer-product" form: loop over the INPUT node m, form the contravariant
//     fluxes a_c = rx F_c + ry G_c, b_c = sx F_c + sy G_c once, and accumulate
//     R_c[i] -= Dr[i][m] a_c + Ds[i][m] b_c into the 3*Np outputs. Only the
//     outputs are hot; flux columns are never stored, and the flop count halves
//     (2 instead of 4 FMAs per (i, m, field));
//   * the lifted surface term is accumulated the same way, face node by face
//     node: R_c[i] += Lift[i][j] s_c[j];
//   * operator entries are wave-uniform: they are read with scalar loads into
//     SGPRs and used directly as FMA operands (no LDS traffic, no VGPRs);
//   * the neighbour-trace gathers are issued before the volume loop and land
//     while it runs;
//   * the modal filter folds into the operators (Filt*Dr, Filt*Ds, Filt*Lift are
//     prepared on the host), so a filtered RHS costs nothing extra.
// HBM traffic per element and stage: q 360 + res 360 in, res 360 + q 360 out,
// geometry 104, gather index 60 (+ neighbour traces through L2) = 1604 bytes.
#pragma once
#include "sw2d_kernels.hpp"

namespace bdg_dev {

template <int N>
struct AffineOps {
    using E = Elem<N>;
    // [m][i]{Dr[i][m], Ds[i][m]} then [j][i] Lift[i][j]
    static constexpr int OFF_D = 0;
    static constexpr int OFF_LIFT = 2 * E::Np * E::Np;
    static constexpr int DOUBLES = OFF_LIFT + E::NFN * E::Np;
};

template <int N, int MODE>
__global__ __launch_bounds__(256) void sw2d_stage_affine_kernel(const StageParams p) {
    using E = Elem<N>;
    constexpr int Np = E::Np, Nfp = E::Nfp, NFN = E::NFN;

    const unsigned nwg = gridDim.x, xcd = blockIdx.x % 8u, q8 = nwg / 8u, r8 = nwg % 8u;
    const unsigned tile = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + blockIdx.x / 8u;
    const long long k = p.kbegin + static_cast<long long>(tile) * blockDim.x + threadIdx.x;
    if (k >= p.kend) return;

    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld;
    const double* __restrict__ ops = p.opsAffine; // wave-uniform reads -> scalar loads
    const double* __restrict__ qh = p.qin + k;

    // ---- issue the independent loads: gather indices, own state, element geometry
    int idx[NFN];
#pragma unroll
    for (int j = 0; j < NFN; ++j) idx[j] = p.vmapP[j * ld + k];
    double h[Np], hu[Np], hv[Np];
#pragma unroll
    for (int n = 0; n < Np; ++n) {
        h[n] = qh[n * ld];
        hu[n] = qh[plane + n * ld];
        hv[n] = qh[2 * plane + n * ld];
    }
    const double* __restrict__ ag = p.ageo + k;
    const double rx = ag[0], sx = ag[ld], ry = ag[2 * ld], sy = ag[3 * ld];
    double fnx[3], fny[3], fsc[3];
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        fnx[f] = ag[(4 + f) * ld];
        fny[f] = ag[(7 + f) * ld];
        fsc[f] = ag[(10 + f) * ld];
    }
    // Keep the loads above in one batch: without this the scheduler sinks each load next to
    // its first use to save registers and the wave pays one memory round trip per node.
    __builtin_amdgcn_sched_barrier(0);
    // ---- neighbour ('+') traces: in flight during the volume loop
    double hP[NFN], huP[NFN], hvP[NFN];
#pragma unroll
    for (int j = 0; j < NFN; ++j) {
        const int o = idx[j] < 0 ? -(idx[j] + 1) : idx[j];
        hP[j] = p.qin[o];
        huP[j] = p.qin[plane + o];
        hvP[j] = p.qin[2 * plane + o];
    }

    __builtin_amdgcn_sched_barrier(0);
    const double g = p.g, halfg = 0.5 * p.g;
    double R1[Np], R2[Np], R3[Np];
#pragma unroll
    for (int i = 0; i < Np; ++i) R1[i] = R2[i] = R3[i] = 0.0;

    // ---- surface term, face by face
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        const double nxf = fnx[f], nyf = fny[f];
        double lam = 0.0;
        double uM[Nfp], vM[Nfp], uP[Nfp], vP[Nfp], hq[Nfp], huq[Nfp], hvq[Nfp];
#pragma unroll
        for (int n = 0; n < Nfp; ++n) {
            const int j = f * Nfp + n, m = E::fmask(f, n);
            const double rM = 1.0 / h[m];
            uM[n] = hu[m] * rM;
            vM[n] = hv[m] * rM;
            const double spdM = sqrt(uM[n] * uM[n] + vM[n] * vM[n]) + sqrt(g * h[m]);
            hq[n] = hP[j];
            huq[n] = huP[j];
            hvq[n] = hvP[j];
            if (idx[j] < 0) { // reflective wall: no normal flow
                const double un = hu[m] * nxf + hv[m] * nyf;
                huq[n] = hu[m] - 2 * nxf * un;
                hvq[n] = hv[m] - 2 * nyf * un;
            }
            const double r = 1.0 / hq[n];
            uP[n] = huq[n] * r;
            vP[n] = hvq[n] * r;
            const double spdP = sqrt(uP[n] * uP[n] + vP[n] * vP[n]) + sqrt(g * hq[n]);
            lam = fmax(lam, fmax(spdM, spdP));
        }
        const double half_fs = 0.5 * fsc[f];
#pragma unroll
        for (int n = 0; n < Nfp; ++n) {
            const int j = f * Nfp + n, m = E::fmask(f, n);
            const double prM = halfg * h[m] * h[m], prP = halfg * hq[n] * hq[n];
            const double F2M = hu[m] * uM[n] + prM, G2M = hu[m] * vM[n], G3M = hv[m] * vM[n] + prM;
            const double F2P = huq[n] * uP[n] + prP, G2P = huq[n] * vP[n], G3P = hvq[n] * vP[n] + prP;
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
    }

    // ---- stage inputs that are only needed at the very end: issue now, land during the volume loop
    double old1[Np], old2[Np], old3[Np];
    if constexpr (MODE == MODE_LSERK) {
        const double* __restrict__ rs = p.res + k;
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            old1[i] = rs[i * ld];
            old2[i] = rs[plane + i * ld];
            old3[i] = rs[2 * plane + i * ld];
        }
    } else if constexpr (MODE == MODE_COMBINE) {
        const double* __restrict__ qb = p.qbase + k;
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            old1[i] = qb[i * ld];
            old2[i] = qb[plane + i * ld];
            old3[i] = qb[2 * plane + i * ld];
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- volume term, one input node at a time
#pragma unroll
    for (int m = 0; m < Np; ++m) {
        // The reciprocal is recomputed here (from h * 1.0 with a run-time 1.0, so the compiler
        // cannot reuse the surface term's value): keeping Np reciprocals live across the
        // surface term costs more in register traffic than Np divisions.
        const double r = 1.0 / (h[m] * p.one);
        const double u = hu[m] * r, v = hv[m] * r;
        const double pr = halfg * h[m] * h[m];
        const double F2 = hu[m] * u + pr, G2 = hu[m] * v, G3 = hv[m] * v + pr;
        const double a1 = -(rx * hu[m] + ry * hv[m]), b1 = -(sx * hu[m] + sy * hv[m]);
        const double a2 = -(rx * F2 + ry * G2), b2 = -(sx * F2 + sy * G2);
        const double a3 = -(rx * G2 + ry * G3), b3 = -(sx * G2 + sy * G3);
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            const double dr = ops[AffineOps<N>::OFF_D + 2 * (m * Np + i)];
            const double ds = ops[AffineOps<N>::OFF_D + 2 * (m * Np + i) + 1];
            R1[i] = fma(ds, b1, fma(dr, a1, R1[i]));
            R2[i] = fma(ds, b2, fma(dr, a2, R2[i]));
            R3[i] = fma(ds, b3, fma(dr, a3, R3[i]));
        }
    }

    // ---- stage update / output
    if constexpr (MODE == MODE_RHS) {
        double* __restrict__ o = p.rhs + k;
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            o[i * ld] = R1[i];
            o[plane + i * ld] = R2[i];
            o[2 * plane + i * ld] = R3[i];
        }
    } else if constexpr (MODE == MODE_LSERK) {
        double* __restrict__ rs = p.res + k;
        double* __restrict__ o = p.qout + k;
        const double a = p.ca, b = p.cb, dt = p.cc;
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            const double n1 = a * old1[i] + dt * R1[i];
            const double n2 = a * old2[i] + dt * R2[i];
            const double n3 = a * old3[i] + dt * R3[i];
            rs[i * ld] = n1;
            rs[plane + i * ld] = n2;
            rs[2 * plane + i * ld] = n3;
            o[i * ld] = h[i] + b * n1;
            o[plane + i * ld] = hu[i] + b * n2;
            o[2 * plane + i * ld] = hv[i] + b * n3;
        }
    } else {
        double* __restrict__ o = p.qout + k;
        const double a = p.ca, b = p.cb, c = p.cc;
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            o[i * ld] = a * old1[i] + b * h[i] + c * R1[i];
            o[plane + i * ld] = a * old2[i] + b * hu[i] + c * R2[i];
            o[2 * plane + i * ld] = a * old3[i] + b * hv[i] + c * R3[i];
        }
    }
}

} // namespace bdg_dev
