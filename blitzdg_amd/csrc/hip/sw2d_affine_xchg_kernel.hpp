// sw2d_affine_xchg_kernel.hpp -- A/B variant 9 of the straight-element stage kernel (round 4): the neighbour traces of faces whose
// two elements sit in the SAME wavefront are exchanged inside the wave instead of gathered from memory.
//
// north_star asks for "face-trace gathers done with wavefront shuffles where both elements live in the same block". With one lane
// per element a shuffle cannot do it -- which of the neighbour's Np nodes a face node pairs with differs from lane to lane, and a
// shuffle reads ONE register of another lane -- so the exchange goes through LDS: every lane writes its 3 Np state values to a
// wave-private tile (23 KB at N = 4, one wave per workgroup), and a face node whose neighbour element lies in this wave's 64
// elements reads (node n', lane k' - wave base) back from it. The other lanes gather from memory as before, through buffer loads
// whose offset is pushed out of range for the in-wave lanes (the hardware drops those lanes' accesses: no branch, no traffic).
// On the generator's row-major mesh two of an element's three neighbours are within +-2 element numbers, so about two thirds
// of the gathers are served from LDS.
// Arithmetic, operator image and results are those of variant 0 (bit for bit); LSERK stages; BDG_SW2D_AFFINE_VARIANT=9.
#pragma once
#include "sw2d_affine_kernel.hpp"
#include "sw2d_mfma3_kernel.hpp" // plane_rsrc, bld_f64

namespace bdg_dev {

template <int N>
__global__ __launch_bounds__(64) void sw2d_stage_affine_xchg_kernel(const StageParams p, unsigned ldMagic) {
    using E = Elem<N>;
    constexpr int Np = E::Np, Nfp = E::Nfp, NFN = E::NFN;
    __shared__ double xs[3 * Np * 64];

    const unsigned nwg = gridDim.x, xcd = blockIdx.x % 8u, q8 = nwg / 8u, r8 = nwg % 8u;
    const unsigned tile = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + blockIdx.x / 8u;
    const unsigned base = static_cast<unsigned>(p.kbegin) + tile * 64u, lane = threadIdx.x;
    // (lanes beyond kend leave, as in variant 0: no live lane has a neighbour there; a branch around the update instead cost
    // 1371 spilled scalar and 106 vector registers -- the unrolled body must stay one basic block)
    const unsigned k = base + lane;
    if (k >= static_cast<unsigned>(p.kend)) return;
    const unsigned k8 = k * 8u, k4 = k * 4u;

    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld;
    const unsigned ldu = static_cast<unsigned>(ld), planeBytes = static_cast<unsigned>(plane * 8);
    const double* __restrict__ ops = p.opsAffine;
    const double* __restrict__ qin = p.qin;
    const __amdgpu_buffer_rsrc_t rq = plane_rsrc(qin, 3u * planeBytes);

    // ---- first batch: gather indices, own state, element geometry
    int idx[NFN];
#pragma unroll
    for (int j = 0; j < NFN; ++j) idx[j] = ld_row(p.vmapP + j * ld, k4);
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
    __builtin_amdgcn_sched_barrier(0);

    // ---- neighbour traces: where the neighbour element is one of this wave's 64, from the LDS tile; else from memory
    double hP[NFN], huP[NFN], hvP[NFN];
    unsigned ldsAt[NFN]; // (node n', lane) of the neighbour's value in the tile, or 0xffffffff: not in this wave
#pragma unroll
    for (int j = 0; j < NFN; ++j) {
        const unsigned o = static_cast<unsigned>(idx[j] < 0 ? -(idx[j] + 1) : idx[j]);   // n' ld + k'
        unsigned nn = __umulhi(o, ldMagic);                                                // o / ld, at most one too large
        nn -= (nn * ldu > o) ? 1u : 0u;
        const unsigned kl = o - nn * ldu - base;
        const bool inwave = kl < 64u;
        ldsAt[j] = inwave ? nn * 64u + kl : 0xffffffffu;
        const unsigned o8 = inwave ? 0xfffffff8u : o * 8u;                                 // out of range: the access is dropped
        hP[j] = bld_f64(rq, o8, 0u);
        huP[j] = bld_f64(rq, o8, planeBytes);
        hvP[j] = bld_f64(rq, o8, 2u * planeBytes);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < Np; ++n) {
        xs[n * 64 + lane] = h[n];
        xs[(Np + n) * 64 + lane] = hu[n];
        xs[(2 * Np + n) * 64 + lane] = hv[n];
    }
    __builtin_amdgcn_wave_barrier(); // one wave per workgroup: its DS operations execute in order
#pragma unroll
    for (int j = 0; j < NFN; ++j) {
        const bool inwave = ldsAt[j] != 0xffffffffu;
        const unsigned at = inwave ? ldsAt[j] : lane;
        const double a = xs[at], b = xs[Np * 64 + at], c = xs[2 * Np * 64 + at];
        hP[j] = inwave ? a : hP[j];
        huP[j] = inwave ? b : huP[j];
        hvP[j] = inwave ? c : hvP[j];
    }
    __builtin_amdgcn_sched_barrier(0);

    const double g = p.g, halfg = 0.5 * p.g;
    double R1[Np], R2[Np], R3[Np];
#pragma unroll
    for (int i = 0; i < Np; ++i) R1[i] = R2[i] = R3[i] = 0.0;

    // ---- surface term, face by face (as variant 0)
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        const double nxf = fnx[f], nyf = fny[f];
        double lam = 0.0;
        double uM[Nfp], vM[Nfp], uP[Nfp], vP[Nfp], hq[Nfp], huq[Nfp], hvq[Nfp];
#pragma unroll
        for (int n = 0; n < Nfp; ++n) {
            const int j = f * Nfp + n, m = E::fmask(f, n);
            const double rM = fast_rcp(h[m]);
            uM[n] = hu[m] * rM;
            vM[n] = hv[m] * rM;
            const double spdM = fast_sqrt(uM[n] * uM[n] + vM[n] * vM[n]) + fast_sqrt(g * h[m]);
            hq[n] = hP[j];
            huq[n] = huP[j];
            hvq[n] = hvP[j];
            if (idx[j] < 0) { // reflective wall: no normal flow
                const double un = hu[m] * nxf + hv[m] * nyf;
                huq[n] = hu[m] - 2 * nxf * un;
                hvq[n] = hv[m] - 2 * nyf * un;
            }
            const double r = fast_rcp(hq[n]);
            uP[n] = huq[n] * r;
            vP[n] = hvq[n] * r;
            const double spdP = fast_sqrt(uP[n] * uP[n] + vP[n] * vP[n]) + fast_sqrt(g * hq[n]);
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

    // ---- residual rows: issue now, land during the volume loop
    double old1[Np], old2[Np], old3[Np];
    const double* __restrict__ rs0 = p.res;
#pragma unroll
    for (int i = 0; i < Np; ++i) {
        old1[i] = ld_row(rs0 + i * ld, k8);
        old2[i] = ld_row(rs0 + plane + i * ld, k8);
        old3[i] = ld_row(rs0 + 2 * plane + i * ld, k8);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- volume term (as variant 0)
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

    // ---- LSERK4 stage update
    {
        double* __restrict__ rs = p.res;
        double* __restrict__ o = p.qout;
        const double a = p.ca, b = p.cb, dt = p.cc;
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            const double n1 = a * old1[i] + dt * R1[i];
            const double n2 = a * old2[i] + dt * R2[i];
            const double n3 = a * old3[i] + dt * R3[i];
            st_row(rs + i * ld, k8, n1);
            st_row(rs + plane + i * ld, k8, n2);
            st_row(rs + 2 * plane + i * ld, k8, n3);
            st_row(o + i * ld, k8, h[i] + b * n1);
            st_row(o + plane + i * ld, k8, hu[i] + b * n2);
            st_row(o + 2 * plane + i * ld, k8, hv[i] + b * n3);
        }
    }
}

} // namespace bdg_dev
