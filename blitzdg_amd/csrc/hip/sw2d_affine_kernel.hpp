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

// 1/x to ~1 ulp for well-scaled positive x (water depths): hardware estimate + two Newton steps.
// (The default fp64 division also rescales and fixes up denormals/infinities, which cannot
// occur for h > 0; results differ from IEEE division by at most 1 ulp.)
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// sqrt(x) for x >= 0 to ~1 ulp: rsq estimate + coupled Newton iterations on (g, h) = (sqrt x, 1/(2 sqrt x)).
// x is clamped at 1e-290 so that x = 0 (fluid at rest) gives 1e-145 instead of 0*inf.
__device__ __forceinline__ double fast_sqrt(double x) {
    x = fmax(x, 1e-290);
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    g = fma(fma(-g, g, x), h, g); // 0.5 ulp already: a second correction changes nothing (profiles/r02_rcp_rsq_accuracy.txt)
    return g;
}

// Load/store at a wave-uniform row pointer plus a 32-bit per-lane BYTE offset: lowers to the
// scalar-base + vector-offset addressing mode (no 64-bit vector address arithmetic per row).
template <typename T>
__device__ __forceinline__ T ld_row(const T* row, unsigned byteOff) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(row) + byteOff);
}
template <typename T>
__device__ __forceinline__ void st_row(T* row, unsigned byteOff, T v) {
    *reinterpret_cast<T*>(reinterpret_cast<char*>(row) + byteOff) = v;
}

// Source terms evaluated inside the unrolled kernel (PHYS = 1): the momentum sources of the reference's
// Python RHS (swhelpers/rhs.py:300-309; sw2d.py:140-141) and of its C++ sw2d driver
// (src/sw2d/main.cpp:461-478) in one form,
//   S2 = f hv - cd |u| u + slope g h sx,     S3 = -f hu + dragSign cd |u| v + slope g h sy
// (rhs.py: slope = -1 with sx, sy = zx, zy and dragSign = +1, its sign quirk; main.cpp: slope = +1 with
// Hx, Hy and dragSign = -1), added node by node: R_c[m] += S_c[m]. PHYS = 2 is the filtered RHS: the
// reference drivers filter the whole RHS, sources included, so the kernel runs with the PLAIN operators
// and multiplies the finished R_c by the filter (fmat), instead of the pre-filtered operators of PHYS = 0.
struct PhysParams {
    const double* sx;    // (Np, ld) planes or nullptr
    const double* sy;
    const double* fcor;  // (Np, ld) Coriolis parameter or nullptr -> fconst
    const double* fmat;  // [m][i] = F[i][m] or nullptr
    double fconst, cd, slope, dragSign;
    // variant B on the matrix-core kernel (PHYS = 2 there): depth planes, open-boundary bit masks, the global
    // Lax-Friedrichs speed (device scalar), the tide elevation of this evaluation, the sponge field
    const double* H;
    const int* obc;
    const double* lam;
    const double* spongeField;
    double tide;
};

// TRACER: the passive tracer hN (field 3; F4 = hN u, G4 = hN v, no sources) in the same pass.
// SPONGE (MODE_COMBINE only): the momentum relaxation of the SSP-RK2 + sponge scheme is applied after the update; a
// launch with StageParams::sponge == 0 takes the instance without it.
template <int N, int MODE, int PHYS = 0, bool TRACER = false, bool SPONGE = false>
__global__ __launch_bounds__(256) void sw2d_stage_affine_kernel(const StageParams p, const PhysParams ph) {
    using E = Elem<N>;
    constexpr int Np = E::Np, Nfp = E::Nfp, NFN = E::NFN;

    const unsigned nwg = gridDim.x, xcd = blockIdx.x % 8u, q8 = nwg / 8u, r8 = nwg % 8u;
    const unsigned tile = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + blockIdx.x / 8u;
    // 32-bit element index: every row pointer below is wave-uniform (SGPR base) and the lane
    // part is a 32-bit offset, so a load needs no 64-bit vector address arithmetic.
    const unsigned k = static_cast<unsigned>(p.kbegin) + tile * blockDim.x + threadIdx.x;
    if (k >= static_cast<unsigned>(p.kend)) return;
    const unsigned k8 = k * 8u, k4 = k * 4u; // Np*ld*8 < 2^32 is checked on the host

    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld;
    const double* __restrict__ ops = p.opsAffine; // wave-uniform reads -> scalar loads
    const double* __restrict__ qin = p.qin;

    // ---- issue the independent loads: gather indices, own state, element geometry
    int idx[NFN];
#pragma unroll
    for (int j = 0; j < NFN; ++j) idx[j] = ld_row(p.vmapP + j * ld, k4);
    double h[Np], hu[Np], hv[Np], hN[TRACER ? Np : 1];
#pragma unroll
    for (int n = 0; n < Np; ++n) {
        h[n] = ld_row(qin + n * ld, k8);
        hu[n] = ld_row(qin + plane + n * ld, k8);
        hv[n] = ld_row(qin + 2 * plane + n * ld, k8);
        if constexpr (TRACER) hN[n] = ld_row(qin + 3 * plane + n * ld, k8);
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
    // source planes ride in the same batch and are consumed right away (they seed R2, R3 below)
    double ssx[PHYS != 0 ? Np : 1], ssy[PHYS != 0 ? Np : 1], sfc[PHYS != 0 ? Np : 1];
    if constexpr (PHYS != 0) {
        // (a table that is absent is a constant: decided once per TABLE, not per node -- 3 Np little branches in the
        // unrolled body cost the scheduler its one big region, see the update at the end)
#pragma unroll
        for (int i = 0; i < Np; ++i) { ssx[i] = 0.0; ssy[i] = 0.0; sfc[i] = ph.fconst; }
        if (ph.sx) {
#pragma unroll
            for (int i = 0; i < Np; ++i) ssx[i] = ld_row(ph.sx + i * ld, k8);
        }
        if (ph.sy) {
#pragma unroll
            for (int i = 0; i < Np; ++i) ssy[i] = ld_row(ph.sy + i * ld, k8);
        }
        if (ph.fcor) {
#pragma unroll
            for (int i = 0; i < Np; ++i) sfc[i] = ld_row(ph.fcor + i * ld, k8);
        }
    }
    // Keep the loads above in one batch: without this the scheduler sinks each load next to
    // its first use to save registers and the wave pays one memory round trip per node.
    __builtin_amdgcn_sched_barrier(0);
    // ---- neighbour ('+') traces: gathers from the same planes, served by L1/L2
    double hP[NFN], huP[NFN], hvP[NFN], hNP[TRACER ? NFN : 1];
    auto gatherFace = [&](int f) {
#pragma unroll
        for (int n = 0; n < Nfp; ++n) {
            const int j = f * Nfp + n;
            const unsigned o8 = static_cast<unsigned>(idx[j] < 0 ? -(idx[j] + 1) : idx[j]) * 8u;
            hP[j] = ld_row(qin, o8);
            huP[j] = ld_row(qin + plane, o8);
            hvP[j] = ld_row(qin + 2 * plane, o8);
            if constexpr (TRACER) hNP[j] = ld_row(qin + 3 * plane, o8);
        }
    };
    // Three fields: all three faces' traces in one batch. With the tracer (4 x 3 Nfp traces beside 4 Np state values and 4 Np
    // accumulators) the third face is requested when the first has been worked on: 28-55 spilled registers otherwise.
    // (the LSERK form keeps the single batch and its residual rows in the early batch: with the late requests it came out
    // at 40-60 spilled registers instead of 28, and 5 % slower)
    constexpr bool kLateFace = TRACER && MODE != MODE_LSERK;
    if constexpr (kLateFace) {
        gatherFace(0);
        gatherFace(1);
    } else { // (the loop as it always was: the register allocation of these kernels is sensitive to the very order of requests)
#pragma unroll
        for (int j = 0; j < NFN; ++j) {
            const unsigned o8 = static_cast<unsigned>(idx[j] < 0 ? -(idx[j] + 1) : idx[j]) * 8u;
            hP[j] = ld_row(qin, o8);
            huP[j] = ld_row(qin + plane, o8);
            hvP[j] = ld_row(qin + 2 * plane, o8);
            if constexpr (TRACER) hNP[j] = ld_row(qin + 3 * plane, o8);
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    const double g = p.g, halfg = 0.5 * p.g;
    double R1[Np], R2[Np], R3[Np], R4[TRACER ? Np : 1];
#pragma unroll
    for (int i = 0; i < Np; ++i) R1[i] = R2[i] = R3[i] = 0.0;
    if constexpr (TRACER) {
#pragma unroll
        for (int i = 0; i < Np; ++i) R4[i] = 0.0;
    }
    if constexpr (PHYS != 0) { // momentum sources, node by node
#pragma unroll
        for (int m = 0; m < Np; ++m) {
            const double r = fast_rcp(h[m]);
            const double u = hu[m] * r, v = hv[m] * r;
            const double cdn = ph.cd * fast_sqrt(u * u + v * v);
            const double gh = ph.slope * g * h[m];
            R2[m] = fma(gh, ssx[m], fma(sfc[m], hv[m], -(cdn * u)));
            R3[m] = fma(gh, ssy[m], fma(ph.dragSign * cdn, v, -(sfc[m] * hu[m])));
        }
    }

    // ---- surface term, face by face: R_c[i] += Lift[i][j] * s_c[j]
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
            double s4 = 0.0;
            if constexpr (TRACER) // the tracer's '+' trace at a wall is the element's own value
                s4 = half_fs * ((hN[m] * uM[n] - hNP[j] * uP[n]) * nxf + (hN[m] * vM[n] - hNP[j] * vP[n]) * nyf -
                                lam * (hN[m] - hNP[j]));
#pragma unroll
            for (int i = 0; i < Np; ++i) {
                const double lj = ops[AffineOps<N>::OFF_LIFT + j * Np + i];
                R1[i] = fma(lj, s1, R1[i]);
                R2[i] = fma(lj, s2, R2[i]);
                R3[i] = fma(lj, s3, R3[i]);
                if constexpr (TRACER) R4[i] = fma(lj, s4, R4[i]);
            }
        }
        if constexpr (kLateFace) {
            if (f == 0) {
                gatherFace(2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- stage inputs that are only needed at the very end: issue now, land during the volume loop
    double old1[Np], old2[Np], old3[Np], old4[TRACER ? Np : 1];
    if constexpr (MODE == MODE_LSERK) {
        const double* __restrict__ rs = p.res;
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            old1[i] = ld_row(rs + i * ld, k8);
            old2[i] = ld_row(rs + plane + i * ld, k8);
            old3[i] = ld_row(rs + 2 * plane + i * ld, k8);
            if constexpr (TRACER) old4[i] = ld_row(rs + 3 * plane + i * ld, k8);
        }
    } else if constexpr (MODE == MODE_COMBINE) {
        const double* __restrict__ qb = p.qbase;
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            old1[i] = ld_row(qb + i * ld, k8);
            old2[i] = ld_row(qb + plane + i * ld, k8);
            old3[i] = ld_row(qb + 2 * plane + i * ld, k8);
        }
    }
    // (combine steps: the tracer's base rows are requested after the volume loop)
    __builtin_amdgcn_sched_barrier(0);

    // ---- volume term, one input node at a time:
    //      R_c[i] -= Dr[i][m] (rx F_c + ry G_c)[m] + Ds[i][m] (sx F_c + sy G_c)[m]
#pragma unroll
    for (int m = 0; m < Np; ++m) {
        // The reciprocal is recomputed here (from h * 1.0 with a run-time 1.0, so the compiler
        // cannot reuse the surface term's value): keeping Np reciprocals live across the
        // surface term costs more in register traffic than Np short reciprocal sequences.
        const double r = fast_rcp(h[m] * p.one);
        const double u = hu[m] * r, v = hv[m] * r;
        const double pr = halfg * h[m] * h[m];
        const double F2 = hu[m] * u + pr, G2 = hu[m] * v, G3 = hv[m] * v + pr;
        const double a1 = -(rx * hu[m] + ry * hv[m]), b1 = -(sx * hu[m] + sy * hv[m]);
        const double a2 = -(rx * F2 + ry * G2), b2 = -(sx * F2 + sy * G2);
        const double a3 = -(rx * G2 + ry * G3), b3 = -(sx * G2 + sy * G3);
        double a4 = 0.0, b4 = 0.0;
        if constexpr (TRACER) {
            const double F4 = hN[m] * u, G4 = hN[m] * v;
            a4 = -(rx * F4 + ry * G4);
            b4 = -(sx * F4 + sy * G4);
        }
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            const double dr = ops[AffineOps<N>::OFF_D + 2 * (m * Np + i)];
            R1[i] = fma(dr, a1, R1[i]);
            R2[i] = fma(dr, a2, R2[i]);
            R3[i] = fma(dr, a3, R3[i]);
            if constexpr (TRACER) R4[i] = fma(dr, a4, R4[i]);
        }
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            const double ds = ops[AffineOps<N>::OFF_D + 2 * (m * Np + i) + 1];
            R1[i] = fma(ds, b1, R1[i]);
            R2[i] = fma(ds, b2, R2[i]);
            R3[i] = fma(ds, b3, R3[i]);
            if constexpr (TRACER) R4[i] = fma(ds, b4, R4[i]);
        }
    }

    // ---- PHYS == 2: the filtered RHS with sources is Filter * (flux terms + sources); the operators are
    //      then the plain ones and the filter is applied here, one field at a time
    if constexpr (PHYS == 2) {
        auto applyFilter = [&](double (&R)[Np]) {
            double T[Np];
#pragma unroll
            for (int i = 0; i < Np; ++i) T[i] = 0.0;
#pragma unroll
            for (int m = 0; m < Np; ++m)
#pragma unroll
                for (int i = 0; i < Np; ++i) T[i] = fma(ph.fmat[m * Np + i], R[m], T[i]);
#pragma unroll
            for (int i = 0; i < Np; ++i) R[i] = T[i];
        };
        applyFilter(R1);
        applyFilter(R2);
        applyFilter(R3);
        if constexpr (TRACER) applyFilter(R4);
    }

    if constexpr (TRACER && MODE == MODE_COMBINE) {
        const double* __restrict__ o4 = p.qbase + 3 * plane;
#pragma unroll
        for (int i = 0; i < Np; ++i) old4[i] = ld_row(o4 + i * ld, k8);
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- stage update / output
    if constexpr (MODE == MODE_RHS) {
        double* __restrict__ o = p.rhs;
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            st_row(o + i * ld, k8, R1[i]);
            st_row(o + plane + i * ld, k8, R2[i]);
            st_row(o + 2 * plane + i * ld, k8, R3[i]);
            if constexpr (TRACER) st_row(o + 3 * plane + i * ld, k8, R4[i]);
        }
    } else if constexpr (MODE == MODE_LSERK) {
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
            if constexpr (TRACER) {
                const double n4 = a * old4[i] + dt * R4[i];
                st_row(rs + 3 * plane + i * ld, k8, n4);
                st_row(o + 3 * plane + i * ld, k8, hN[i] + b * n4);
            }
        }
    } else {
        // The sponge (a division per momentum value) is a template parameter: with the test inside sponge_relax the unrolled
        // body became 2 Np little branches, the scheduler lost its one big region, the operator entries (scalar loads) were
        // all hoisted and spilled -- 1369 scalar and 283-666 vector registers, 1.56 ms per evaluation at N = 4 where the
        // LSERK form of the same kernel takes 0.35; and even one branch around two copies of the loop cost the tracer +
        // sources form 118 spilled vector registers.
        static_assert(!SPONGE || MODE == MODE_COMBINE, "the sponge follows a combine step");
        double* __restrict__ o = p.qout;
        const double a = p.ca, b = p.cb, c = p.cc;
        const double sg = p.sponge;
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            const double v2 = a * old2[i] + b * hu[i] + c * R2[i], v3 = a * old3[i] + b * hv[i] + c * R3[i];
            st_row(o + i * ld, k8, a * old1[i] + b * h[i] + c * R1[i]);
            st_row(o + plane + i * ld, k8, SPONGE ? v2 / (1.0 + sg * v2 * v2) : v2);
            st_row(o + 2 * plane + i * ld, k8, SPONGE ? v3 / (1.0 + sg * v3 * v3) : v3);
        }
        if constexpr (TRACER) {
#pragma unroll
            for (int i = 0; i < Np; ++i) st_row(o + 3 * plane + i * ld, k8, a * old4[i] + b * hN[i] + c * R4[i]);
        }
    }
}

} // namespace bdg_dev

namespace bdg_dev {

// ---------------------------------------------------------------------------------------------
// Streamed variant: the 3*Np accumulators are the only long-lived registers; the state is read
// three times (volume term, face traces, stage update) with the 2nd and 3rd read served by L2.
// The register budget (WAVES per SIMD as launch bound) lets two or three waves share a SIMD: one
// wave alone can issue a vector instruction only every other slot, and the partner wave also
// hides the per-face gather latency.
template <int N, int MODE, int WAVES>
__global__ __launch_bounds__(256, WAVES) void sw2d_stage_affine_stream_kernel(const StageParams p) {
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

    double R1[Np], R2[Np], R3[Np];
#pragma unroll
    for (int i = 0; i < Np; ++i) R1[i] = R2[i] = R3[i] = 0.0;

    // ---- volume term: stream the own state in chunks of VCH nodes, next chunk in flight
    {
        const double rx = ld_row(ag, k8), sx = ld_row(ag + ld, k8), ry = ld_row(ag + 2 * ld, k8),
                     sy = ld_row(ag + 3 * ld, k8);
        constexpr int VCH = 5, NCH = (Np + VCH - 1) / VCH;
        double hc[2][VCH], huc[2][VCH], hvc[2][VCH];
#pragma unroll
        for (int t = 0; t < VCH; ++t)
            if (t < Np) {
                hc[0][t] = ld_row(qin + t * ld, k8);
                huc[0][t] = ld_row(qin + plane + t * ld, k8);
                hvc[0][t] = ld_row(qin + 2 * plane + t * ld, k8);
            }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int cur = c & 1, nxt = cur ^ 1;
#pragma unroll
            for (int t = 0; t < VCH; ++t) {
                const int m = (c + 1) * VCH + t;
                if (m < Np) {
                    hc[nxt][t] = ld_row(qin + m * ld, k8);
                    huc[nxt][t] = ld_row(qin + plane + m * ld, k8);
                    hvc[nxt][t] = ld_row(qin + 2 * plane + m * ld, k8);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < VCH; ++t) {
                const int m = c * VCH + t;
                if (m < Np) {
                    const double h = hc[cur][t], hu = huc[cur][t], hv = hvc[cur][t];
                    const double r = fast_rcp(h);
                    const double u = hu * r, v = hv * r;
                    const double pr = halfg * h * h;
                    const double F2 = hu * u + pr, G2 = hu * v, G3 = hv * v + pr;
                    const double a1 = -(rx * hu + ry * hv), b1 = -(sx * hu + sy * hv);
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
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- surface term, face by face; '-' and '+' traces are (re)read per face
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        const double nxf = ld_row(ag + (4 + f) * ld, k8), nyf = ld_row(ag + (7 + f) * ld, k8);
        const double half_fs = 0.5 * ld_row(ag + (10 + f) * ld, k8);
        int idx[Nfp];
        double hM[Nfp], huM[Nfp], hvM[Nfp], hq[Nfp], huq[Nfp], hvq[Nfp];
#pragma unroll
        for (int n = 0; n < Nfp; ++n) idx[n] = ld_row(p.vmapP + (f * Nfp + n) * ld, k4);
#pragma unroll
        for (int n = 0; n < Nfp; ++n) {
            const int m = E::fmask(f, n);
            hM[n] = ld_row(qin + m * ld, k8);
            huM[n] = ld_row(qin + plane + m * ld, k8);
            hvM[n] = ld_row(qin + 2 * plane + m * ld, k8);
        }
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
            if (idx[n] < 0) { // reflective wall: no normal flow
                const double un = huM[n] * nxf + hvM[n] * nyf;
                huq[n] = huM[n] - 2 * nxf * un;
                hvq[n] = hvM[n] - 2 * nyf * un;
            }
            const double rM = fast_rcp(hM[n]), rP = fast_rcp(hq[n]);
            const double uM = huM[n] * rM, vM = hvM[n] * rM, uP = huq[n] * rP, vP = hvq[n] * rP;
            const double spdM = fast_sqrt(uM * uM + vM * vM) + fast_sqrt(g * hM[n]);
            const double spdP = fast_sqrt(uP * uP + vP * vP) + fast_sqrt(g * hq[n]);
            lam = fmax(lam, fmax(spdM, spdP));
        }
#pragma unroll
        for (int n = 0; n < Nfp; ++n) {
            const int j = f * Nfp + n;
            // velocities are recomputed rather than kept across the two passes (registers)
            const double rM = fast_rcp(hM[n] * p.one), rP = fast_rcp(hq[n] * p.one);
            const double uM = huM[n] * rM, vM = hvM[n] * rM, uP = huq[n] * rP, vP = hvq[n] * rP;
            const double prM = halfg * hM[n] * hM[n], prP = halfg * hq[n] * hq[n];
            const double F2M = huM[n] * uM + prM, G2M = huM[n] * vM, G3M = hvM[n] * vM + prM;
            const double F2P = huq[n] * uP + prP, G2P = huq[n] * vP, G3P = hvq[n] * vP + prP;
            const double dh = hM[n] - hq[n], dhu = huM[n] - huq[n], dhv = hvM[n] - hvq[n];
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

    // ---- stage update / output, a few nodes at a time
    if constexpr (MODE == MODE_RHS) {
        double* __restrict__ o = p.rhs;
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            st_row(o + i * ld, k8, R1[i]);
            st_row(o + plane + i * ld, k8, R2[i]);
            st_row(o + 2 * plane + i * ld, k8, R3[i]);
        }
    } else {
        constexpr int CH = 5; // nodes per batch of loads
        const double* __restrict__ base2 = (MODE == MODE_LSERK) ? p.res : p.qbase;
        double* __restrict__ o = p.qout;
        const double a = p.ca, b = p.cb, c = p.cc;
#pragma unroll
        for (int i0 = 0; i0 < Np; i0 += CH) {
            double q1[CH], q2[CH], q3[CH], o1[CH], o2[CH], o3[CH];
#pragma unroll
            for (int t = 0; t < CH; ++t) {
                const int i = i0 + t;
                if (i < Np) {
                    q1[t] = ld_row(qin + i * ld, k8);
                    q2[t] = ld_row(qin + plane + i * ld, k8);
                    q3[t] = ld_row(qin + 2 * plane + i * ld, k8);
                    o1[t] = ld_row(base2 + i * ld, k8);
                    o2[t] = ld_row(base2 + plane + i * ld, k8);
                    o3[t] = ld_row(base2 + 2 * plane + i * ld, k8);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < CH; ++t) {
                const int i = i0 + t;
                if (i < Np) {
                    if constexpr (MODE == MODE_LSERK) {
                        const double n1 = a * o1[t] + c * R1[i], n2 = a * o2[t] + c * R2[i], n3 = a * o3[t] + c * R3[i];
                        st_row(p.res + i * ld, k8, n1);
                        st_row(p.res + plane + i * ld, k8, n2);
                        st_row(p.res + 2 * plane + i * ld, k8, n3);
                        st_row(o + i * ld, k8, q1[t] + b * n1);
                        st_row(o + plane + i * ld, k8, q2[t] + b * n2);
                        st_row(o + 2 * plane + i * ld, k8, q3[t] + b * n3);
                    } else {
                        st_row(o + i * ld, k8, a * o1[t] + b * q1[t] + c * R1[i]);
                        st_row(o + plane + i * ld, k8, sponge_relax_no_test(a * o2[t] + b * q2[t] + c * R2[i], p.sponge));
                        st_row(o + 2 * plane + i * ld, k8, sponge_relax_no_test(a * o3[t] + b * q3[t] + c * R3[i], p.sponge));
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

} // namespace bdg_dev

namespace bdg_dev {

// ---------------------------------------------------------------------------------------------
// Rolled variant for higher orders. The fully unrolled kernels above put the whole operator
// (2*Np^2 + 3*Nfp*Np scalar operands) into one basic block; beyond N=5 the compiler then spills
// thousands of SGPRs/VGPRs. Here the loops over the INPUT node (volume) and over the face nodes
// (surface) are real loops whose body is one operator row applied to the accumulators, so code
// size and register use grow with Np, not Np^2, and 2+ waves share a SIMD.
//   FIELDS = 3: one wavefront lane updates all three fields of its element (3*Np accumulators);
//   FIELDS = 1: a workgroup is three wavefronts over the SAME 64 elements, wave c accumulating
//               field c only (Np accumulators) -- needed from N=7 on, where 3*Np doubles no longer
//               fit the 256 architectural VGPRs a vector FMA can address. The pointwise physics is
//               evaluated by all three waves (O(Np) against the O(Np^2) contractions) and their
//               redundant loads of the element state hit L1/L2.
template <int N>
__device__ __forceinline__ int fmask_rt(int f, int n) {
    const int rs = n * (N + 1) - (n * (n - 1)) / 2;
    return f == 0 ? n : (f == 1 ? rs + (N - n) : rs);
}

template <int N, int MODE, int FIELDS>
__global__ __launch_bounds__(FIELDS == 3 ? 256 : 192, 2) void sw2d_stage_affine_rolled_kernel(const StageParams p) {
    using E = Elem<N>;
    constexpr int Np = E::Np, Nfp = E::Nfp;
    static_assert(FIELDS == 1 || FIELDS == 3, "FIELDS is 1 or 3");

    const unsigned nwg = gridDim.x, xcd = blockIdx.x % 8u, q8 = nwg / 8u, r8 = nwg % 8u;
    const unsigned tile = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + blockIdx.x / 8u;
    const int c = FIELDS == 3 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // field of this wave
    const unsigned k = static_cast<unsigned>(p.kbegin) +
                       (FIELDS == 3 ? tile * 256u + threadIdx.x : tile * 64u + (threadIdx.x & 63u));
    if (k >= static_cast<unsigned>(p.kend)) return;
    const unsigned k8 = k * 8u, k4 = k * 4u;

    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld;
    const double* __restrict__ ops = p.opsAffine;
    const double* __restrict__ qin = p.qin;
    const double* __restrict__ ag = p.ageo;
    const double g = p.g, halfg = 0.5 * p.g;

    double R[FIELDS][Np];
#pragma unroll
    for (int q = 0; q < FIELDS; ++q)
#pragma unroll
        for (int i = 0; i < Np; ++i) R[q][i] = 0.0;

    // ---- volume term: one input node per iteration, next node's state in flight
    {
        const double rx = ld_row(ag, k8), sx = ld_row(ag + ld, k8), ry = ld_row(ag + 2 * ld, k8),
                     sy = ld_row(ag + 3 * ld, k8);
        double h = ld_row(qin, k8), hu = ld_row(qin + plane, k8), hv = ld_row(qin + 2 * plane, k8);
#pragma unroll 1
        for (int m = 0; m < Np; ++m) {
            const int mn = m + 1 < Np ? m + 1 : m;
            const double hn = ld_row(qin + mn * ld, k8), hun = ld_row(qin + plane + mn * ld, k8),
                         hvn = ld_row(qin + 2 * plane + mn * ld, k8);
            const double r = fast_rcp(h);
            const double u = hu * r, v = hv * r;
            const double pr = halfg * h * h;
            const double F2 = hu * u + pr, G2 = hu * v, G3 = hv * v + pr;
            const double* __restrict__ row = ops + AffineOps<N>::OFF_D + 2 * m * Np;
            if constexpr (FIELDS == 3) {
                const double a1 = -(rx * hu + ry * hv), b1 = -(sx * hu + sy * hv);
                const double a2 = -(rx * F2 + ry * G2), b2 = -(sx * F2 + sy * G2);
                const double a3 = -(rx * G2 + ry * G3), b3 = -(sx * G2 + sy * G3);
#pragma unroll
                for (int i = 0; i < Np; ++i) {
                    const double dr = row[2 * i], ds = row[2 * i + 1];
                    R[0][i] = fma(ds, b1, fma(dr, a1, R[0][i]));
                    R[FIELDS - 2][i] = fma(ds, b2, fma(dr, a2, R[FIELDS - 2][i]));
                    R[FIELDS - 1][i] = fma(ds, b3, fma(dr, a3, R[FIELDS - 1][i]));
                }
            } else {
                const double F = c == 0 ? hu : (c == 1 ? F2 : G2);
                const double G = c == 0 ? hv : (c == 1 ? G2 : G3);
                const double a = -(rx * F + ry * G), b = -(sx * F + sy * G);
#pragma unroll
                for (int i = 0; i < Np; ++i) R[0][i] = fma(row[2 * i + 1], b, fma(row[2 * i], a, R[0][i]));
            }
            h = hn; hu = hun; hv = hvn;
        }
    }

    // ---- surface term: per face, pass 1 finds the Lax-Friedrichs speed, pass 2 lifts the jumps
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
                    const double prM = halfg * hM * hM, prP = halfg * hq * hq;
                    const double F2M = huM * uM + prM, G2M = huM * vM, G3M = hvM * vM + prM;
                    const double F2P = huq * uP + prP, G2P = huq * vP, G3P = hvq * vP + prP;
                    const double dh = hM - hq, dhu = huM - huq, dhv = hvM - hvq;
                    const double* __restrict__ row = ops + AffineOps<N>::OFF_LIFT + j * Np;
                    if constexpr (FIELDS == 3) {
                        const double s1 = half_fs * (dhu * nxf + dhv * nyf - lam * dh);
                        const double s2 = half_fs * ((F2M - F2P) * nxf + (G2M - G2P) * nyf - lam * dhu);
                        const double s3 = half_fs * ((G2M - G2P) * nxf + (G3M - G3P) * nyf - lam * dhv);
#pragma unroll
                        for (int i = 0; i < Np; ++i) {
                            const double lj = row[i];
                            R[0][i] = fma(lj, s1, R[0][i]);
                            R[FIELDS - 2][i] = fma(lj, s2, R[FIELDS - 2][i]);
                            R[FIELDS - 1][i] = fma(lj, s3, R[FIELDS - 1][i]);
                        }
                    } else {
                        const double dF = c == 0 ? dhu : (c == 1 ? F2M - F2P : G2M - G2P);
                        const double dG = c == 0 ? dhv : (c == 1 ? G2M - G2P : G3M - G3P);
                        const double dq = c == 0 ? dh : (c == 1 ? dhu : dhv);
                        const double s = half_fs * (dF * nxf + dG * nyf - lam * dq);
#pragma unroll
                        for (int i = 0; i < Np; ++i) R[0][i] = fma(row[i], s, R[0][i]);
                    }
                }
            }
        }
    }

    // ---- stage update / output (register indices must be static: unrolled, loads in batches)
#pragma unroll
    for (int q = 0; q < FIELDS; ++q) {
        const long long fo = static_cast<long long>(FIELDS == 3 ? q : c) * plane;
        if constexpr (MODE == MODE_RHS) {
#pragma unroll
            for (int i = 0; i < Np; ++i) st_row(p.rhs + fo + i * ld, k8, R[q][i]);
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
                            const double n1 = a * o1[t] + cc * R[q][i];
                            st_row(p.res + fo + i * ld, k8, n1);
                            st_row(p.qout + fo + i * ld, k8, q1[t] + b * n1);
                        } else {
                            const double val = a * o1[t] + b * q1[t] + cc * R[q][i];
                            const bool momentum = (FIELDS == 3 ? q : c) == 1 || (FIELDS == 3 ? q : c) == 2;
                            st_row(p.qout + fo + i * ld, k8, momentum ? sponge_relax(val, p.sponge) : val);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

} // namespace bdg_dev
