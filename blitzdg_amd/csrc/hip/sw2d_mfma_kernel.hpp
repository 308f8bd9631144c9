// sw2d_mfma_kernel.hpp -- fused RHS + stage update with the operator applies on the matrix cores
// (v_mfma_f64_16x16x4_f64), straight-sided elements, any order.
//
// Same mathematics as sw2d_stage_affine_kernel (reference src/sw2d-simple/main.cpp:181-356 + stage
// update): per element  R_c = Dr' a_c + Ds' b_c + Lift' s_c  with the contravariant fluxes
// a_c = -(rx F_c + ry G_c), b_c = -(sx F_c + sy G_c) and the scaled Lax-Friedrichs jumps s_c.
// Here that is a batched GEMM: for a tile of 16 elements,
//     R_c (Np x 16)  =  [Dr' | Ds' | Lift'] (Np x (2Np + 3Nfp))  *  [a_c ; b_c ; s_c] ((2Np + 3Nfp) x 16)
// evaluated with 16x16x4 f64 MFMAs. A wavefront owns 16 elements; lane l = (q = l>>4, j = l&15)
// works on element j of the tile and on the nodes congruent to q mod 4. That is exactly the B-operand
// layout of the instruction (B[k = l>>4][col = l&15]), so the pointwise physics is computed directly
// in operand layout -- no cross-lane traffic, no redundancy -- and the C/D layout
// (col = l&15, row = (l>>4) + 4*reg) hands every lane the output nodes q, q+4, q+8, ... of its own
// element for the stage update. A operands (operator tiles, zero padded to 16 x 4) are staged once
// per workgroup in LDS in lane order and read with one conflict-free ds_read_b64 per MFMA.
// Memory accesses are 128-byte row segments (16 elements x 8 B), four rows per wave instruction.
// Only the per-face maximum of the wave speed crosses lanes (two xor-shuffles per face).
#pragma once
#include "sw2d_affine_kernel.hpp"

namespace bdg_dev {

// Operator image -> LDS at the start of a workgroup: every thread requests ALL its 16-byte pieces before it stores the
// first (one round trip to L2 instead of one per piece: a plain copy loop compiles to load, wait, store per iteration, 25-33
// round trips for the 50 KB image of N = 8 -- 5 us at the start of every launch, and of every partition-boundary strip).
typedef double bdg_f64x2 __attribute__((ext_vector_type(2)));
template <int COUNT, int THREADS>
__device__ __attribute__((noinline)) void stage_image(double* __restrict__ lds, const double* __restrict__ src) {
    static_assert(COUNT % 2 == 0, "operator images are whole 64-double tiles");
    constexpr int PAIRS = COUNT / 2, STEPS = (PAIRS + THREADS - 1) / THREADS;
    const bdg_f64x2* __restrict__ s2 = reinterpret_cast<const bdg_f64x2*>(src);
    bdg_f64x2* __restrict__ d2 = reinterpret_cast<bdg_f64x2*>(lds);
    bdg_f64x2 v[STEPS];
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        const int at = i * THREADS + static_cast<int>(threadIdx.x);
        if (at < PAIRS) v[i] = s2[at];
    }
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        const int at = i * THREADS + static_cast<int>(threadIdx.x);
        if (at < PAIRS) d2[at] = v[i];
    }
}


typedef double mfma_acc_t __attribute__((ext_vector_type(4)));

#ifndef BDG_MFMA_WAVES
#define BDG_MFMA_WAVES 1 // waves per SIMD the register budget is set for (2 spills at N >= 6)
#endif

template <int N>
struct MfmaOps {
    using E = Elem<N>;
    static constexpr int MT = (E::Np + 15) / 16;   // 16-row tiles of output nodes
    static constexpr int KV = (E::Np + 3) / 4;     // 4-deep steps over input nodes
    static constexpr int KS = (E::NFN + 3) / 4;    // 4-deep steps over face nodes
    // image: [matrix: Dr', Ds'][r][t][64] then [Lift'][r][t][64]; entry lane l = A[16r + (l&15)][4t + (l>>4)]
    static constexpr int OFF_DR = 0;
    static constexpr int OFF_DS = MT * KV * 64;
    static constexpr int OFF_LIFT = 2 * MT * KV * 64;
    static constexpr int DOUBLES = OFF_LIFT + MT * KS * 64;
};

// HALO (with MODE_LSERK, partition-boundary launches): pack and unpack of the ghost exchange folded in, see
// StageParams::haloRecv.
// SYNC (partitioned stages, StageParams::syncWait ...): an interior launch's tiles from syncFirstTile on -- the ring of
// elements next to the partition boundary -- wait for the previous stage's boundary launch before they read or write anything
// and signal when their stores are visible; a boundary launch (HALO) waits for the previous interior launch's ring tiles at
// its top and signals once per workgroup at its end.
// (N <= 4: the register budget of two waves per SIMD -- with every request ahead of the first product the kernel sits at 232-241
// vector registers beside its 24 accumulation registers, a few above the 256 that two waves may hold together)
// THREADS = 64 (boundary launches of N <= 4): one-wave workgroups. A 256-thread workgroup needs register room on all four SIMDs of a CU at
// the same moment; beside an interior launch on the unrolled kernel -- one-wave workgroups of 410 registers that the dispatcher replaces one by
// one -- that moment never comes before the interior grid is exhausted, and the boundary chain of a 2- or 4-way stage ran BEHIND the interior
// launch instead of beside it. A one-wave workgroup takes the first SIMD that falls free.
template <int N, int MODE, bool HALO = false, bool SYNC = false, int THREADS = 256>
__global__ __launch_bounds__(THREADS, (N <= 4 ? 2 : BDG_MFMA_WAVES)) void sw2d_stage_mfma_kernel(const StageParams p) {
    constexpr unsigned WAVES = THREADS / 64;
    using E = Elem<N>;
    using O = MfmaOps<N>;
    constexpr int Np = E::Np, Nfp = E::Nfp, NFN = E::NFN, MT = O::MT, KV = O::KV, KS = O::KS;

    extern __shared__ double sOps[];
    stage_image<O::DOUBLES, THREADS>(sOps, p.opsAffine);
    __syncthreads();

    const unsigned lane = threadIdx.x & 63u, q = lane >> 4, j = lane & 15u;
    // XCD-aware, contiguous chunks of tiles per wave (neighbouring tiles share an L2)
    const unsigned nwg = gridDim.x, xcd = blockIdx.x % 8u, q8 = nwg / 8u, r8 = nwg % 8u;
    const unsigned blk = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + blockIdx.x / 8u;
    const unsigned wave = blk * WAVES + (threadIdx.x >> 6), nwaves = nwg * WAVES;
    const unsigned ntiles = (static_cast<unsigned>(p.kend - p.kbegin) + 15u) / 16u;
    const unsigned perWave = (ntiles + nwaves - 1u) / nwaves;
    const unsigned tileEnd = min(ntiles, (wave + 1u) * perWave);

    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld;
    const double* __restrict__ qin = p.qin;
    const double* __restrict__ ag = p.ageo;
    const double g = p.g, halfg = 0.5 * p.g;
    const unsigned kLast = static_cast<unsigned>(p.kend) - 1u;

    bool synced = false;
    for (unsigned tile = wave * perWave; tile < tileEnd; ++tile) {
        if constexpr (SYNC) {
            if (!synced && tile >= static_cast<unsigned>(p.syncFirstTile)) {
                sync_wait(p.syncWait, p.syncWaitValue, p.syncError);
                synced = true;
            }
        }
        const unsigned kTrue = static_cast<unsigned>(p.kbegin) + tile * 16u + j;
        const bool live = kTrue <= kLast;
        const unsigned k = live ? kTrue : kLast; // padding lanes recompute the last element, store nothing
        const unsigned k8 = k * 8u, k4 = k * 4u;

        // Round 4: everything the tile reads is REQUESTED in dependency order before the first product -- indices, geometry, own
        // rows of the volume steps and of the face nodes; then the neighbour traces (they need the indices) while the volume
        // operands are formed; the update's rows (state again, residual) behind the volume products -- instead of where each value
        // is first used. A wave of this kernel runs ONE tile of a small launch (a rank's share of a many-way split, its boundary
        // strip): its time is the chain of dependent round trips, which this order cuts from four or five to two. No request sits
        // inside a lane-dependent branch (padding rows / face nodes beyond the element read a valid address and are ignored).
        // Same arithmetic in the same order as before: results are bit-identical.
        int id[KS];
        unsigned mface[KS];
        int fidx[KS];
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            const int jf = 4 * t + static_cast<int>(q), jc = jf < NFN ? jf : 0;
            fidx[t] = jc / Nfp;
            mface[t] = static_cast<unsigned>(fmask_rt<N>(fidx[t], jc - fidx[t] * Nfp));
            id[t] = ld_row(p.vmapP + jc * ld, k4);
        }
        const double rx = ld_row(ag, k8), sx = ld_row(ag + ld, k8), ry = ld_row(ag + 2 * ld, k8), sy = ld_row(ag + 3 * ld, k8);
        const double hf0 = 0.5 * ld_row(ag + 10 * ld, k8), hf1 = 0.5 * ld_row(ag + 11 * ld, k8), hf2 = 0.5 * ld_row(ag + 12 * ld, k8);
        double vh[KV], vhu[KV], vhv[KV];
#pragma unroll
        for (int t = 0; t < KV; ++t) {
            const int m = 4 * t + static_cast<int>(q), mc = m < Np ? m : 0;
            vh[t] = ld_row(qin + mc * ld, k8);
            vhu[t] = ld_row(qin + plane + mc * ld, k8);
            vhv[t] = ld_row(qin + 2 * plane + mc * ld, k8);
        }
        double nxv[KS], nyv[KS], hMv[KS], huMv[KS], hvMv[KS];
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            nxv[t] = ld_row(ag + (4 + fidx[t]) * ld, k8);
            nyv[t] = ld_row(ag + (7 + fidx[t]) * ld, k8);
            hMv[t] = ld_row(qin + static_cast<long long>(mface[t]) * ld, k8);
            huMv[t] = ld_row(qin + plane + static_cast<long long>(mface[t]) * ld, k8);
            hvMv[t] = ld_row(qin + 2 * plane + static_cast<long long>(mface[t]) * ld, k8);
        }
        int sendRec[3] = {-1, -1, -1};
        if constexpr (HALO) {
            const unsigned b3 = (k - static_cast<unsigned>(p.kbegin)) * 3u;
            sendRec[0] = p.haloSendOf[b3];
            sendRec[1] = p.haloSendOf[b3 + 1];
            sendRec[2] = p.haloSendOf[b3 + 2];
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- neighbour ('+') traces: from the state planes, or (HALO) from the received record of a ghost element
        double hqv[KS], huqv[KS], hvqv[KS];
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            const unsigned idp = static_cast<unsigned>(id[t] < 0 ? -(id[t] + 1) : id[t]);
            const double* b0 = qin;
            const double* b1 = qin + plane;
            const double* b2 = qin + 2 * plane;
            unsigned o8 = idp * 8u;
            if constexpr (HALO) {
                const unsigned row = idp / static_cast<unsigned>(ld), slot = idp - row * static_cast<unsigned>(ld);
                const bool ghost = slot >= static_cast<unsigned>(p.haloOwned);
                const unsigned rec8 = ((slot - static_cast<unsigned>(p.haloOwned)) * static_cast<unsigned>(p.haloRows) + row) * 8u;
                b0 = ghost ? p.haloRecv : b0;            // the neighbour's record as it arrived: [field][node]
                b1 = ghost ? p.haloRecv + Np : b1;
                b2 = ghost ? p.haloRecv + 2 * Np : b2;
                o8 = ghost ? rec8 : o8;
            }
            hqv[t] = ld_row(b0, o8);
            huqv[t] = ld_row(b1, o8);
            hvqv[t] = ld_row(b2, o8);
        }
        __builtin_amdgcn_sched_barrier(0);

        mfma_acc_t acc[3][MT];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < MT; ++r) acc[c][r] = mfma_acc_t{0.0, 0.0, 0.0, 0.0};

        // ---- volume term: k-steps of 4 input nodes, this lane supplies node m = 4t + q
#pragma unroll
        for (int t = 0; t < KV; ++t) {
            const int m = 4 * t + static_cast<int>(q);
            double a1 = 0, b1 = 0, a2 = 0, b2 = 0, a3 = 0, b3 = 0;
            {
                const bool pad = m >= Np;
                const double h = pad ? 1.0 : vh[t], hu = vhu[t], hv = vhv[t];
                const double r = fast_rcp(h);
                const double u = hu * r, v = hv * r;
                const double pr = halfg * h * h;
                const double F2 = hu * u + pr, G2 = hu * v, G3 = hv * v + pr;
                const double a1n = -(rx * hu + ry * hv), b1n = -(sx * hu + sy * hv);
                const double a2n = -(rx * F2 + ry * G2), b2n = -(sx * F2 + sy * G2);
                const double a3n = -(rx * G2 + ry * G3), b3n = -(sx * G2 + sy * G3);
                a1 = pad ? 0.0 : a1n; b1 = pad ? 0.0 : b1n;
                a2 = pad ? 0.0 : a2n; b2 = pad ? 0.0 : b2n;
                a3 = pad ? 0.0 : a3n; b3 = pad ? 0.0 : b3n;
            }
#pragma unroll
            for (int r = 0; r < MT; ++r) {
                const double Adr = sOps[O::OFF_DR + (r * KV + t) * 64 + lane];
                const double Ads = sOps[O::OFF_DS + (r * KV + t) * 64 + lane];
                acc[0][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(Adr, a1, acc[0][r], 0, 0, 0);
                acc[1][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(Adr, a2, acc[1][r], 0, 0, 0);
                acc[2][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(Adr, a3, acc[2][r], 0, 0, 0);
                acc[0][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ads, b1, acc[0][r], 0, 0, 0);
                acc[1][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ads, b2, acc[1][r], 0, 0, 0);
                acc[2][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ads, b3, acc[2][r], 0, 0, 0);
            }
        }

        // ---- the update's inputs: requested now, land during the surface term. The element's own state at this lane's OUTPUT nodes
        //      i = 16 r + q + 4 reg is what the volume steps already hold -- node m = 4 t + q with t = 4 r + reg -- so it is not read again.
        double oldv[3][MT][4];
        if constexpr (MODE != MODE_RHS) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const long long fo = static_cast<long long>(c) * plane;
                const double* __restrict__ base2 = ((MODE == MODE_LSERK) ? p.res : p.qbase) + fo;
#pragma unroll
                for (int r = 0; r < MT; ++r)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int i = 16 * r + static_cast<int>(q) + 4 * reg, ic = i < Np ? i : 0;
                        oldv[c][r][reg] = ld_row(base2 + ic * ld, k8);
                    }
            }
        }
        __builtin_amdgcn_sched_barrier(0);

        // ---- surface term: k-steps of 4 face nodes, this lane supplies face node jf = 4t + q
        {
            double e1[KS], e2[KS], e3[KS], d1[KS], d2[KS], d3[KS], spd[KS];
            double lamF[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int t = 0; t < KS; ++t) {
                const int jf = 4 * t + static_cast<int>(q);
                e1[t] = e2[t] = e3[t] = d1[t] = d2[t] = d3[t] = 0.0;
                spd[t] = 0.0;
                if (jf < NFN) {
                    const int f = fidx[t];
                    const double nxf = nxv[t], nyf = nyv[t];
                    const double hM = hMv[t], huM = huMv[t], hvM = hvMv[t];
                    const double hq = hqv[t];
                    double huq = huqv[t], hvq = hvqv[t];
                    if (id[t] < 0) { // reflective wall: no normal flow
                        const double un = huM * nxf + hvM * nyf;
                        huq = huM - 2 * nxf * un;
                        hvq = hvM - 2 * nyf * un;
                    }
                    const double rM = fast_rcp(hM), rP = fast_rcp(hq);
                    const double uM = huM * rM, vM = hvM * rM, uP = huq * rP, vP = hvq * rP;
                    const double spdM = fast_sqrt(uM * uM + vM * vM) + fast_sqrt(g * hM);
                    const double spdP = fast_sqrt(uP * uP + vP * vP) + fast_sqrt(g * hq);
                    spd[t] = fmax(spdM, spdP);
                    const double prM = halfg * hM * hM, prP = halfg * hq * hq;
                    const double F2M = huM * uM + prM, G2M = huM * vM, G3M = hvM * vM + prM;
                    const double F2P = huq * uP + prP, G2P = huq * vP, G3P = hvq * vP + prP;
                    d1[t] = hM - hq; d2[t] = huM - huq; d3[t] = hvM - hvq;
                    e1[t] = d2[t] * nxf + d3[t] * nyf;
                    e2[t] = (F2M - F2P) * nxf + (G2M - G2P) * nyf;
                    e3[t] = (G2M - G2P) * nxf + (G3M - G3P) * nyf;
                    lamF[0] = f == 0 ? fmax(lamF[0], spd[t]) : lamF[0];
                    lamF[1] = f == 1 ? fmax(lamF[1], spd[t]) : lamF[1];
                    lamF[2] = f == 2 ? fmax(lamF[2], spd[t]) : lamF[2];
                }
            }
            // per-face maximum over the face's nodes: they sit in the 4 lanes (q) of this element
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                lamF[f] = fmax(lamF[f], __shfl_xor(lamF[f], 16));
                lamF[f] = fmax(lamF[f], __shfl_xor(lamF[f], 32));
            }
#pragma unroll
            for (int t = 0; t < KS; ++t) {
                const int jf = 4 * t + static_cast<int>(q);
                const int f = jf / Nfp;
                const double lam = f == 0 ? lamF[0] : (f == 1 ? lamF[1] : lamF[2]);
                const double hfs = jf < NFN ? (f == 0 ? hf0 : (f == 1 ? hf1 : hf2)) : 0.0;
                const double s1 = hfs * (e1[t] - lam * d1[t]);
                const double s2 = hfs * (e2[t] - lam * d2[t]);
                const double s3 = hfs * (e3[t] - lam * d3[t]);
#pragma unroll
                for (int r = 0; r < MT; ++r) {
                    const double Al = sOps[O::OFF_LIFT + (r * KS + t) * 64 + lane];
                    acc[0][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(Al, s1, acc[0][r], 0, 0, 0);
                    acc[1][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(Al, s2, acc[1][r], 0, 0, 0);
                    acc[2][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(Al, s3, acc[2][r], 0, 0, 0);
                }
            }
        }

        // ---- stage update / output: this lane holds output nodes i = 16r + q + 4*reg of its element
        if (live) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const long long fo = static_cast<long long>(c) * plane;
#pragma unroll
                for (int r = 0; r < MT; ++r) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int i = 16 * r + static_cast<int>(q) + 4 * reg;
                        if (i < Np) {
                            const double R = acc[c][r][reg];
                            const int ts = 4 * r + reg;                                    // (i < Np implies ts < KV)
                            const double own = c == 0 ? vh[ts < KV ? ts : 0] : (c == 1 ? vhu[ts < KV ? ts : 0] : vhv[ts < KV ? ts : 0]);
                            if constexpr (MODE == MODE_RHS) {
                                st_row(p.rhs + fo + i * ld, k8, R);
                            } else if constexpr (MODE == MODE_LSERK) {
                                const double n1 = p.ca * oldv[c][r][reg] + p.cc * R;
                                const double qn = own + p.cb * n1;
                                st_row(p.res + fo + i * ld, k8, n1);
                                if constexpr (SYNC) { // a tile that signals hands its new state to the other chain: write-through
                                    if (HALO || tile >= static_cast<unsigned>(p.syncFirstTile)) st_row_wt(p.qout + fo + i * ld, k8, qn);
                                    else st_row(p.qout + fo + i * ld, k8, qn);
                                } else {
                                    st_row(p.qout + fo + i * ld, k8, qn);
                                }
                                if constexpr (HALO) {
#pragma unroll
                                    for (int sr = 0; sr < 3; ++sr)
                                        if (sendRec[sr] >= 0)
                                            p.haloSend[static_cast<size_t>(sendRec[sr]) * p.haloRows + c * Np + i] = qn;
                                }
                            } else {
                                const double val = p.ca * oldv[c][r][reg] + p.cb * own + p.cc * R;
                                st_row(p.qout + fo + i * ld, k8, c == 0 ? val : sponge_relax(val, p.sponge));
                            }
                        }
                    }
                }
            }
        }
        if constexpr (SYNC && !HALO) {
            if (tile >= static_cast<unsigned>(p.syncFirstTile)) sync_signal_wave(p.syncSignal);
        }
    }
    if constexpr (SYNC && HALO) sync_signal_workgroup(p.syncSignal);
}

} // namespace bdg_dev

namespace bdg_dev {

// ---------------------------------------------------------------------------------------------
// Second schedule of the same computation, organised for a 256-register budget (two waves per
// SIMD, so one wave's loads, shuffles and stores hide under the other's MFMAs):
//   * the surface term is processed face by face, each face padded to KF = ceil(Nfp/4) k-steps, so
//     the jump data of only one face is alive while its Lax-Friedrichs speed is reduced;
//   * the volume term runs in chunks of VC k-steps with the next chunk's state loads in flight;
//   * the stage update handles one field at a time.
#ifndef BDG_MFMA2_WAVES
#define BDG_MFMA2_WAVES 2
#endif

template <int N>
struct MfmaOps2 {
    using E = Elem<N>;
    static constexpr int MT = (E::Np + 15) / 16;
    static constexpr int KV = (E::Np + 3) / 4;
    static constexpr int KF = (E::Nfp + 3) / 4;    // k-steps per face
    static constexpr int OFF_DR = 0;
    static constexpr int OFF_DS = MT * KV * 64;
    static constexpr int OFF_LIFT = 2 * MT * KV * 64; // [r][f][tf][64], lane l = Lift'[16r + (l&15)][f*Nfp + 4tf + (l>>4)]
    static constexpr int DOUBLES = OFF_LIFT + MT * 3 * KF * 64;
};

// PHYS = 1: momentum sources of variants C/D (PhysParams, sw2d_affine_kernel.hpp) through the matrix cores as
// well: the operator image carries one more block of tiles, F' = Filter for the filtered RHS (the drivers
// filter the whole RHS, sources included) or the identity, and R_c += F' S_c joins the volume term's k-steps,
// where the lane already holds h, hu, hv of the node it needs.
// TRACER: a fourth accumulator set for the passive tracer hN (F4 = hN u, G4 = hN v) in the same pass -- where
// the register budget allows it (MT <= 2, i.e. N <= 6); above that the tracer runs as its own pass below.
// HALO (MODE_LSERK, PHYS = 0, partition-boundary launches): ghost traces read from the received records and the
// new state written to the send records, as in sw2d_stage_mfma_kernel.
template <int N, int MODE, int PHYS = 0, bool TRACER = false, bool HALO = false>
__global__ __launch_bounds__(256, BDG_MFMA2_WAVES) void sw2d_stage_mfma2_kernel(const StageParams p, const PhysParams ph) {
    constexpr int NFLD = TRACER ? 4 : 3;
    using E = Elem<N>;
    using O = MfmaOps2<N>;
    constexpr int Np = E::Np, Nfp = E::Nfp, MT = O::MT, KV = O::KV, KF = O::KF;
    constexpr int OFF_F = O::DOUBLES;                                  // [r][t][64] tiles of F'
    constexpr int IMAGE = O::DOUBLES + (PHYS != 0 ? MT * KV * 64 : 0);

    extern __shared__ double sOps[];
    stage_image<IMAGE, 256>(sOps, p.opsAffine);
    __syncthreads();

    const unsigned lane = threadIdx.x & 63u, q = lane >> 4, j = lane & 15u;
    const unsigned nwg = gridDim.x, xcd = blockIdx.x % 8u, q8 = nwg / 8u, r8 = nwg % 8u;
    const unsigned blk = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + blockIdx.x / 8u;
    const unsigned wave = blk * 4u + (threadIdx.x >> 6), nwaves = nwg * 4u;
    const unsigned ntiles = (static_cast<unsigned>(p.kend - p.kbegin) + 15u) / 16u;
    const unsigned perWave = (ntiles + nwaves - 1u) / nwaves;
    const unsigned tileEnd = min(ntiles, (wave + 1u) * perWave);

    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld;
    const double* __restrict__ qin = p.qin;
    const double* __restrict__ ag = p.ageo;
    const double g = p.g, halfg = 0.5 * p.g;
    const unsigned kLast = static_cast<unsigned>(p.kend) - 1u;

#pragma unroll 1
    for (unsigned tile = wave * perWave; tile < tileEnd; ++tile) {
        const unsigned kTrue = static_cast<unsigned>(p.kbegin) + tile * 16u + j;
        const bool live = kTrue <= kLast;
        const unsigned k = live ? kTrue : kLast;
        const unsigned k8 = k * 8u, k4 = k * 4u;

        mfma_acc_t acc[NFLD][MT];
#pragma unroll
        for (int c = 0; c < NFLD; ++c)
#pragma unroll
            for (int r = 0; r < MT; ++r) acc[c][r] = mfma_acc_t{0.0, 0.0, 0.0, 0.0};

        // gather indices of all three faces: issued before the volume term so that each face's dependent
        // trace gathers can start the moment its turn comes (N=8, 250 k elements: 0.418 -> 0.394 ms)
        int fidx[3][KF];
#pragma unroll
        for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int tf = 0; tf < KF; ++tf) {
                const int n = 4 * tf + static_cast<int>(q);
                fidx[f][tf] = n < Nfp ? ld_row(p.vmapP + (f * Nfp + n) * ld, k4) : 0;
            }

        int btags = 0;
        double lamGlobal = 0.0;
        if constexpr (PHYS == 2) {
            btags = ld_row(ph.obc, k4);
            lamGlobal = *ph.lam;
        }

        // ---- volume term in chunks of VC k-steps, next chunk's loads in flight
        {
            const double rx = ld_row(ag, k8), sx = ld_row(ag + ld, k8), ry = ld_row(ag + 2 * ld, k8),
                         sy = ld_row(ag + 3 * ld, k8);
            constexpr int VC = 3, NC = (KV + VC - 1) / VC;
            double hb[2][VC], hub[2][VC], hvb[2][VC];
            double sxb[2][PHYS != 0 ? VC : 1], syb[2][PHYS != 0 ? VC : 1], fcb[2][PHYS != 0 ? VC : 1];
            double hnb[2][TRACER ? VC : 1];
            auto loadChunk = [&](int ch, int buf) {
#pragma unroll
                for (int s = 0; s < VC; ++s) {
                    const int t = ch * VC + s, m = 4 * t + static_cast<int>(q);
                    hb[buf][s] = 1.0; hub[buf][s] = 0.0; hvb[buf][s] = 0.0;
                    if constexpr (PHYS != 0) { sxb[buf][s] = 0.0; syb[buf][s] = 0.0; fcb[buf][s] = ph.fconst; }
                    if constexpr (TRACER) hnb[buf][s] = 0.0;
                    if (t < KV && m < Np) {
                        if constexpr (TRACER) hnb[buf][s] = ld_row(qin + 3 * plane + m * ld, k8);
                        hb[buf][s] = ld_row(qin + m * ld, k8);
                        hub[buf][s] = ld_row(qin + plane + m * ld, k8);
                        hvb[buf][s] = ld_row(qin + 2 * plane + m * ld, k8);
                        if constexpr (PHYS != 0) {
                            if (ph.sx) sxb[buf][s] = ld_row(ph.sx + m * ld, k8);
                            if (ph.sy) syb[buf][s] = ld_row(ph.sy + m * ld, k8);
                            if (ph.fcor) fcb[buf][s] = ld_row(ph.fcor + m * ld, k8);
                        }
                    }
                }
            };
            loadChunk(0, 0);
#pragma unroll
            for (int ch = 0; ch < NC; ++ch) {
                const int cur = ch & 1;
                if (ch + 1 < NC) loadChunk(ch + 1, cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < VC; ++s) {
                    const int t = ch * VC + s;
                    if (t < KV) {
                        const int m = 4 * t + static_cast<int>(q);
                        const double h = hb[cur][s], hu = hub[cur][s], hv = hvb[cur][s];
                        const double r = fast_rcp(h);
                        const double u = hu * r, v = hv * r;
                        const double pr = halfg * h * h;
                        const double F2 = hu * u + pr, G2 = hu * v, G3 = hv * v + pr;
                        const double w = m < Np ? -1.0 : 0.0; // zero the padded rows of the operand
                        const double a1 = w * (rx * hu + ry * hv), b1 = w * (sx * hu + sy * hv);
                        const double a2 = w * (rx * F2 + ry * G2), b2 = w * (sx * F2 + sy * G2);
                        const double a3 = w * (rx * G2 + ry * G3), b3 = w * (sx * G2 + sy * G3);
                        if constexpr (TRACER) {
                            const double F4 = hnb[cur][s] * u, G4 = hnb[cur][s] * v;
                            const double a4 = w * (rx * F4 + ry * G4), b4 = w * (sx * F4 + sy * G4);
#pragma unroll
                            for (int r2 = 0; r2 < MT; ++r2) {
                                acc[3][r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(sOps[O::OFF_DR + (r2 * KV + t) * 64 + lane], a4, acc[3][r2], 0, 0, 0);
                                acc[3][r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(sOps[O::OFF_DS + (r2 * KV + t) * 64 + lane], b4, acc[3][r2], 0, 0, 0);
                            }
                        }
                        if constexpr (PHYS != 0) {
                            const double cdn = ph.cd * fast_sqrt(u * u + v * v), gh = ph.slope * g * h;
                            const double s2 = -w * fma(gh, sxb[cur][s], fma(fcb[cur][s], hv, -(cdn * u)));
                            const double s3 = -w * fma(gh, syb[cur][s], fma(ph.dragSign * cdn, v, -(fcb[cur][s] * hu)));
#pragma unroll
                            for (int r2 = 0; r2 < MT; ++r2) {
                                const double Af = sOps[OFF_F + (r2 * KV + t) * 64 + lane];
                                acc[1][r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Af, s2, acc[1][r2], 0, 0, 0);
                                acc[2][r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Af, s3, acc[2][r2], 0, 0, 0);
                            }
                        }
#pragma unroll
                        for (int r2 = 0; r2 < MT; ++r2) {
                            const double Adr = sOps[O::OFF_DR + (r2 * KV + t) * 64 + lane];
                            const double Ads = sOps[O::OFF_DS + (r2 * KV + t) * 64 + lane];
                            acc[0][r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Adr, a1, acc[0][r2], 0, 0, 0);
                            acc[1][r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Adr, a2, acc[1][r2], 0, 0, 0);
                            acc[2][r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Adr, a3, acc[2][r2], 0, 0, 0);
                            acc[0][r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ads, b1, acc[0][r2], 0, 0, 0);
                            acc[1][r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ads, b2, acc[1][r2], 0, 0, 0);
                            acc[2][r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ads, b3, acc[2][r2], 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- surface term, one face at a time (face node n = 4*tf + q of face f)
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            const double nxf = ld_row(ag + (4 + f) * ld, k8), nyf = ld_row(ag + (7 + f) * ld, k8);
            const double hfs = 0.5 * ld_row(ag + (10 + f) * ld, k8);
            double e1[KF], e2[KF], e3[KF], d1[KF], d2[KF], d3[KF];
            double e4[TRACER ? KF : 1], d4[TRACER ? KF : 1];
            double lam = 0.0;
#pragma unroll
            for (int tf = 0; tf < KF; ++tf) {
                const int n = 4 * tf + static_cast<int>(q);
                e1[tf] = e2[tf] = e3[tf] = d1[tf] = d2[tf] = d3[tf] = 0.0;
                if constexpr (TRACER) e4[tf] = d4[tf] = 0.0;
                if (n < Nfp) {
                    const int m = fmask_rt<N>(f, n);
                    const int id = fidx[f][tf];
                    double hM = ld_row(qin + m * ld, k8), huM = ld_row(qin + plane + m * ld, k8),
                           hvM = ld_row(qin + 2 * plane + m * ld, k8);
                    const unsigned idp = static_cast<unsigned>(id < 0 ? -(id + 1) : id), o8 = idp * 8u;
                    double hq, huq, hvq;
                    bool ghost = false;
                    unsigned rec8 = 0;
                    if constexpr (HALO) {
                        const unsigned row = idp / static_cast<unsigned>(ld), slot = idp - row * static_cast<unsigned>(ld);
                        ghost = slot >= static_cast<unsigned>(p.haloOwned);
                        rec8 = ((slot - static_cast<unsigned>(p.haloOwned)) * static_cast<unsigned>(p.haloRows) + row) * 8u;
                    }
                    if (ghost) { // the neighbour's record as it arrived: [field][node]
                        hq = ld_row(p.haloRecv, rec8);
                        huq = ld_row(p.haloRecv + Np, rec8);
                        hvq = ld_row(p.haloRecv + 2 * Np, rec8);
                    } else {
                        hq = ld_row(qin, o8);
                        huq = ld_row(qin + plane, o8);
                        hvq = ld_row(qin + 2 * plane, o8);
                    }
                    double nM = 0.0, nP = 0.0;
                    if constexpr (TRACER) {
                        nM = ld_row(qin + 3 * plane + m * ld, k8);
                        nP = ld_row(qin + 3 * plane, o8);
                    }
                    if constexpr (PHYS == 2) {
                        const double HM = ld_row(ph.H + m * ld, k8), HP = ld_row(ph.H, o8);
                        if ((btags >> (f * Nfp + n)) & 1) {   // open boundary (:348-353)
                            huq = huM;
                            hvq = hvM;
                            hq = HM + ph.tide;
                        } else if (id < 0) {                   // reflective wall (:340-345)
                            const double un = huM * nxf + hvM * nyf;
                            hq = hM;
                            huq = huM - 2 * nxf * un;
                            hvq = hvM - 2 * nyf * un;
                        }
                        const double bM = -HM, bP = -HP, mx = fmax(bP, bM);
                        const double hMs = fmax(0.0, hM + bM - mx), hPs = fmax(0.0, hq + bP - mx);
                        const double rMs = fast_rcp(hMs), rPs = fast_rcp(hPs);
                        huM = hMs * (huM * rMs); hvM = hMs * (hvM * rMs);  // hMstar*(huM/hM), hM = hMstar
                        huq = hPs * (huq * rPs); hvq = hPs * (hvq * rPs);
                        hM = hMs;
                        hq = hPs;
                    } else {
                        if (id < 0) { // reflective wall: no normal flow
                            const double un = huM * nxf + hvM * nyf;
                            huq = huM - 2 * nxf * un;
                            hvq = hvM - 2 * nyf * un;
                        }
                    }
                    const double rM = fast_rcp(hM), rP = fast_rcp(hq);
                    const double uM = huM * rM, vM = hvM * rM, uP = huq * rP, vP = hvq * rP;
                    if constexpr (PHYS != 2) {
                        const double spdM = fast_sqrt(uM * uM + vM * vM) + fast_sqrt(g * hM);
                        const double spdP = fast_sqrt(uP * uP + vP * vP) + fast_sqrt(g * hq);
                        lam = fmax(lam, fmax(spdM, spdP));
                    }
                    const double prM = halfg * hM * hM, prP = halfg * hq * hq;
                    const double F2M = huM * uM + prM, G2M = huM * vM, G3M = hvM * vM + prM;
                    const double F2P = huq * uP + prP, G2P = huq * vP, G3P = hvq * vP + prP;
                    d1[tf] = hM - hq; d2[tf] = huM - huq; d3[tf] = hvM - hvq;
                    e1[tf] = d2[tf] * nxf + d3[tf] * nyf;
                    e2[tf] = (F2M - F2P) * nxf + (G2M - G2P) * nyf;
                    e3[tf] = (G2M - G2P) * nxf + (G3M - G3P) * nyf;
                    if constexpr (TRACER) { // the tracer's '+' trace at a wall is the element's own value
                        d4[tf] = nM - nP;
                        e4[tf] = (nM * uM - nP * uP) * nxf + (nM * vM - nP * vP) * nyf;
                    }
                }
            }
            if constexpr (PHYS == 2) {
                lam = lamGlobal;
            } else {
                lam = fmax(lam, __shfl_xor(lam, 16));
                lam = fmax(lam, __shfl_xor(lam, 32));
            }
#pragma unroll
            for (int tf = 0; tf < KF; ++tf) {
                const double s1 = hfs * (e1[tf] - lam * d1[tf]);
                const double s2 = hfs * (e2[tf] - lam * d2[tf]);
                const double s3 = hfs * (e3[tf] - lam * d3[tf]);
#pragma unroll
                for (int r = 0; r < MT; ++r) {
                    const double Al = sOps[O::OFF_LIFT + ((r * 3 + f) * KF + tf) * 64 + lane];
                    acc[0][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(Al, s1, acc[0][r], 0, 0, 0);
                    acc[1][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(Al, s2, acc[1][r], 0, 0, 0);
                    acc[2][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(Al, s3, acc[2][r], 0, 0, 0);
                    if constexpr (TRACER)
                        acc[3][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(Al, hfs * (e4[tf] - lam * d4[tf]), acc[3][r], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- stage update / output, one field at a time
        if (live) {
            int sendRec[3] = {-1, -1, -1};
            if constexpr (HALO) {
                const unsigned b3 = (k - static_cast<unsigned>(p.kbegin)) * 3u;
                sendRec[0] = p.haloSendOf[b3];
                sendRec[1] = p.haloSendOf[b3 + 1];
                sendRec[2] = p.haloSendOf[b3 + 2];
            }
#pragma unroll
            for (int c = 0; c < NFLD; ++c) {
                const long long fo = static_cast<long long>(c) * plane;
                double oldv[MT][4], qv[MT][4];
                if constexpr (MODE != MODE_RHS) {
                    const double* __restrict__ base2 = ((MODE == MODE_LSERK) ? p.res : p.qbase) + fo;
#pragma unroll
                    for (int r = 0; r < MT; ++r)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const int i = 16 * r + static_cast<int>(q) + 4 * reg;
                            if (i < Np) {
                                qv[r][reg] = ld_row(qin + fo + i * ld, k8);
                                oldv[r][reg] = ld_row(base2 + i * ld, k8);
                            }
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int r = 0; r < MT; ++r)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int i = 16 * r + static_cast<int>(q) + 4 * reg;
                        if (i < Np) {
                            const double R = acc[c][r][reg];
                            if constexpr (MODE == MODE_RHS) {
                                st_row(p.rhs + fo + i * ld, k8, R);
                            } else if constexpr (MODE == MODE_LSERK) {
                                const double n1 = p.ca * oldv[r][reg] + p.cc * R;
                                const double qn = qv[r][reg] + p.cb * n1;
                                st_row(p.res + fo + i * ld, k8, n1);
                                st_row(p.qout + fo + i * ld, k8, qn);
                                if constexpr (HALO) {
#pragma unroll
                                    for (int sr = 0; sr < 3; ++sr)
                                        if (sendRec[sr] >= 0)
                                            p.haloSend[static_cast<size_t>(sendRec[sr]) * p.haloRows + c * Np + i] = qn;
                                }
                            } else {
                                const double val = p.ca * oldv[r][reg] + p.cb * qv[r][reg] + p.cc * R;
                                double sp = p.sponge;
                                if constexpr (PHYS == 2)
                                    if (ph.spongeField) sp = ld_row(ph.spongeField + i * ld, k8);
                                st_row(p.qout + fo + i * ld, k8, (c == 0 || c == 3) ? val : sponge_relax(val, sp));
                            }
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The passive tracer (field 3 of a four-field state; swhelpers/flux.py:17-19) on the matrix cores, as its own
// pass after the three-field kernel above, with the same tile / lane layout and operator image (MfmaOps2,
// plain or pre-filtered: the tracer has no sources): F4 = hN u, G4 = hN v, Lax-Friedrichs with the flow's
// wave speed. MT accumulator tiles only, so the register budget allows three waves per SIMD.
template <int N, int MODE>
__global__ __launch_bounds__(256, 3) void sw2d_stage_mfma2_tracer_kernel(const StageParams p) {
    using E = Elem<N>;
    using O = MfmaOps2<N>;
    constexpr int Np = E::Np, Nfp = E::Nfp, MT = O::MT, KV = O::KV, KF = O::KF;

    extern __shared__ double sOps[];
    stage_image<O::DOUBLES, 256>(sOps, p.opsAffine);
    __syncthreads();

    const unsigned lane = threadIdx.x & 63u, q = lane >> 4, j = lane & 15u;
    const unsigned nwg = gridDim.x, xcd = blockIdx.x % 8u, q8 = nwg / 8u, r8 = nwg % 8u;
    const unsigned blk = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + blockIdx.x / 8u;
    const unsigned wave = blk * 4u + (threadIdx.x >> 6), nwaves = nwg * 4u;
    const unsigned ntiles = (static_cast<unsigned>(p.kend - p.kbegin) + 15u) / 16u;
    const unsigned perWave = (ntiles + nwaves - 1u) / nwaves;
    const unsigned tileEnd = min(ntiles, (wave + 1u) * perWave);

    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld, fo = 3 * plane;
    const double* __restrict__ qin = p.qin;
    const double* __restrict__ ag = p.ageo;
    const double g = p.g;
    const unsigned kLast = static_cast<unsigned>(p.kend) - 1u;

#pragma unroll 1
    for (unsigned tile = wave * perWave; tile < tileEnd; ++tile) {
        const unsigned kTrue = static_cast<unsigned>(p.kbegin) + tile * 16u + j;
        const bool live = kTrue <= kLast;
        const unsigned k = live ? kTrue : kLast;
        const unsigned k8 = k * 8u, k4 = k * 4u;

        mfma_acc_t acc[MT];
#pragma unroll
        for (int r = 0; r < MT; ++r) acc[r] = mfma_acc_t{0.0, 0.0, 0.0, 0.0};
        int fidx[3][KF];
#pragma unroll
        for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int tf = 0; tf < KF; ++tf) {
                const int n = 4 * tf + static_cast<int>(q);
                fidx[f][tf] = n < Nfp ? ld_row(p.vmapP + (f * Nfp + n) * ld, k4) : 0;
            }

        // ---- volume term
        {
            const double rx = ld_row(ag, k8), sx = ld_row(ag + ld, k8), ry = ld_row(ag + 2 * ld, k8),
                         sy = ld_row(ag + 3 * ld, k8);
            constexpr int VC = 3, NC = (KV + VC - 1) / VC;
            double hb[2][VC], hub[2][VC], hvb[2][VC], hnb[2][VC];
            auto loadChunk = [&](int ch, int buf) {
#pragma unroll
                for (int s = 0; s < VC; ++s) {
                    const int t = ch * VC + s, m = 4 * t + static_cast<int>(q);
                    hb[buf][s] = 1.0; hub[buf][s] = 0.0; hvb[buf][s] = 0.0; hnb[buf][s] = 0.0;
                    if (t < KV && m < Np) {
                        hb[buf][s] = ld_row(qin + m * ld, k8);
                        hub[buf][s] = ld_row(qin + plane + m * ld, k8);
                        hvb[buf][s] = ld_row(qin + 2 * plane + m * ld, k8);
                        hnb[buf][s] = ld_row(qin + fo + m * ld, k8);
                    }
                }
            };
            loadChunk(0, 0);
#pragma unroll
            for (int ch = 0; ch < NC; ++ch) {
                const int cur = ch & 1;
                if (ch + 1 < NC) loadChunk(ch + 1, cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < VC; ++s) {
                    const int t = ch * VC + s;
                    if (t < KV) {
                        const int m = 4 * t + static_cast<int>(q);
                        const double r = fast_rcp(hb[cur][s]);
                        const double F4 = hnb[cur][s] * (hub[cur][s] * r), G4 = hnb[cur][s] * (hvb[cur][s] * r);
                        const double w = m < Np ? -1.0 : 0.0;
                        const double a = w * (rx * F4 + ry * G4), b = w * (sx * F4 + sy * G4);
#pragma unroll
                        for (int r2 = 0; r2 < MT; ++r2) {
                            acc[r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(sOps[O::OFF_DR + (r2 * KV + t) * 64 + lane], a, acc[r2], 0, 0, 0);
                            acc[r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(sOps[O::OFF_DS + (r2 * KV + t) * 64 + lane], b, acc[r2], 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- surface term
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            const double nxf = ld_row(ag + (4 + f) * ld, k8), nyf = ld_row(ag + (7 + f) * ld, k8);
            const double hfs = 0.5 * ld_row(ag + (10 + f) * ld, k8);
            double e[KF], d[KF];
            double lam = 0.0;
#pragma unroll
            for (int tf = 0; tf < KF; ++tf) {
                const int n = 4 * tf + static_cast<int>(q);
                e[tf] = d[tf] = 0.0;
                if (n < Nfp) {
                    const int m = fmask_rt<N>(f, n);
                    const int id = fidx[f][tf];
                    const double hM = ld_row(qin + m * ld, k8), huM = ld_row(qin + plane + m * ld, k8),
                                 hvM = ld_row(qin + 2 * plane + m * ld, k8), nM = ld_row(qin + fo + m * ld, k8);
                    const unsigned o8 = static_cast<unsigned>(id < 0 ? -(id + 1) : id) * 8u;
                    const double hq = ld_row(qin, o8), nP = ld_row(qin + fo, o8);
                    double huq = ld_row(qin + plane, o8), hvq = ld_row(qin + 2 * plane, o8);
                    if (id < 0) { // reflective wall (the tracer trace is the element's own)
                        const double un = huM * nxf + hvM * nyf;
                        huq = huM - 2 * nxf * un;
                        hvq = hvM - 2 * nyf * un;
                    }
                    const double rM = fast_rcp(hM), rP = fast_rcp(hq);
                    const double uM = huM * rM, vM = hvM * rM, uP = huq * rP, vP = hvq * rP;
                    const double spdM = fast_sqrt(uM * uM + vM * vM) + fast_sqrt(g * hM);
                    const double spdP = fast_sqrt(uP * uP + vP * vP) + fast_sqrt(g * hq);
                    lam = fmax(lam, fmax(spdM, spdP));
                    d[tf] = nM - nP;
                    e[tf] = (nM * uM - nP * uP) * nxf + (nM * vM - nP * vP) * nyf;
                }
            }
            lam = fmax(lam, __shfl_xor(lam, 16));
            lam = fmax(lam, __shfl_xor(lam, 32));
#pragma unroll
            for (int tf = 0; tf < KF; ++tf) {
                const double s4 = hfs * (e[tf] - lam * d[tf]);
#pragma unroll
                for (int r = 0; r < MT; ++r)
                    acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(sOps[O::OFF_LIFT + ((r * 3 + f) * KF + tf) * 64 + lane], s4, acc[r], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- stage update / output of the tracer field
        if (live) {
            double oldv[MT][4], qv[MT][4];
            if constexpr (MODE != MODE_RHS) {
                const double* __restrict__ base2 = ((MODE == MODE_LSERK) ? p.res : p.qbase) + fo;
#pragma unroll
                for (int r = 0; r < MT; ++r)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int i = 16 * r + static_cast<int>(q) + 4 * reg;
                        if (i < Np) {
                            qv[r][reg] = ld_row(qin + fo + i * ld, k8);
                            oldv[r][reg] = ld_row(base2 + i * ld, k8);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < MT; ++r)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int i = 16 * r + static_cast<int>(q) + 4 * reg;
                    if (i < Np) {
                        const double R = acc[r][reg];
                        if constexpr (MODE == MODE_RHS) {
                            st_row(p.rhs + fo + i * ld, k8, R);
                        } else if constexpr (MODE == MODE_LSERK) {
                            const double n1 = p.ca * oldv[r][reg] + p.cc * R;
                            st_row(p.res + fo + i * ld, k8, n1);
                            st_row(p.qout + fo + i * ld, k8, qv[r][reg] + p.cb * n1);
                        } else {
                            st_row(p.qout + fo + i * ld, k8, p.ca * oldv[r][reg] + p.cb * qv[r][reg] + p.cc * R);
                        }
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

} // namespace bdg_dev
