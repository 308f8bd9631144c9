// sw2d_mfma3src_kernel.hpp -- the state-once matrix-core schedule (sw2d_mfma3_kernel.hpp) for variants C / D of the
// reference's Python drivers: momentum sources (Coriolis, drag, bed slope; swhelpers/rhs.py:300-309, sw2d.py:140-141) and,
// with TRACER, the passive tracer hN as a fourth conserved field (F4 = hN u, G4 = hN v; swhelpers/flux.py:17-19), straight-
// sided elements, single-domain launches. sw2d.py itself runs N = 6 with tracer, Coriolis array, filter and midpoint RK2:
// on the two-waves-per-SIMD kernel (sw2d_stage_mfma2_kernel<N, MODE, 1, true>) that evaluation reads the state three
// times; here it is read once, exactly as for the three-field variant A.
//
// Same tile arithmetic as sw2d_stage_mfma2_kernel<N, MODE, 1, TRACER>:
//   * operator image = MfmaOps2 (plain or pre-filtered) + MT KV tiles of F' (identity or Filter) behind it: the sources
//     enter the volume term as two more matrix instructions per k-step and row block, R_2 += F' S_2, R_3 += F' S_3;
//   * the tracer is one more set of operands / accumulators / traces; its '+' trace at a wall is the element's own value
//     (the index table points there), the wave speed is the flow's.
// Same schedule as sw2d_stage_mfma3_kernel (one wave per SIMD, next tile requested piece by piece, update per 16-row
// block); the source tables' rows of the NEXT tile are requested with its state rows. A table that is absent gets an empty
// descriptor (its loads return 0) and the Coriolis constant is picked by a select: no branch in the tile loop.
// N = 8 (three fields only: four fields' state tiles would not fit the LDS beside the operators): nothing of the next tile
// is requested ahead -- it would not fit the registers, and a spilled register costs more than an exposed round trip.
// PHYS = 2: the stage kernel of variant B (the reference's tidal driver), see the template's comment.
// TPHASE (round 4; N = 8, where four fields' operands, accumulators and traces do not fit one wave's registers): the tracer
// equation as a SECOND PHASE of every tile, in the same launch. Phase one is the three-field kernel unchanged; when its last
// row block is stored the tile's h, hu, hv are still in the wave's LDS tile, so the tracer's volume operands (hN u, hN v) and
// its '-' traces come from there -- no second read of the state from HBM, which is what the separate tracer pass of rounds
// 2-3 cost (a launch that re-read h, hu, hv and took as long as the three conserved fields together). The neighbours' traces
// are gathered again (the registers that held them were given up to phase one's face products; these are L2 hits: a sibling
// wave streamed those rows moments ago). LDS: operators + F' tiles 69 KB + four waves x four fields x 5.6 KB = 157.5 KB.
#pragma once
#include "sw2d_mfma3_kernel.hpp"

namespace bdg_dev {

template <int N, bool TRACER>   // TRACER: four fields in the wave's state tile (fused tracer, or the tracer as a second phase)
struct Mfma3SrcLds {
    using O = MfmaOps2<N>;
    static constexpr int NF = TRACER ? 4 : 3;
    static constexpr int IMAGE = O::DOUBLES + O::MT * O::KV * 64;      // operators + F' tiles (r, t)
    static constexpr int TILE_DOUBLES = NF * Elem<N>::Np * 16;          // one wave's state tile [field][node][element]
    static constexpr int DOUBLES = IMAGE + 4 * TILE_DOUBLES;
};

// PHYS = 2: the surface term of the reference's variable-depth tidal driver (variant B, src/sw2d/main.cpp:340-368, as in
// sw2d_stage_mfma2_kernel<N, MODE, 2>): depth traces H at both sides of a face node, open-boundary nodes (bit mask per
// element) take the tide elevation, hydrostatic-reconstruction star states, the GLOBAL Lax-Friedrichs speed (a device
// scalar reduced beforehand), sources with the depth gradient (slope = +1, dragSign = -1), and in combine steps the
// per-node sponge coefficient. No tracer.
// IDF (round 4): the evaluation is not filtered, F' is the identity -- the sources are ADDED to the accumulator element of their own
// node (the lane that forms S(node 4 t + q) holds accumulator row 4 t + q of block t >> 2 as element t & 3) instead of being multiplied
// by MT KV identity tiles: 2 MT KV fewer matrix instructions per tile (72 of 369 at N = 8, three fields). The add goes where the
// product went in the order of operations on that accumulator (after the k-step's D products, i.e. in front of the next k-step's
// matrix instructions, behind that step's pointwise work: the products it follows have long retired), and the other row blocks'
// F' tiles are zero: bit-identical to the product form.
template <int N, int MODE, bool TRACER, int PHYS = 1, bool TPHASE = false, bool IDF = false>
__global__ __launch_bounds__(256, 1) void sw2d_stage_mfma3src_kernel(const StageParams p, const PhysParams ph) {
    static_assert(PHYS == 1 || (PHYS == 2 && !TRACER), "variant B has three fields");
    static_assert(!TPHASE || (PHYS == 1 && !TRACER), "the tracer phase follows the three-field form of variants C / D");
    using E = Elem<N>;
    using O = MfmaOps2<N>;
    using L = Mfma3SrcLds<N, TRACER || TPHASE>;
    constexpr int Np = E::Np, Nfp = E::Nfp, MT = O::MT, KV = O::KV, KF = O::KF, NF = TRACER ? 4 : 3;
    constexpr int NFD = (TRACER || TPHASE) ? 4 : 3;   // planes behind the state / residual / output descriptors
    constexpr int IMAGE = L::IMAGE, OFF_F = O::DOUBLES;
    // N = 8: nothing of the next tile is requested ahead (it would not fit the registers, and a spilled register's reload
    // drains the memory pipeline); the tile's own data is requested at its top: one exposed round trip per tile
    constexpr bool PF = KV < 12;
    // TPHASE: the tracer phase holds fewer registers than phase one, so the NEXT tile's indices, state rows and geometry are
    // requested during it (a state row per k-step of its volume term); its source rows and neighbour traces -- L2 hits, the
    // indices being there -- follow at the top of the tile. (Requesting those ahead as well: 114-174 spilled registers.)
    constexpr bool PF2 = TPHASE && !PF;

    extern __shared__ double sOps[];
    stage_image<IDF ? O::DOUBLES : IMAGE, 256>(sOps, p.opsAffine); // (IDF: the F' tiles are not used; the tiles' places in LDS stay where they are)
    __syncthreads();
    const int sBase = IMAGE + static_cast<int>(threadIdx.x >> 6) * L::TILE_DOUBLES + static_cast<int>(threadIdx.x & 15u);

    const unsigned lane = threadIdx.x & 63u, q = lane >> 4, j = lane & 15u;
    const unsigned nwg = gridDim.x, xcd = blockIdx.x % 8u, q8 = nwg / 8u, r8 = nwg % 8u;
    const unsigned blk = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + blockIdx.x / 8u;
    const unsigned ntiles = (static_cast<unsigned>(p.kend - p.kbegin) + 15u) / 16u;
    // the waves of one XCD walk its share of the tiles side by side (sw2d_mfma3_kernel.hpp)
    const unsigned wgHere = (nwg + 7u - xcd) / 8u, wgBefore = blk - blockIdx.x / 8u;
    unsigned tile = ntiles * wgBefore / nwg + (blockIdx.x / 8u) * 4u + (threadIdx.x >> 6);
    const unsigned tileEnd = ntiles * (wgBefore + wgHere) / nwg, tileStep = wgHere * 4u;
    if (tile >= tileEnd) return;

    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld;
    const double g = p.g, halfg = 0.5 * p.g;
    const unsigned kLast = static_cast<unsigned>(p.kend) - 1u;
    const unsigned planeBytes = static_cast<unsigned>(plane * 8), ld8 = static_cast<unsigned>(ld) * 8u, ld4 = static_cast<unsigned>(ld) * 4u;
    // one descriptor per array, the field picked by the scalar offset (an array of NF planes stays below 4 GiB: checked
    // by the launcher); the source tables get an empty descriptor when absent
    const __amdgpu_buffer_rsrc_t rq = plane_rsrc(p.qin, NFD * planeBytes),
                                 rold = plane_rsrc(MODE == MODE_LSERK ? p.res : (MODE == MODE_COMBINE ? p.qbase : p.qin), NFD * planeBytes),
                                 rout = plane_rsrc(MODE == MODE_RHS ? p.rhs : p.qout, NFD * planeBytes),
                                 rgeo = plane_rsrc(p.ageo, 13u * ld8), ridx = plane_rsrc(p.vmapP, 3u * Nfp * ld4),
                                 rsx = plane_rsrc(ph.sx ? ph.sx : p.ageo, ph.sx ? planeBytes : 0u),
                                 rsy = plane_rsrc(ph.sy ? ph.sy : p.ageo, ph.sy ? planeBytes : 0u),
                                 rfc = plane_rsrc(ph.fcor ? ph.fcor : p.ageo, ph.fcor ? planeBytes : 0u);
    const __amdgpu_buffer_rsrc_t rH = plane_rsrc(PHYS == 2 ? ph.H : p.ageo, PHYS == 2 ? planeBytes : 0u),
                                 robc = plane_rsrc(PHYS == 2 ? static_cast<const void*>(ph.obc) : static_cast<const void*>(p.ageo), PHYS == 2 ? ld4 : 0u),
                                 rsp = plane_rsrc((PHYS == 2 && ph.spongeField) ? ph.spongeField : p.ageo,
                                                  (PHYS == 2 && ph.spongeField) ? planeBytes : 0u);
    const bool hasSp = PHYS == 2 && ph.spongeField != nullptr;
    const double lamGlobal = PHYS == 2 ? *ph.lam : 0.0, tide = ph.tide;
    const bool hasF = ph.fcor != nullptr;
    const double cd = ph.cd, slope = ph.slope, dragSign = ph.dragSign, fconst = ph.fconst;

    auto elementOf = [&](unsigned tl, bool& live) {
        const unsigned kTrue = static_cast<unsigned>(p.kbegin) + tl * 16u + j;
        live = kTrue <= kLast;
        return live ? kTrue : kLast;
    };
    // state row t of all fields and the source tables' row t (node 4 t + q; padding rows: 0)
    auto loadStateRow = [&](unsigned kk, int t, double (&qs)[NF][KV], double (&src)[3][KV]) {
        const unsigned vo = row_voffset<Np, KV>(t, q, (q * static_cast<unsigned>(ld) + kk) * 8u), so = static_cast<unsigned>(4 * t) * ld8;
#pragma unroll
        for (int c = 0; c < NF; ++c) qs[c][t] = bld_f64(rq, vo, static_cast<unsigned>(c) * planeBytes + so);
        src[0][t] = bld_f64(rsx, vo, so);
        src[1][t] = bld_f64(rsy, vo, so);
        src[2][t] = bld_f64(rfc, vo, so);
    };
    auto loadStateOnly = [&](unsigned kk, int t, double (&qs)[NF][KV]) {
        const unsigned vo = row_voffset<Np, KV>(t, q, (q * static_cast<unsigned>(ld) + kk) * 8u), so = static_cast<unsigned>(4 * t) * ld8;
#pragma unroll
        for (int c = 0; c < NF; ++c) qs[c][t] = bld_f64(rq, vo, static_cast<unsigned>(c) * planeBytes + so);
    };
    auto loadSourceRow = [&](unsigned kk, int t, double (&src)[3][KV]) {
        const unsigned vo = row_voffset<Np, KV>(t, q, (q * static_cast<unsigned>(ld) + kk) * 8u), so = static_cast<unsigned>(4 * t) * ld8;
        src[0][t] = bld_f64(rsx, vo, so);
        src[1][t] = bld_f64(rsy, vo, so);
        src[2][t] = bld_f64(rfc, vo, so);
    };
    auto loadIndices = [&](unsigned kk, int (&ix)[3][KF]) {
        const unsigned v4 = (q * static_cast<unsigned>(ld) + kk) * 4u;
#pragma unroll
        for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int tf = 0; tf < KF; ++tf)
                ix[f][tf] = bld_i32(ridx, (4 * tf + static_cast<int>(q) < Nfp) ? v4 : 0xfffffffcu, static_cast<unsigned>(f * Nfp + 4 * tf) * ld4);
    };
    auto loadGeometry = [&](unsigned kk, double (&gg)[13]) {
#pragma unroll
        for (int i = 0; i < 13; ++i) gg[i] = bld_f64(rgeo, kk * 8u, static_cast<unsigned>(i) * ld8);
    };
    // neighbour traces of face f (face node n = 4 tf + q); lanes beyond the face read node 0 of element 0 and ignore it
    // (PHYS = 2: also the depth at the node itself and at its neighbour, dep[0] = H-, dep[1] = H+)
    auto loadTraces = [&](int f, unsigned kk, const int (&ix)[3][KF], double (&tr)[NF][3][KF], double (&dep)[2][3][KF]) {
#pragma unroll
        for (int tf = 0; tf < KF; ++tf) {
            const int n = 4 * tf + static_cast<int>(q);
            const int id = n < Nfp ? ix[f][tf] : 0;
            const unsigned o8 = static_cast<unsigned>(id < 0 ? -(id + 1) : id) * 8u;
#pragma unroll
            for (int c = 0; c < NF; ++c) tr[c][f][tf] = bld_f64(rq, o8, static_cast<unsigned>(c) * planeBytes);
            if constexpr (PHYS == 2) {
                const int m = n < Nfp ? fmask_rt<N>(f, n) : 0;
                dep[0][f][tf] = bld_f64(rH, (static_cast<unsigned>(m) * static_cast<unsigned>(ld) + kk) * 8u, 0u);
                dep[1][f][tf] = bld_f64(rH, o8, 0u);
            }
        }
    };

    bool live;
    unsigned k = elementOf(tile, live);
    double qB[NF][KV], srcB[3][KV], geo[13], trP[NF][3][KF], depP[2][3][KF];
    int fidx[3][KF], btags = 0; // btags (PHYS = 2): bit f Nfp + n set = open-boundary node n of face f
    auto loadTile = [&](unsigned kk) { // everything a tile needs before its first product
        loadIndices(kk, fidx);
#pragma unroll
        for (int t = 0; t < KV; ++t) loadStateRow(kk, t, qB, srcB);
        loadGeometry(kk, geo);
#pragma unroll
        for (int f = 0; f < 3; ++f) loadTraces(f, kk, fidx, trP, depP);
        if constexpr (PHYS == 2) btags = bld_i32(robc, kk * 4u, 0u);
    };
    if constexpr (PF) loadTile(k); // the first tile; the following ones are requested piece by piece a tile ahead
    if constexpr (PF2) {           // ... or their indices, state rows and geometry only (TPHASE)
        loadIndices(k, fidx);
#pragma unroll
        for (int t = 0; t < KV; ++t) loadStateOnly(k, t, qB);
        loadGeometry(k, geo);
    }

#pragma unroll 1
    for (;;) {
        const unsigned v8 = (q * static_cast<unsigned>(ld) + k) * 8u;
        const bool more = tile + tileStep < tileEnd;
        bool liveN = false;
        const unsigned kN = more ? elementOf(tile + tileStep, liveN) : k;
        double qN[NF][KV], srcN[3][KV], geoN[13], trN[NF][3][KF], depN[2][3][KF];
        int btagsN = 0;
        int fidxN[3][KF];
        if constexpr (!PF && !PF2) loadTile(k);
        if constexpr (PF2) { // no branch around any of these requests (a join of control flow costs the prefetches their overlap)
#pragma unroll
            for (int t = 0; t < KV; ++t) loadSourceRow(k, t, srcB);
#pragma unroll
            for (int f = 0; f < 3; ++f) loadTraces(f, k, fidx, trP, depP);
        }

        // ---- own state into the wave's LDS tile (face traces and the update read it back from there)
#pragma unroll
        for (int c = 0; c < NF; ++c)
#pragma unroll
            for (int t = 0; t < KV; ++t) {
                const int m = 4 * t + static_cast<int>(q);
                if (m < Np) sOps[sBase + (c * Np + m) * 16] = qB[c][t];
            }
        __builtin_amdgcn_wave_barrier();

        double oldv[NF][KV], spv[KV]; // spv (PHYS = 2, combine steps): the nodes' sponge coefficients (0 without a field)
        auto loadOldRow = [&](int t) {
            if constexpr (MODE != MODE_RHS) {
#pragma unroll
                for (int c = 0; c < NF; ++c)
                    oldv[c][t] = bld_f64(rold, row_voffset<Np, KV>(t, q, v8), static_cast<unsigned>(c) * planeBytes + static_cast<unsigned>(4 * t) * ld8);
                if constexpr (PHYS == 2 && MODE == MODE_COMBINE)
                    spv[t] = bld_f64(rsp, row_voffset<Np, KV>(t, q, v8), static_cast<unsigned>(4 * t) * ld8);
            }
        };
        mfma_acc_t acc[NF][MT];
#pragma unroll
        for (int c = 0; c < NF; ++c)
#pragma unroll
            for (int r = 0; r < MT; ++r) acc[c][r] = mfma_zero();

        // operands of k-step t: two per field (Dr and Ds side) and the two source values
        auto volumeOperands = [&](int t, double (&ab)[2 * NF], double (&ss)[2]) {
            const int m = 4 * t + static_cast<int>(q);
            const bool pad = m >= Np;
            const double h = pad ? 1.0 : qB[0][t], hu = qB[1][t], hv = qB[2][t];
            const double r = fast_rcp(h);
            const double u = hu * r, v = hv * r;
            const double pr = halfg * h * h;
            const double F2 = hu * u + pr, G2 = hu * v, G3 = hv * v + pr;
            const double w = pad ? 0.0 : -1.0; // zero the padded rows of the operand
            const double rx = geo[0], sx = geo[1], ry = geo[2], sy = geo[3];
            ab[0] = w * (rx * hu + ry * hv); ab[1] = w * (sx * hu + sy * hv);
            ab[2] = w * (rx * F2 + ry * G2); ab[3] = w * (sx * F2 + sy * G2);
            ab[4] = w * (rx * G2 + ry * G3); ab[5] = w * (sx * G2 + sy * G3);
            if constexpr (TRACER) {
                const double F4 = qB[3][t] * u, G4 = qB[3][t] * v;
                ab[6] = w * (rx * F4 + ry * G4); ab[7] = w * (sx * F4 + sy * G4);
            }
            // S2 = f hv - cd |u| u + slope g h sx, S3 = -f hu + dragSign cd |u| v + slope g h sy (sw2d_affine_kernel.hpp)
            const double fco = hasF ? srcB[2][t] : fconst;
            const double cdn = cd * fast_sqrt(u * u + v * v), gh = slope * g * h;
            ss[0] = -w * fma(gh, srcB[0][t], fma(fco, hv, -(cdn * u)));
            ss[1] = -w * fma(gh, srcB[1][t], fma(dragSign * cdn, v, -(fco * hu)));
        };
        // pointwise work of the surface term, one face node per lane at a time, spread over the volume k-steps
        double sF[3][NF][KF];
        double eF[NF][KF], dF[NF][KF], lamF = 0.0;
        double lamKeep[TPHASE ? 3 : 1]; // (TPHASE) the faces' speeds, kept for the tracer phase: 36 square roots per tile it need not redo
        auto faceNode = [&](int f, int tf) {
            const double nxf = geo[4 + f], nyf = geo[7 + f];
            const int n = 4 * tf + static_cast<int>(q);
#pragma unroll
            for (int c = 0; c < NF; ++c) eF[c][tf] = dF[c][tf] = 0.0;
            if (n < Nfp) {
                const int m = fmask_rt<N>(f, n);
                const double hM0 = sOps[sBase + m * 16], huM0 = sOps[sBase + (Np + m) * 16], hvM0 = sOps[sBase + (2 * Np + m) * 16];
                double hM = hM0, huM = huM0, hvM = hvM0, hq = trP[0][f][tf];
                double huq = trP[1][f][tf], hvq = trP[2][f][tf];
                if constexpr (PHYS == 2) { // as sw2d_stage_mfma2_kernel<N, MODE, 2> (src/sw2d/main.cpp:340-368)
                    const double HM = depP[0][f][tf], HP = depP[1][f][tf];
                    if ((btags >> (f * Nfp + n)) & 1) {   // open boundary (:348-353)
                        huq = huM;
                        hvq = hvM;
                        hq = HM + tide;
                    } else if (fidx[f][tf] < 0) {          // reflective wall (:340-345)
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
                    if (fidx[f][tf] < 0) { // reflective wall: no normal flow
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
                    lamF = fmax(lamF, fmax(spdM, spdP));
                }
                const double prM = halfg * hM * hM, prP = halfg * hq * hq;
                const double F2M = huM * uM + prM, G2M = huM * vM, G3M = hvM * vM + prM;
                const double F2P = huq * uP + prP, G2P = huq * vP, G3P = hvq * vP + prP;
                dF[0][tf] = hM - hq; dF[1][tf] = huM - huq; dF[2][tf] = hvM - hvq;
                eF[0][tf] = dF[1][tf] * nxf + dF[2][tf] * nyf;
                eF[1][tf] = (F2M - F2P) * nxf + (G2M - G2P) * nyf;
                eF[2][tf] = (G2M - G2P) * nxf + (G3M - G3P) * nyf;
                if constexpr (TRACER) { // the tracer's '+' trace at a wall is the element's own value (index table)
                    const double nM = sOps[sBase + (3 * Np + m) * 16], nP = trP[3][f][tf];
                    dF[3][tf] = nM - nP;
                    eF[3][tf] = (nM * uM - nP * uP) * nxf + (nM * vM - nP * vP) * nyf;
                }
            }
            if (tf == KF - 1) { // the face is complete: its speed (the face's nodes sit in the 4 lanes q of this element)
                double lam = lamGlobal;
                if constexpr (PHYS != 2) {
                    lam = fmax(lamF, __shfl_xor(lamF, 16));
                    lam = fmax(lam, __shfl_xor(lam, 32));
                }
                const double hfs = 0.5 * geo[10 + f];
                if constexpr (TPHASE) lamKeep[f] = lam;
#pragma unroll
                for (int t2 = 0; t2 < KF; ++t2)
#pragma unroll
                    for (int c = 0; c < NF; ++c) sF[f][c][t2] = hfs * (eF[c][t2] - lam * dF[c][t2]);
                lamF = 0.0;
                if constexpr (PF) loadTraces(f, kN, fidxN, trN, depN); // this face's '+' traces are dead: request the next tile's
            }
        };
        constexpr int FACE_ITEMS = 3 * KF, PER_STEP = (FACE_ITEMS + KV - 1) / KV;

        if constexpr (PF) loadIndices(kN, fidxN);
        if constexpr (PF && PHYS == 2) btagsN = bld_i32(robc, kN * 4u, 0u);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int OLD_EARLY = (MT > 1 ? 4 * (MT - 1) : 0) < KV ? (MT > 1 ? 4 * (MT - 1) : 0) : KV;
        double ssPend[2] = {0.0, 0.0}; // (IDF) the previous k-step's sources, added in front of this step's matrix instructions
#pragma unroll
        for (int t = 0; t < KV; ++t) {
            double ab[2 * NF], ss[2];
            volumeOperands(t, ab, ss);
            if constexpr (PF) loadStateRow(kN, t, qN, srcN);
            if (t >= KV - OLD_EARLY) loadOldRow(t - (KV - OLD_EARLY));
#pragma unroll
            for (int it = t * PER_STEP; it < (t + 1) * PER_STEP; ++it)
                if (it < FACE_ITEMS) faceNode(it / KF, it % KF);
            if constexpr (IDF) {
                if (t > 0) {
                    acc[1][(t - 1) >> 2][(t - 1) & 3] += ssPend[0];
                    acc[2][(t - 1) >> 2][(t - 1) & 3] += ssPend[1];
                }
                ssPend[0] = ss[0];
                ssPend[1] = ss[1];
            }
#pragma unroll
            for (int r2 = 0; r2 < MT; ++r2) {
                const double Adr = sOps[O::OFF_DR + (r2 * KV + t) * 64 + lane];
                const double Ads = sOps[O::OFF_DS + (r2 * KV + t) * 64 + lane];
#pragma unroll
                for (int c = 0; c < NF; ++c) acc[c][r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Adr, ab[2 * c], acc[c][r2], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < NF; ++c) acc[c][r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ads, ab[2 * c + 1], acc[c][r2], 0, 0, 0);
                if constexpr (!IDF) {
                    const double Af = sOps[OFF_F + (r2 * KV + t) * 64 + lane];
                    acc[1][r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Af, ss[0], acc[1][r2], 0, 0, 0);
                    acc[2][r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Af, ss[1], acc[2][r2], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (IDF) {
            acc[1][(KV - 1) >> 2][(KV - 1) & 3] += ssPend[0];
            acc[2][(KV - 1) >> 2][(KV - 1) & 3] += ssPend[1];
        }
#pragma unroll
        for (int t = OLD_EARLY; t < KV; ++t) loadOldRow(t);
        if constexpr (PF) loadGeometry(kN, geoN);
        __builtin_amdgcn_sched_barrier(0);

        // ---- stage update / output of the rows of block r (node m = 4 t + q is accumulator row 16 (t >> 2) + q + 4 (t & 3))
        auto updateBlock = [&](int r) {
            if (!live) return;
            double own[NF][4];
#pragma unroll
            for (int c = 0; c < NF; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = 4 * (4 * r + e) + static_cast<int>(q);
                    own[c][e] = (4 * r + e < KV && m < Np) ? sOps[sBase + (c * Np + m) * 16] : 0.0;
                }
#pragma unroll
            for (int c = 0; c < NF; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int t = 4 * r + e;
                    if (t >= KV) continue;
                    const unsigned vo = row_voffset<Np, KV>(t, q, v8), soff = static_cast<unsigned>(c) * planeBytes + static_cast<unsigned>(4 * t) * ld8;
                    const double R = acc[c][r][e];
                    if constexpr (MODE == MODE_RHS) {
                        bst_f64(rout, vo, soff, R);
                    } else if constexpr (MODE == MODE_LSERK) {
                        const double n1 = p.ca * oldv[c][t] + p.cc * R;
                        bst_f64(rold, vo, soff, n1); // the residual, in place
                        bst_f64(rout, vo, soff, own[c][e] + p.cb * n1);
                    } else {
                        const double val = p.ca * oldv[c][t] + p.cb * own[c][e] + p.cc * R;
                        const double sp = (PHYS == 2 && hasSp) ? spv[t] : p.sponge;
                        bst_f64(rout, vo, soff, (c == 1 || c == 2) ? sponge_relax(val, sp) : val);
                    }
                }
        };

        // (TPHASE) the tracer's own rows, requested here so that they are there when the second phase starts
        double qT[TPHASE ? KV : 1];
        if constexpr (TPHASE) {
#pragma unroll
            for (int t = 0; t < KV; ++t) qT[t] = bld_f64(rq, row_voffset<Np, KV>(t, q, v8), 3u * planeBytes + static_cast<unsigned>(4 * t) * ld8);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- surface term: matrix instructions only, one block of 16 output rows at a time, then that block's update
#pragma unroll
        for (int r = 0; r < MT; ++r) {
#pragma unroll
            for (int f = 0; f < 3; ++f)
#pragma unroll
                for (int tf = 0; tf < KF; ++tf) {
                    const double Al = sOps[O::OFF_LIFT + ((r * 3 + f) * KF + tf) * 64 + lane];
#pragma unroll
                    for (int c = 0; c < NF; ++c) acc[c][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(Al, sF[f][c][tf], acc[c][r], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            updateBlock(r);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- TPHASE: the tracer equation of this tile (F4 = hN u, G4 = hN v, Lax-Friedrichs with the flow's speed, no sources;
        //      swhelpers/flux.py:17-19, rhs.py:253-258) from the state tile that is still in LDS
        if constexpr (TPHASE) {
            const unsigned f3 = 3u * planeBytes;
            double old4[KV], tq[4][3][KF];
#pragma unroll
            for (int f = 0; f < 3; ++f)
#pragma unroll
                for (int tf = 0; tf < KF; ++tf) {
                    const int n = 4 * tf + static_cast<int>(q);
                    const int id = n < Nfp ? fidx[f][tf] : 0;
                    const unsigned o8 = static_cast<unsigned>(id < 0 ? -(id + 1) : id) * 8u;
#pragma unroll
                    for (int c = 0; c < 4; ++c) tq[c][f][tf] = bld_f64(rq, o8, static_cast<unsigned>(c) * planeBytes);
                }
            if constexpr (MODE != MODE_RHS) {
#pragma unroll
                for (int t = 0; t < KV; ++t) old4[t] = bld_f64(rold, row_voffset<Np, KV>(t, q, v8), f3 + static_cast<unsigned>(4 * t) * ld8);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < KV; ++t) {
                const int m = 4 * t + static_cast<int>(q);
                if (m < Np) sOps[sBase + (3 * Np + m) * 16] = qT[t];
            }
            __builtin_amdgcn_wave_barrier();
            mfma_acc_t a4[MT];
#pragma unroll
            for (int r = 0; r < MT; ++r) a4[r] = mfma_zero();
            if constexpr (PF2) { // (kN = k on a wave's last tile: a harmless repeat instead of a branch)
                loadIndices(kN, fidxN);
                loadGeometry(kN, geoN);
            }
            // volume term: the lane's own nodes m = 4 t + q, all four fields back from the LDS tile
#pragma unroll
            for (int t = 0; t < KV; ++t) {
                if constexpr (PF2) loadStateOnly(kN, t, qN);
                const int m = 4 * t + static_cast<int>(q);
                const bool pad = m >= Np;
                const int ms = pad ? 0 : m;
                const double h = pad ? 1.0 : sOps[sBase + ms * 16], hu = sOps[sBase + (Np + ms) * 16], hv = sOps[sBase + (2 * Np + ms) * 16];
                const double hn = sOps[sBase + (3 * Np + ms) * 16];
                const double r = fast_rcp(h);
                const double F4 = hn * (hu * r), G4 = hn * (hv * r);
                const double w = pad ? 0.0 : -1.0;
                const double a = w * (geo[0] * F4 + geo[2] * G4), b = w * (geo[1] * F4 + geo[3] * G4);
#pragma unroll
                for (int r2 = 0; r2 < MT; ++r2) {
                    const double Adr = sOps[O::OFF_DR + (r2 * KV + t) * 64 + lane];
                    const double Ads = sOps[O::OFF_DS + (r2 * KV + t) * 64 + lane];
                    a4[r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Adr, a, a4[r2], 0, 0, 0);
                    a4[r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ads, b, a4[r2], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // surface term: the same traces and wall states as phase one, its face speeds (lamKeep)
            double s4[3][KF];
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                const double nxf = geo[4 + f], nyf = geo[7 + f];
                double e4[KF], d4[KF];
                const double lam4 = lamKeep[f];
#pragma unroll
                for (int tf = 0; tf < KF; ++tf) {
                    const int n = 4 * tf + static_cast<int>(q);
                    e4[tf] = d4[tf] = 0.0;
                    if (n < Nfp) {
                        const int m = fmask_rt<N>(f, n);
                        const double hM = sOps[sBase + m * 16], huM = sOps[sBase + (Np + m) * 16], hvM = sOps[sBase + (2 * Np + m) * 16];
                        const double nM = sOps[sBase + (3 * Np + m) * 16];
                        const double hq = tq[0][f][tf], nP = tq[3][f][tf];
                        double huq = tq[1][f][tf], hvq = tq[2][f][tf];
                        if (fidx[f][tf] < 0) { // reflective wall: no normal flow
                            const double un = huM * nxf + hvM * nyf;
                            huq = huM - 2 * nxf * un;
                            hvq = hvM - 2 * nyf * un;
                        }
                        const double rM = fast_rcp(hM), rP = fast_rcp(hq);
                        const double uM = huM * rM, vM = hvM * rM, uP = huq * rP, vP = hvq * rP;
                        d4[tf] = nM - nP;
                        e4[tf] = (nM * uM - nP * uP) * nxf + (nM * vM - nP * vP) * nyf;
                    }
                }
                const double hfs = 0.5 * geo[10 + f];
#pragma unroll
                for (int tf = 0; tf < KF; ++tf) s4[f][tf] = hfs * (e4[tf] - lam4 * d4[tf]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < MT; ++r) {
#pragma unroll
                for (int f = 0; f < 3; ++f)
#pragma unroll
                    for (int tf = 0; tf < KF; ++tf) {
                        const double Al = sOps[O::OFF_LIFT + ((r * 3 + f) * KF + tf) * 64 + lane];
                        a4[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(Al, s4[f][tf], a4[r], 0, 0, 0);
                    }
                if (live) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int t = 4 * r + e;
                        if (t >= KV) continue;
                        const int m = 4 * t + static_cast<int>(q);
                        const double own = m < Np ? sOps[sBase + (3 * Np + m) * 16] : 0.0;
                        const unsigned vo = row_voffset<Np, KV>(t, q, v8), soff = f3 + static_cast<unsigned>(4 * t) * ld8;
                        const double R = a4[r][e];
                        if constexpr (MODE == MODE_RHS) {
                            bst_f64(rout, vo, soff, R);
                        } else if constexpr (MODE == MODE_LSERK) {
                            const double n1 = p.ca * old4[t] + p.cc * R;
                            bst_f64(rold, vo, soff, n1);
                            bst_f64(rout, vo, soff, own + p.cb * n1);
                        } else {
                            bst_f64(rout, vo, soff, p.ca * old4[t] + p.cb * own + p.cc * R);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        if (!more) break;
        tile += tileStep;
        k = kN;
        live = liveN;
        if constexpr (!PF && !PF2) continue;
        if constexpr (PF2) {
#pragma unroll
            for (int t = 0; t < KV; ++t)
#pragma unroll
                for (int c = 0; c < NF; ++c) qB[c][t] = qN[c][t];
#pragma unroll
            for (int i = 0; i < 13; ++i) geo[i] = geoN[i];
#pragma unroll
            for (int f = 0; f < 3; ++f)
#pragma unroll
                for (int tf = 0; tf < KF; ++tf) fidx[f][tf] = fidxN[f][tf];
            continue;
        }
#pragma unroll
        for (int t = 0; t < KV; ++t) {
#pragma unroll
            for (int c = 0; c < NF; ++c) qB[c][t] = qN[c][t];
#pragma unroll
            for (int i = 0; i < 3; ++i) srcB[i][t] = srcN[i][t];
        }
#pragma unroll
        for (int i = 0; i < 13; ++i) geo[i] = geoN[i];
        btags = btagsN;
#pragma unroll
        for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int tf = 0; tf < KF; ++tf) {
                fidx[f][tf] = fidxN[f][tf];
#pragma unroll
                for (int c = 0; c < NF; ++c) trP[c][f][tf] = trN[c][f][tf];
                if constexpr (PHYS == 2) { depP[0][f][tf] = depN[0][f][tf]; depP[1][f][tf] = depN[1][f][tf]; }
            }
    }
}

} // namespace bdg_dev
