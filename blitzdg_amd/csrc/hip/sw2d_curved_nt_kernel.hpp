// sw2d_curved_nt_kernel.hpp -- curved / over-integrated shallow-water RHS, "nodal-trace" form (round 3).
//
// Same mathematics as sw2d_curved_kernel.hpp (reference swhelpers/rhs.py:6-176), one launch instead of two per
// evaluation and about half the bytes. What changed, and why it is the same function:
//
//  * No Gauss-trace planes. The first form materialised gq = Interp q of every element (its own kernel, 12 NGauss
//    doubles written and read back per element) because the exterior trace of a Gauss point lives in the neighbour.
//    But the trace of a degree-N polynomial on an edge is fixed by its N+1 edge nodes: the rows of gauss_ctx.Interp
//    that belong to face f are zero outside Fmask(:, f). So the exterior trace at my Gauss points is MY face block of
//    Interp applied to the neighbour's values at the nodes that coincide with my face nodes:
//        gP(face f) = IF_f (NG x Nfp)  *  q[nodeP(f, :)],     nodeP(f, i) = the neighbour's node at my face node i
//    -- 3 Nfp gathered doubles per field instead of 3 NG written + 3 NG read, and the gathered rows are the state rows
//    the neighbouring tile streams anyway (L2). nodeP is derived at creation from gmapP (any map that pairs whole
//    faces: interior, wall, periodic) and from Interp itself (the face nodes are the columns that are not zero; the
//    node pairing of two faces is the permutation that makes their interpolation columns agree at the paired Gauss
//    points). A context that does not have this structure (a rewired gmapM, a map that pairs single points across
//    different faces, an Interp with entries off the face) keeps the first form (sw2d_curved_kernel.hpp).
//  * Straight-sided tiles read 14 numbers per element: W rx .. W sy = wref * c (as before), and now also the Gauss
//    geometry -- nx, ny constant per face, W = gwref * s_f -- and 1 / J, instead of 9 NGauss + Np doubles.
//  * Orders 7 and 8: the operator image (>= 195 KB) does not fit LDS. The volume term's tiles are streamed through a
//    double buffer, one 16-cubature-row chunk at a time, by the four waves of a workgroup in lockstep (one barrier
//    per chunk; the next chunk's loads are in flight while the current one is multiplied); surface and mass tiles
//    stay resident. The first form read every A tile from L2 per wave and per tile.
//
// Layout of lanes, operands and accumulators: as described at the top of sw2d_curved_kernel.hpp.
#pragma once
#include "sw2d_curved_kernel.hpp"
#include <type_traits>

namespace bdg_dev {

// Operator image of the nodal-trace kernel: zero-padded 16 x 4 A tiles, 64 doubles each, entry l of a tile =
// A[row l & 15][step column l >> 4], grouped so that what one phase needs is contiguous:
//   volume chunk rb (VCH tiles):   Vc[t]            row 16 rb + i = cubature point, column 4 t + s = node
//                                  DrT[r][reg]      row 16 r + i = node, column s <-> cubature point 16 rb + 4 reg + s
//                                  DsT[r][reg]
//   surface block gb = f FB + b (SCH tiles):
//                                  GE[t2]           row = Gauss row 16 b + i of face f, column 4 t2 + s = FACE node of face f
//                                                   (Interp(:, Fmask(:, f)): serves the exterior AND the own traces)
//                                  IT[r][reg]       row 16 r + i = node, column s <-> Gauss row 16 b + 4 reg + s (-Interp^T)
//   mass (3 MT KV tiles):          M[r][t] = V V^T,  MF[r][t] = Filter V V^T,  F[r][t] = Filter
template <int N>
struct CurvedOpsNT {
    static constexpr int Np = (N + 1) * (N + 2) / 2;
    static constexpr int Nfp = N + 1;
    static constexpr int KV = (Np + 3) / 4;
    static constexpr int MT = (Np + 15) / 16;
    static constexpr int KE = (Nfp + 3) / 4;
    static constexpr int VCH = KV + 8 * MT;
    static constexpr int SCH = KE + 4 * MT;
    __host__ __device__ static constexpr int offVol(int rb) { return rb * VCH; }
    __host__ __device__ static constexpr int offSurf(int ncb, int gb) { return ncb * VCH + gb * SCH; }
    __host__ __device__ static constexpr int offMass(int ncb, int fb) { return ncb * VCH + 3 * fb * SCH; }
    __host__ __device__ static constexpr int tiles(int ncb, int fb) { return offMass(ncb, fb) + 3 * MT * KV; }
};

// STREAM = 0: whole image resident in LDS. STREAM = 1: volume chunks through a double buffer, the workgroup's waves in
// lockstep; resident: surface blocks, the mass tiles this launch uses (M or MF, and F when FILTER).
// RL: 4-row steps of a face's LAST 16-row block that hold Gauss points (ceil((NG - 16 (FB - 1)) / 4)): the pointwise work
// and the lift products of the steps beyond are skipped (NG = 10: 3 of 4 steps; NG = 18: 1 of 4 in the second block).
template <int N, int MODE, bool FILTER, int STREAM, int FB, int WAVES, int RL = 4>
__global__ __launch_bounds__(256, WAVES) void sw2d_curved_nt_kernel(const CurvedParams p) {
    using O = CurvedOpsNT<N>;
    constexpr int Np = O::Np, KV = O::KV, MT = O::MT, KE = O::KE, VCH = O::VCH, SCH = O::SCH;
    extern __shared__ double sOps[];
#ifdef BDG_PHASE_CLOCK
    const unsigned long long entryClk = __builtin_readcyclecounter(), entryReal = __builtin_amdgcn_s_memrealtime();
#endif
    const int ncb = p.ncb;
    constexpr int fb = FB;
    typedef double f64x2 __attribute__((ext_vector_type(2)));

    // ---- LDS map (in tiles of 64 doubles): [stream buffers 2 VCH] [resident tiles] [wref 16 ncb] [gwref 16 FB] [face nodes, 12 KE ints]
    const int imgSurf = O::offSurf(ncb, 0), imgMass = O::offMass(ncb, fb);
    const int nSurf = 3 * fb * SCH;
    constexpr int massTiles = MT * KV;
    // resident base (tile index inside LDS) of the surface blocks, of the mass tiles and of the filter tiles
    const int ldsSurf = STREAM ? 2 * VCH : imgSurf;
    const int ldsMass = STREAM ? ldsSurf + nSurf : imgMass + (FILTER ? massTiles : 0);
    const int ldsF = STREAM ? ldsMass + massTiles : imgMass + 2 * massTiles;
    const int ldsTiles = STREAM ? ldsMass + (FILTER ? 2 : 1) * massTiles : O::tiles(ncb, fb);
    const int wrefAt = ldsTiles * 64, gwrefAt = wrefAt + 16 * ncb, fnAt = gwrefAt + 16 * fb;
    int* const sFn = reinterpret_cast<int*>(sOps + fnAt);

    auto copyTiles = [&](const double* src, int dstTile, int ntile) { // all of a thread's 16-byte pieces of a batch in flight
        const f64x2* __restrict__ s2 = reinterpret_cast<const f64x2*>(src);
        f64x2* __restrict__ d2 = reinterpret_cast<f64x2*>(sOps + static_cast<size_t>(dstTile) * 64);
        const int pairs = ntile * 32, nthreads = static_cast<int>(blockDim.x);
        for (int b = threadIdx.x; b < pairs; b += 8 * nthreads) {
            f64x2 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (b + i * nthreads < pairs) v[i] = s2[b + i * nthreads];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (b + i * nthreads < pairs) d2[b + i * nthreads] = v[i];
        }
    };
    if constexpr (STREAM) {
        copyTiles(p.opsNT + static_cast<size_t>(imgSurf) * 64, ldsSurf, nSurf);
        copyTiles(p.opsNT + static_cast<size_t>(imgMass + (FILTER ? massTiles : 0)) * 64, ldsMass, massTiles);
        if constexpr (FILTER) copyTiles(p.opsNT + static_cast<size_t>(imgMass + 2 * massTiles) * 64, ldsF, massTiles);
        copyTiles(p.opsNT, 0, VCH); // chunk 0 of the first tile
    } else {
        copyTiles(p.opsNT, 0, O::tiles(ncb, fb));
    }
    for (int t = threadIdx.x; t < 16 * ncb; t += blockDim.x) sOps[wrefAt + t] = p.cubWref ? p.cubWref[t] : 0.0;
    for (int t = threadIdx.x; t < 16 * fb; t += blockDim.x) sOps[gwrefAt + t] = p.gaussWref ? p.gaussWref[t] : 0.0;
    for (int t = threadIdx.x; t < 12 * KE; t += blockDim.x) sFn[t] = p.faceNodes[t];
    __syncthreads();

    const unsigned lane = threadIdx.x & 63u, q = lane >> 4, j = lane & 15u;
    auto L = [&](int tile) -> double { return sOps[tile * 64 + static_cast<int>(lane)]; };

    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld, gplane = 48ll * fb * ld, cplane = 16ll * ncb * ld;
    const unsigned ld8 = static_cast<unsigned>(ld) * 8u, ld4 = static_cast<unsigned>(ld) * 4u;
    const unsigned planeB = static_cast<unsigned>(plane * 8), gplaneB = static_cast<unsigned>(gplane * 8),
                   cplaneB = static_cast<unsigned>(cplane * 8);
    const unsigned ncoef = 1u + (p.zx ? 1u : 0u) + (p.zy ? 1u : 0u) + (p.fcor ? 1u : 0u) + (p.cd ? 1u : 0u);
    const __amdgpu_buffer_rsrc_t rq = cplane_rsrc(p.qin, 4u * planeB),
                                 rold = cplane_rsrc(MODE == CMODE_LSERK ? p.res : (MODE == CMODE_COMBINE ? p.qbase : p.qin), 4u * planeB),
                                 rout = cplane_rsrc(MODE == CMODE_RHS ? p.rhs : p.qout, 4u * planeB),
                                 rgg = cplane_rsrc(p.gaussG, 3u * gplaneB);
    const __amdgpu_buffer_rsrc_t rcub = cplane_rsrc(p.cubG, 4u * cplaneB); // read on curved / mixed tiles only (< 4 GiB: checked at creation)
    const __amdgpu_buffer_rsrc_t rnodeP = cplane_rsrc(p.nodeP, static_cast<unsigned>(3 * KE * 4) * ld4),
                                 rcoef = cplane_rsrc(p.rJ, ncoef * planeB), // rJ [, zx, zy, fcor, cd]: planes of ONE allocation
                                 raff = cplane_rsrc(p.elAffine, 14u * ld8);
    const unsigned soZx = p.zx ? static_cast<unsigned>((p.zx - p.rJ) * 8) : 0u, soZy = p.zy ? static_cast<unsigned>((p.zy - p.rJ) * 8) : 0u,
                   soFc = p.fcor ? static_cast<unsigned>((p.fcor - p.rJ) * 8) : 0u, soCd = p.cd ? static_cast<unsigned>((p.cd - p.rJ) * 8) : 0u;
    const double g = p.g;
    const unsigned ntiles = (static_cast<unsigned>(p.K - p.kbegin) + 15u) / 16u; // tiles of the element range [kbegin, K)
    // Tiles of this wave. XCD x (workgroups b with b mod 8 = x share its L2) owns one contiguous eighth of the tiles; its waves
    // take them SIDE BY SIDE (wave w of W: tiles w, w + W, ...), so that what an XCD holds at any time is a compact patch of
    // the mesh and the neighbours' face nodes a tile gathers are rows a sibling wave is streaming (p.tileInterleave = 0: one
    // contiguous run of tiles per wave, as in the first form). Every wave of an XCD makes the same number of passes (the
    // lockstep form needs that of a workgroup's waves: a wave without a tile keeps the barriers and the chunk copies company).
    // p.tileOrder: the XCD's eighth of the LIST of tiles, which deals the tiles that are not all straight-sided (1.2-1.5 x the
    // work, and next to each other in the mesh: the elements along a curved wall) evenly to the eight XCDs.
    // (Not kept: positions handed out by a counter per XCD, so that the wave that loses the issue arbitration of its SIMD -- the
    // younger of the two: at N = 4 the first workgroup of a CU ran its 8 tiles in 433 k cycles, the second its 7 in 556 k --
    // takes fewer tiles. 256 waves drawing from one address: 0.14 -> 0.21 ms at N = 2, 0.195 -> 0.24 at N = 3, even at N >= 4.)
    unsigned pos, tileEnd, tileStep, passes;
    const int* order = nullptr;
    if (p.tileInterleave) {
        if (gridDim.x >= 8u) order = p.tileOrder;
        const unsigned nwg = gridDim.x, xcd = blockIdx.x % 8u, wgHere = nwg / 8u + (xcd < nwg % 8u ? 1u : 0u);
        const unsigned nx = nwg < 8u ? nwg : 8u; // XCDs that have a workgroup of this launch
        const unsigned t0 = static_cast<unsigned>((static_cast<unsigned long long>(ntiles) * xcd) / nx),
                       t1 = static_cast<unsigned>((static_cast<unsigned long long>(ntiles) * (xcd + 1u)) / nx);
        const unsigned wavesHere = wgHere * (blockDim.x >> 6), w = (blockIdx.x / 8u) * (blockDim.x >> 6) + (threadIdx.x >> 6);
        pos = t0 + w;
        tileEnd = t1;
        tileStep = wavesHere;
        passes = (t1 - t0 + wavesHere - 1u) / wavesHere;
    } else {
        curved_wave_tiles(ntiles, pos, tileEnd);
        const unsigned nwavesAll = gridDim.x * (blockDim.x >> 6);
        tileStep = 1u;
        passes = STREAM ? (ntiles + nwavesAll - 1u) / nwavesAll : (tileEnd > pos ? tileEnd - pos : 0u);
    }
    int phase = 0; // stream buffer that holds the chunk about to be used
#ifdef BDG_PHASE_CLOCK
    // profiling build only: cycles a wave spends in each phase of a tile, summed over its tiles (0: requests of the tile's
    // first round trip; 1: volume term (waits for them); 2: surface term; 3: sources; 4: mass products, update, stores)
    unsigned long long clk[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp = __builtin_readcyclecounter();
    clk[5] = stamp - entryClk; // operator image to LDS, descriptors, tile range
    clk[8] = entryReal;        // 8, 9: entry and exit of the wave on the chip-wide 100 MHz counter
    const unsigned long long clk0 = stamp, real0 = __builtin_amdgcn_s_memrealtime(); // 6, 7: the loop in shader cycles / in 100 MHz ticks
#define BDG_NT_STAMP(i)                                                       \
    {                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                    \
        const unsigned long long now = __builtin_readcyclecounter();          \
        clk[i] += now - stamp;                                                \
        stamp = now;                                                          \
        __builtin_amdgcn_sched_barrier(0);                                    \
    }
#else
#define BDG_NT_STAMP(i)
#endif

    // Two workgroups per CU: the waves of the second one (dispatched later: blockIdx >= the number of CUs) lose every issue
    // arbitration of their SIMD to the older wave (MI355X_MICROARCH.md, two waves per SIMD). p.prioMode gives them priority 1
    // over a part of each tile, so that the two halves of the launch run at about the same speed and leave together.
    const int prio = (!STREAM && blockIdx.x >= 256u && gridDim.x > 256u) ? p.prioMode : 0;
    if (prio == 3) __builtin_amdgcn_s_setprio(1);
    for (unsigned pass = 0; pass < passes; ++pass, pos += tileStep) {
        const bool act = pos < tileEnd;
        if (prio == 1) __builtin_amdgcn_s_setprio(1);
        if (prio == 2) __builtin_amdgcn_s_setprio(0);
        const unsigned tile = (order && act) ? static_cast<unsigned>(__builtin_amdgcn_readfirstlane(order[pos])) : pos;
        const unsigned kTrue = static_cast<unsigned>(p.kbegin) + tile * 16u + j, kLast = static_cast<unsigned>(p.K) - 1u;
        const bool live = act && kTrue <= kLast;
        const unsigned k = (act && kTrue <= kLast) ? kTrue : kLast; // padding lanes recompute the last element, store nothing
        const unsigned k8 = k * 8u, v8 = (q * static_cast<unsigned>(ld) + k) * 8u, v4 = v8 >> 1;
        auto nodeOff = [&](int t) -> unsigned { // vector offset of node row 4 t + q (out of range on the padding rows of the last k-step)
            if constexpr (Np % 4 != 0) {
                if (t == KV - 1) return (4 * (KV - 1) + static_cast<int>(q) < Np) ? v8 : 0xfffffff8u;
            }
            return v8;
        };

        // ---- requests of the tile's first round trip, all in flight together: own nodal state in operand layout (node
        //      m = 4 t + q, 0 on padding rows), the straight-element numbers (zeros on other elements: read unconditionally, so
        //      that they do not wait for the flag), the flags (bits 0..2: wall faces, bit 3: straight element), and the
        //      neighbours' nodes at my face nodes (face node i = 4 t2 + q of face f; rows beyond Nfp point at an own node, the
        //      matching column of GE is zero)
        double qB[4][KV];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < KV; ++t) qB[c][t] = cbld_f64(rq, nodeOff(t), static_cast<unsigned>(c) * planeB + static_cast<unsigned>(4 * t) * ld8);
        const int flags = p.faceFlags[k];
        int idxP[3][KE];
#pragma unroll
        for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int t2 = 0; t2 < KE; ++t2) idxP[f][t2] = cbld_i32(rnodeP, v4, static_cast<unsigned>(f * KE * 4 + 4 * t2) * ld4);
        double ea[5]; // W rx, W ry, W sx, W sy factors and 1 / J of a straight element (the faces' numbers ride with their gathers)
#pragma unroll
        for (int i = 0; i < 4; ++i) ea[i] = cbld_f64(raff, k8, static_cast<unsigned>(i) * ld8);
        ea[4] = cbld_f64(raff, k8, 13u * ld8);
        __builtin_amdgcn_sched_barrier(0);
        const bool affTile = __all((flags & 8) != 0); // straight-sided tile? (padding lanes repeat the last element)
        BDG_NT_STAMP(0)

        // the neighbour's values at my face nodes and my own (both in the operand layout of GE: the element's own rows were
        // requested a moment ago by this wave, the second request finds them in cache) and, for a straight element, the face's
        // nx, ny, W factor
        auto gather = [&](int f, double (&qP)[4][KE], double (&qM)[4][KE], double (&fa)[3]) {
#pragma unroll
            for (int t2 = 0; t2 < KE; ++t2) {
                const unsigned oP = static_cast<unsigned>(idxP[f][t2]) * 8u;
                const unsigned oM = (static_cast<unsigned>(sFn[f * KE * 4 + 4 * t2 + static_cast<int>(q)]) * static_cast<unsigned>(ld) + k) * 8u;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    qP[c][t2] = cbld_f64(rq, oP, static_cast<unsigned>(c) * planeB);
                    qM[c][t2] = cbld_f64(rq, oM, static_cast<unsigned>(c) * planeB);
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) fa[i] = cbld_f64(raff, k8, static_cast<unsigned>(4 + 3 * f + i) * ld8);
        };

        // The tile's work, compiled twice: AFF = every element of the tile is straight-sided (geometry from the 14 numbers and
        // the reference weights in LDS) or not (geometry from the planes). One wave-uniform branch per tile instead of one per
        // cubature row / Gauss row / node: the bodies of the loops below are single basic blocks the scheduler can work in.
        auto body = [&](auto affC) {
            constexpr bool AFF = decltype(affC)::value;
            double qP[4][KE], qM[4][KE], fa[3];
            // face 0's nodes on both sides: the volume term hides their round trip -- up to order 6. Beyond, a tile is > 100 us of
            // products and the 28 registers are worth more than one exposed L2 round trip: requested after the volume term.
            constexpr bool kGatherEarly = KV <= 8;
            if constexpr (kGatherEarly) gather(0, qP, qM, fa);
            __builtin_amdgcn_sched_barrier(0);

            cmfma_t acc[4][MT];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < MT; ++r) acc[c][r] = cmfma_zero();

            // ---- volume term, 16 cubature points at a time
            for (int rb = 0; rb < ncb; ++rb) {
                int base; // LDS tile index of this chunk
                f64x2 pre[(VCH * 32 + 255) / 256];
                if constexpr (STREAM) {
                    base = phase * VCH;
                    // request the next chunk (this tile's rb + 1, or chunk 0 for the next pass) before the products
                    const int nextRb = rb + 1 < ncb ? rb + 1 : 0;
                    const f64x2* __restrict__ s2 = reinterpret_cast<const f64x2*>(p.opsNT + static_cast<size_t>(O::offVol(nextRb)) * 64);
#pragma unroll
                    for (int i = 0; i < (VCH * 32 + 255) / 256; ++i) {
                        const int at = static_cast<int>(threadIdx.x) + i * 256;
                        if (at < VCH * 32) pre[i] = s2[at];
                    }
                } else {
                    base = O::offVol(rb);
                }
                if (act) {
                    // A tiles are read from LDS one phase AHEAD of the products that use them (all KV interpolation tiles before the
                    // first product; the 2 MT DrT / DsT tiles of step reg + 1 before the pointwise work of step reg): left to
                    // itself the compiler reads a tile, waits for it and issues its 4..8 products, exposing the LDS latency every time.
                    double aV[KV], aD[2][2 * MT];
#pragma unroll
                    for (int t = 0; t < KV; ++t) aV[t] = L(base + t);
#pragma unroll
                    for (int r = 0; r < MT; ++r) { aD[0][r] = L(base + KV + r * 4); aD[0][MT + r] = L(base + KV + 4 * MT + r * 4); }
                    __builtin_amdgcn_sched_barrier(0);
                    cmfma_t cv[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) cv[c] = cmfma_zero();
#pragma unroll
                    for (int t = 0; t < KV; ++t) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) cv[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(aV[t], qB[c][t], cv[c], 0, 0, 0);
                    }
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) { // 4 cubature points of this lane = contraction step reg of DrT / DsT
                        const int row = 16 * rb + static_cast<int>(q) + 4 * reg;
                        const unsigned so = static_cast<unsigned>(16 * rb + 4 * reg) * ld8;
                        if (reg + 1 < 4) {
#pragma unroll
                            for (int r = 0; r < MT; ++r) {
                                aD[(reg + 1) & 1][r] = L(base + KV + r * 4 + reg + 1);
                                aD[(reg + 1) & 1][MT + r] = L(base + KV + 4 * MT + r * 4 + reg + 1);
                            }
                        }
                        double wrx, wry, wsx, wsy;
                        if constexpr (AFF) { // rule weight (times a reference Jacobian) of this point, the element's four numbers
                            const double w = sOps[wrefAt + row]; // zero on padding rows
                            wrx = w * ea[0]; wry = w * ea[1]; wsx = w * ea[2]; wsy = w * ea[3];
                        } else {
                            wrx = cbld_f64(rcub, v8, so); wry = cbld_f64(rcub, v8, cplaneB + so);
                            wsx = cbld_f64(rcub, v8, 2u * cplaneB + so); wsy = cbld_f64(rcub, v8, 3u * cplaneB + so);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        const bool valid = row < p.ncub;
                        const CurvedFlux fl = curved_fluxes(valid ? cv[0][reg] : 1.0, cv[1][reg], cv[2][reg], cv[3][reg], g);
                        double tr[4], ts[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            tr[c] = wrx * fl.F[c] + wry * fl.G[c];
                            ts[c] = wsx * fl.F[c] + wsy * fl.G[c];
                        }
#pragma unroll
                        for (int r = 0; r < MT; ++r) {
                            const double aDr = aD[reg & 1][r], aDs = aD[reg & 1][MT + r];
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                acc[c][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(aDr, tr[c], acc[c][r], 0, 0, 0);
                                acc[c][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(aDs, ts[c], acc[c][r], 0, 0, 0);
                            }
                        }
                    }
                }
                if constexpr (STREAM) {
                    f64x2* __restrict__ d2 = reinterpret_cast<f64x2*>(sOps + static_cast<size_t>((phase ^ 1) * VCH) * 64);
#pragma unroll
                    for (int i = 0; i < (VCH * 32 + 255) / 256; ++i) {
                        const int at = static_cast<int>(threadIdx.x) + i * 256;
                        if (at < VCH * 32) d2[at] = pre[i];
                    }
                    __syncthreads();
                    phase ^= 1;
                }
            }
            if (!act) return; // (lockstep form: nothing but barriers above for a wave without a tile)
            BDG_NT_STAMP(1)
            if (prio == 1) __builtin_amdgcn_s_setprio(0);
            if (prio == 2) __builtin_amdgcn_s_setprio(1);
            if constexpr (!kGatherEarly) gather(0, qP, qM, fa);

            // ---- what the last phase (mass inverse, sources, update) reads
            // Rows of this phase -- the source tables, 1 / J on general tiles, the residual / base state of the update -- are
            // requested TC k-steps (one field) ahead of their use: everything at once is 17 rows per k-step, 400 registers at N = 8.
            // Order: sources at every node first (S2, S3), then ONE field at a time: mass products (and Filter S for the momentum
            // fields), update, stores -- MT result tiles live instead of 4 MT.
            constexpr int TC = KV < 4 ? KV : 4, NCH = (KV + TC - 1) / TC;
            struct Rows { double cf[TC], cd[TC], zx[TC], zy[TC], rj[TC]; };
            auto requestRows = [&](int ch, Rows& w) {
#pragma unroll
                for (int i = 0; i < TC; ++i) {
                    const int t = ch * TC + i;
                    if (t >= KV) break;
                    const unsigned so = static_cast<unsigned>(4 * t) * ld8, vo = nodeOff(t);
                    if constexpr (!AFF) w.rj[i] = cbld_f64(rcoef, vo, so); // 0 on padding rows
                    w.cf[i] = p.fcor ? cbld_f64(rcoef, vo, soFc + so) : p.fconst;
                    w.cd[i] = p.cd ? cbld_f64(rcoef, vo, soCd + so) : p.cdconst;
                    w.zx[i] = p.zx ? cbld_f64(rcoef, vo, soZx + so) : 0.0;
                    w.zy[i] = p.zy ? cbld_f64(rcoef, vo, soZy + so) : 0.0;
                }
            };
            // At high order the update reads the element's own state again (an L2 hit, requested with the residual rows) instead
            // of holding all 4 KV operand registers to the end of the tile: the compiler kept them in scratch, and every reload
            // of a spilled register waits for ALL requests in flight. (The sources still read h, hu, hv from the operand
            // registers: reloading those too measured 6 % slower at N = 8.)
            constexpr bool REQO = KV > 8 && MODE != CMODE_RHS;
            auto requestOld = [&](int c, double (&o)[KV], double (&own)[REQO ? KV : 1]) {
                if constexpr (MODE != CMODE_RHS) {
#pragma unroll
                    for (int t = 0; t < KV; ++t) {
                        const unsigned so = static_cast<unsigned>(c) * planeB + static_cast<unsigned>(4 * t) * ld8;
                        o[t] = cbld_f64(rold, v8, so);
                        if constexpr (REQO) own[t] = cbld_f64(rq, v8, so);
                    }
                }
            };
            Rows rows[2];
            double oldv[2][KV], ownv[2][REQO ? KV : 1];
            // (Not kept: the first of these requests behind the last face's lift products, in the registers of the face nodes --
            // no gain at N = 2, 3, 5, 6, and 28 spilled registers at N = 4.)

            // ---- surface term, face by face; Gauss row of this lane: 16 b + q + 4 reg of the face
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                const bool wall = (flags >> f) & 1;
                const double fnx = fa[0], fny = fa[1], fw = fa[2];
                // One block per face: the exterior traces first, then the next face's nodes are requested while this face is
                // worked on. Two blocks: each block forms its own exterior traces (16 fewer live doubles) and the request follows
                // the last block's pointwise work, behind the face's lift products.
                // Both traces at a block's 16 Gauss rows come from the face's N + 1 nodes (the interpolation rows of a face are zero
                // off the face): KE k-steps per side with ONE A tile, instead of KV k-steps over the whole element for the own side.
                cmfma_t gP[4], gM[4];
                auto faceTraces = [&](int b) {
                    const int sbase = ldsSurf + (f * FB + b) * SCH;
                    double a[KE];
#pragma unroll
                    for (int t2 = 0; t2 < KE; ++t2) a[t2] = L(sbase + t2);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { gP[c] = cmfma_zero(); gM[c] = cmfma_zero(); }
#pragma unroll
                    for (int t2 = 0; t2 < KE; ++t2) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            gP[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t2], qP[c][t2], gP[c], 0, 0, 0);
                            gM[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t2], qM[c][t2], gM[c], 0, 0, 0);
                        }
                    }
                };
                if constexpr (FB == 1) faceTraces(0);
                double lam = 0.0;
                double ef[FB][4][4], dj[FB][4][4];
#pragma unroll
                for (int b = 0; b < FB; ++b) {
                    const int gb = f * FB + b;
                    if constexpr (FB > 1) faceTraces(b);
#pragma unroll
                    for (int reg = 0; reg < (b == FB - 1 ? RL : 4); ++reg) {
                        const int local = 16 * b + static_cast<int>(q) + 4 * reg;
                        const bool valid = local < p.ng;
                        double nx, ny, hW;
                        if constexpr (AFF) {
                            nx = fnx; ny = fny;
                            hW = sOps[gwrefAt + local] * fw; // gwref holds half the reference weights; zero on padding rows
                        } else {
                            const unsigned so8 = static_cast<unsigned>(16 * gb + 4 * reg) * ld8;
                            nx = cbld_f64(rgg, v8, so8); ny = cbld_f64(rgg, v8, gplaneB + so8);
                            hW = 0.5 * cbld_f64(rgg, v8, 2u * gplaneB + so8); // zero on padding rows
                        }
                        double hM = gM[0][reg], huM = gM[1][reg], hvM = gM[2][reg], hNM = gM[3][reg];
                        double hP = gP[0][reg], huP = gP[1][reg], hvP = gP[2][reg], hNP = gP[3][reg];
                        if (!valid) { hM = 1.0; hP = 1.0; huM = hvM = hNM = huP = hvP = hNP = 0.0; }
                        const double rM = crcp(hM), rP = crcp(hP);
                        // the wave speeds use the exterior velocity BEFORE the wall condition (rhs.py:81-85, :93-94)
                        const double uM = huM * rM, vM = hvM * rM, uP0 = huP * rP, vP0 = hvP * rP;
                        const double spdM = csqrt(uM * uM + vM * vM) + csqrt(g * hM);
                        const double spdP = csqrt(uP0 * uP0 + vP0 * vP0) + csqrt(g * hP);
                        lam = valid ? fmax(lam, fmax(spdM, spdP)) : lam;
                        if (wall) { // reflective wall (rhs.py:87-88)
                            const double un = huM * nx + hvM * ny;
                            huP = huM - 2 * nx * un;
                            hvP = hvM - 2 * ny * un;
                        }
                        const double uP = huP * rP, vP = hvP * rP;
                        const double prM = 0.5 * g * hM * hM, prP = 0.5 * g * hP * hP;
                        const double F[4] = {huM + huP, (huM * uM + prM) + (huP * uP + prP), hvM * uM + hvP * uP, hNM * uM + hNP * uP};
                        const double G[4] = {hvM + hvP, huM * vM + huP * vP, (hvM * vM + prM) + (hvP * vP + prP), hNM * vM + hNP * vP};
                        const double dq[4] = {hM - hP, huM - huP, hvM - hvP, hNM - hNP};
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            ef[b][reg][c] = hW * (F[c] * nx + G[c] * ny);
                            dj[b][reg][c] = hW * dq[c];
                        }
                        // one Gauss row at a time at high order: interleaved, the rows' temporaries do not fit beside 2 x 48 resident doubles
                        if constexpr (KV > 8) __builtin_amdgcn_sched_barrier(0);
                    }
                }
                lam = fmax(lam, __shfl_xor(lam, 16)); // the face's Gauss points sit in the 4 lanes q of this element
                lam = fmax(lam, __shfl_xor(lam, 32));
                if (f < 2) gather(f + 1, qP, qM, fa); // the next face's nodes: their round trip hides behind this face's lift products
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < FB; ++b)
#pragma unroll
                    for (int reg = 0; reg < (b == FB - 1 ? RL : 4); ++reg) {
                        double sf[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) sf[c] = fma(lam, dj[b][reg][c], ef[b][reg][c]);
                        const int sbase = ldsSurf + (f * FB + b) * SCH + KE;
#pragma unroll
                        for (int r = 0; r < MT; ++r) {
                            const double a = L(sbase + r * 4 + reg);
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc[c][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sf[c], acc[c][r], 0, 0, 0);
                        }
                    }
            }

            BDG_NT_STAMP(2)
            // ---- mass inverse, sources, update. acc[c][t >> 2][t & 3] is node m = 4 t + q: the operand layout again.
            const int slot = p.curvedSlot ? p.curvedSlot[k] : -1;
            requestRows(0, rows[0]);
            requestOld(0, oldv[0], ownv[0]);
            __builtin_amdgcn_sched_barrier(0);
            double S2[KV], S3[KV], rjn[AFF ? 1 : KV];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                if (ch + 1 < NCH) requestRows(ch + 1, rows[(ch + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const Rows& w = rows[ch & 1];
#pragma unroll
                for (int i = 0; i < TC; ++i) {
                    const int t = ch * TC + i;
                    if (t >= KV) break;
                    const int m = 4 * t + static_cast<int>(q);
                    if constexpr (!AFF) rjn[t] = w.rj[i];
                    // momentum sources at the node (rhs.py:165-174): S2 = f hv - CD |u| u - g h zx, S3 = -(f hu - CD |u| v) - g h zy
                    const double h = m < Np ? qB[0][t] : 1.0, hu = qB[1][t], hv = qB[2][t];
                    const double rh = crcp(h);
                    const double u = hu * rh, v = hv * rh;
                    const double cdn = w.cd[i] * csqrt(u * u + v * v);
                    S2[t] = m < Np ? (w.cf[i] * hv - cdn * u) - g * h * w.zx[i] : 0.0;
                    S3[t] = m < Np ? -(w.cf[i] * hu - cdn * v) - g * h * w.zy[i] : 0.0;
                }
            }
            BDG_NT_STAMP(3)
            if (live && slot >= 0) { // element of curvedEls: its own mass matrix is applied by the fix-up kernel
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int t = 0; t < KV; ++t) {
                        const int m = 4 * t + static_cast<int>(q);
                        if (m < Np) p.mmSide[(static_cast<size_t>(slot) * 4 + c) * Np + m] = acc[c][t >> 2][t & 3];
                    }
            }
            const bool store = live && slot < 0;
            const unsigned vo = store ? v8 : 0xfffffff8u; // (an out-of-range offset drops the store)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c + 1 < 4) requestOld(c + 1, oldv[(c + 1) & 1], ownv[(c + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                cmfma_t out[MT];
#pragma unroll
                for (int r = 0; r < MT; ++r) out[r] = cmfma_zero();
#pragma unroll
                for (int t = 0; t < KV; ++t) {
                    const int m = 4 * t + static_cast<int>(q);
                    double rj;
                    if constexpr (AFF) rj = m < Np ? ea[4] : 0.0;
                    else rj = rjn[t];
                    const double b = acc[c][t >> 2][t & 3] * rj;
#pragma unroll
                    for (int r = 0; r < MT; ++r) out[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(L(ldsMass + r * KV + t), b, out[r], 0, 0, 0);
                    if constexpr (FILTER) {
                        if (c == 1 || c == 2) {
                            const double sv = c == 1 ? S2[t] : S3[t];
#pragma unroll
                            for (int r = 0; r < MT; ++r) out[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(L(ldsF + r * KV + t), sv, out[r], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < KV; ++t) {
                    const int m = 4 * t + static_cast<int>(q);
                    if (m >= Np) continue;
                    double R = out[t >> 2][t & 3];
                    if constexpr (!FILTER) R += c == 1 ? S2[t] : (c == 2 ? S3[t] : 0.0);
                    const unsigned so = static_cast<unsigned>(c) * planeB + static_cast<unsigned>(4 * t) * ld8;
                    double own = 0.0;
                    if constexpr (REQO) own = ownv[c & 1][t];
                    else if constexpr (MODE != CMODE_RHS) own = qB[c][t];
                    if constexpr (MODE == CMODE_RHS) {
                        cbst_f64(rout, vo, so, R);
                    } else if constexpr (MODE == CMODE_LSERK) {
                        const double n1 = p.ca * oldv[c & 1][t] + p.cc * R;
                        cbst_f64(rold, vo, so, n1); // the residual, in place
                        cbst_f64(rout, vo, so, own + p.cb * n1);
                    } else {
                        cbst_f64(rout, vo, so, p.ca * oldv[c & 1][t] + p.cb * own + p.cc * R);
                    }
                }
            }
        };
        if (affTile) body(std::true_type{});
        else body(std::false_type{});
        BDG_NT_STAMP(4)
    }
#ifdef BDG_PHASE_CLOCK
    clk[6] = __builtin_readcyclecounter() - clk0;
    clk[7] = __builtin_amdgcn_s_memrealtime() - real0;
    clk[9] = __builtin_amdgcn_s_memrealtime();
    if (p.phaseClock != nullptr && lane == 0 && blockIdx.x < 1024u) {
#pragma unroll
        for (int i = 0; i < 12; ++i) p.phaseClock[(blockIdx.x * 4u + (threadIdx.x >> 6)) * 12u + i] = clk[i];
    }
#endif
#undef BDG_NT_STAMP
}

} // namespace bdg_dev
