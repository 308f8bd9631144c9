// sw2d_mfma3_kernel.hpp -- third schedule of the matrix-core stage kernel (variant A, three fields, straight-sided
// elements): every element's state is read from HBM ONCE per stage.
//
// The two-waves-per-SIMD schedule (sw2d_stage_mfma2_kernel) touches the state three times -- volume chunks, interior
// face traces, stage update -- and with 256 waves per XCD each holding a 17 KB tile (N=8) the second and third touch
// miss the 4 MB L2: profiles/r01_n8_pmc_summary.json shows 1.43 GB of reads per launch against 0.59 GB compulsory.
// Here a wave (one per SIMD, 512 registers) keeps its tile of 16 elements in registers in MFMA operand layout
// (node m = 4 t + q, which is both the B-operand layout of the volume term and the C/D layout of the result, so the
// stage update needs no second load), parks a copy in a wave-private LDS tile from which the interior face traces
// are read back (a face node lives in another lane's registers), and hides memory latency by software pipelining
// instead of by a second wave: the next tile's state and gather indices, this tile's neighbour traces and residual
// are all requested well before their first use, in batches pinned with scheduling barriers.
//   per tile:  LDS copy  ->  volume k-step t  [pointwise work of one face node; next tile: state row t, traces of a
//              finished face; this tile: residual rows]  ->  [next tile: geometry]  ->  for each block of 16 output rows:
//              the faces' matrix instructions, then the block's update (own state back from the LDS copy) + stores
// A register that held a state row (a face's neighbour traces) is free once that k-step's (face's) operands exist, so
// the next tile's copy of it is requested right there and nothing of a tile is waited for at its start.
//
// What the schedule has to respect on gfx950 (measured: profiles/microbench/mfma_valu_overlap.hip, profiles/README.md):
//  * v_mfma_f64_16x16x4_f64 occupies the SIMD's vector ALU for its 64 cycles: no vector instruction of this wave or of
//    any other wave on the SIMD runs beside it, so there is nothing to gain from interleaving vector work with matrix
//    instructions; what counts is the total of the two and that memory instructions keep flowing between them.
//  * A wave has at most 63 vector-memory instructions in flight (vmcnt) and a tile issues about 190 (N=8): requests are
//    spread over the tile -- a burst (all residual rows at once, all 6 KV stores at once) stalls on acknowledgements.
//  * A spilled register is reloaded with a scratch load followed by s_waitcnt vmcnt(0), which drains every prefetch in
//    flight: the straight-sided forms are kept free of spills at every order (tests/test_isa_hazards.py).
// Same operator image (MfmaOps2) and the same arithmetic as sw2d_stage_mfma2_kernel<N, MODE, 0>.
#pragma once
#include "sw2d_mfma_kernel.hpp"

namespace bdg_dev {

// Row accesses go through buffer instructions: a wave-uniform descriptor per plane, ONE per-lane byte offset per tile
// ((q ld + k) 8: lane (q, j) always touches rows 4 t + q of its element k) and the row group 4 t as a scalar offset.
// No per-load 64-bit address lives in vector registers (72 row loads in flight would need 144 of them). The hardware
// drops an access whose vector + scalar offset reaches the descriptor's size (profiles/microbench/buffer_range_check.hip;
// no 32-bit wrap-around in the sum), which already takes out the padding nodes m >= Np of the last k-step of a plane that
// has its own descriptor; those lanes also get an out-of-range vector offset (row_voffset), so the same holds where one
// descriptor spans several planes: their loads return 0, their stores are dropped, nothing beyond a plane is touched.
typedef unsigned int bdg_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ double bld_f64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ int bld_i32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return static_cast<int>(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void bst_f64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(bdg_u32x2, v), r, voff, soff, 0);
}
// write-through (sc1 = cache-policy bit 4: buffer_store_dwordx2 ... sc1): the store leaves the XCD's L2 for memory by itself --
// what a tile that hands its result to a concurrently running kernel uses (sync_signal_wave, sw2d_kernels.hpp)
__device__ __forceinline__ void bst_f64_wt(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(bdg_u32x2, v), r, voff, soff, 16);
}

// Zero accumulator the compiler cannot see through. With a literal 0 as the C operand of the first matrix instruction of
// a chain, hipcc is free to allocate a fresh destination for it, and under register pressure it has picked one that
// overlaps the A operand (v_mfma_f64_16x16x4_f64 a[16:23], a[22:23], v[..], 0 in the strip kernel below at N = 8). The
// multi-pass instruction reads its A operand while it writes its destination: sixteen result rows came out wrong. An
// accumulator that is already live is updated in place (dst = C) and cannot overlap the A / B operands.
// tests/test_isa_hazards.py scans the generated code for such overlaps.
__device__ __forceinline__ mfma_acc_t mfma_zero() {
    mfma_acc_t z{0.0, 0.0, 0.0, 0.0};
    asm volatile("" : "+v"(z));
    return z;
}

// Vector offset of row 4 t + q for the lane whose in-range offset is v8: lanes whose row is a padding row get an offset
// beyond every plane (planes are < 4 GiB, checked at creation).
template <int Np, int KV>
__device__ __forceinline__ unsigned row_voffset(int t, unsigned q, unsigned v8) {
    if constexpr (Np % 4 != 0) {
        if (t == KV - 1) return (4 * (KV - 1) + static_cast<int>(q) < Np) ? v8 : 0xfffffff8u;
    }
    return v8;
}

template <int N>
struct Mfma3Lds {
    using O = MfmaOps2<N>;
    static constexpr int TILE_DOUBLES = 3 * Elem<N>::Np * 16;          // one wave's state tile [field][node][element]
    static constexpr int DOUBLES = O::DOUBLES + 4 * TILE_DOUBLES;      // operator image + four waves' tiles
};

// HALO (MODE_LSERK, partition-boundary launches): the ghost exchange's pack and unpack folded in, as in
// sw2d_stage_mfma_kernel -- a neighbour trace that lives in a ghost slot is read from the received record, and the new
// state of an element is also written to its (up to three) send records. Same arithmetic as HALO = false, so a
// partitioned run reproduces the single-domain run bit for bit.
// NODAL: per-node geometry (callers whose tables are not those of straight-sided elements: rx, sx, ry, sy at every node
// in StageParams::geo, nx, ny, Fscale at every face node in StageParams::fgeo) instead of the 13 per-element values.
// With metric terms that vary inside an element the reference's form is R_c[i] = -(rx_i (Dr F_c)_i + sx_i (Ds F_c)_i +
// ry_i (Dr G_c)_i + sy_i (Ds G_c)_i) -- the metric multiplies AFTER the differentiation, at the output node -- so the
// volume term keeps ten products apart (Dr and Ds of the five distinct flux functions hu, hv, F2, G2, G3: 10 MT
// accumulators, 10 MT KV matrix instructions) and combines them with the output nodes' metric rows, requested during the
// last k-steps, before the faces are added. A face's normals and scales are requested for the next tile when the face
// is finished, like its neighbour traces.
// NFILT (NODAL only): the result is Filter * RHS. With the metric applied after the differentiation the filter cannot be
// folded into the operators (Filter (rx o Dr F) != rx o (Filter Dr) F): the image carries MT KV Filter tiles behind the
// plain operators, and the finished RHS -- accumulator layout = operand layout -- goes through one more product.
// SYNC (interior launches of a partitioned stage, StageParams::syncWait ...): the tiles from syncFirstTile on -- the ring of
// elements next to the partition boundary, ordered last -- are not requested, read or written before the previous stage's
// boundary launch has signalled, and each signals when its stores are visible device-wide (sw2d_kernels.hpp: sync_wait,
// sync_signal_wave). Everything of a tile is requested a tile ahead, so a wave checks before it starts the tile in front of its
// first ring tile.
template <int N, int MODE, bool HALO = false, bool NODAL = false, bool NFILT = false, bool SYNC = false>
__global__ __launch_bounds__(256, 1) void sw2d_stage_mfma3_kernel(const StageParams p) {
    static_assert(!HALO || MODE == MODE_LSERK, "halo staging is folded into LSERK stages only");
    static_assert(!SYNC || (MODE == MODE_LSERK && !HALO && !NODAL), "in-kernel stage dependencies: interior launches of LSERK stages");
    static_assert(!(HALO && NODAL), "partitioned runs use the straight-sided form");
    static_assert(NODAL || !NFILT, "straight-sided elements take the filter through pre-multiplied operators");
    using E = Elem<N>;
    using O = MfmaOps2<N>;
    constexpr int Np = E::Np, Nfp = E::Nfp, MT = O::MT, KV = O::KV, KF = O::KF;

    constexpr int IMAGE = O::DOUBLES + (NFILT ? MT * KV * 64 : 0); // [+ Filter tiles (r, t)]
    // Per-node geometry and the halo form at N = 8: nothing of the NEXT tile is requested ahead (its state, traces, indices
    // and face geometry are 230 registers; with them these forms spilled 125-170 registers, and a spilled register's reload
    // drains the memory pipeline every time: 0.96 ms per stage at C5 with per-node geometry, 0.41 ms without the prefetch).
    // The tile's own data is requested at its top instead: one exposed round trip per tile.
    constexpr bool PF = !((NODAL || HALO) && MT >= 3 && KV >= 12);
    extern __shared__ double sOps[];
    stage_image<IMAGE, 256>(sOps, p.opsAffine);
    __syncthreads();
    // this wave's state tile; indexed through sOps so that the accesses stay LDS instructions (a generic pointer
    // would turn them into flat_load / flat_store, which also wait for every outstanding global load)
    const int sBase = IMAGE + static_cast<int>(threadIdx.x >> 6) * Mfma3Lds<N>::TILE_DOUBLES + static_cast<int>(threadIdx.x & 15u);

    const unsigned lane = threadIdx.x & 63u, q = lane >> 4, j = lane & 15u;
    const unsigned nwg = gridDim.x, xcd = blockIdx.x % 8u, q8 = nwg / 8u, r8 = nwg % 8u;
    const unsigned blk = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + blockIdx.x / 8u;
    const unsigned wave = blk * 4u + (threadIdx.x >> 6), nwaves = nwg * 4u;
    const unsigned ntiles = (static_cast<unsigned>(p.kend - p.kbegin) + 15u) / 16u;
    const unsigned perWave = (ntiles + nwaves - 1u) / nwaves;
    unsigned tileEnd = min(ntiles, (wave + 1u) * perWave), tileStep = 1u;
    unsigned tile = wave * perWave;
    if (p.tileInterleave) {
        // the waves of one XCD walk its share of the tiles side by side: at any time the XCD works on one compact
        // patch of the mesh, whose interior face neighbours are in its L2 because a sibling wave is reading them
        // (its share of the tiles is proportional to its share of the workgroups: none if it has none)
        const unsigned wgHere = (nwg + 7u - xcd) / 8u, wgBefore = blk - blockIdx.x / 8u;
        tile = ntiles * wgBefore / nwg + (blockIdx.x / 8u) * 4u + (threadIdx.x >> 6);
        tileEnd = ntiles * (wgBefore + wgHere) / nwg;
        tileStep = wgHere * 4u;
    }
    if (tile >= tileEnd) return;

    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld;
    const double g = p.g, halfg = 0.5 * p.g;
    const unsigned kLast = static_cast<unsigned>(p.kend) - 1u;
    const unsigned planeBytes = static_cast<unsigned>(plane * 8), ld8 = static_cast<unsigned>(ld) * 8u, ld4 = static_cast<unsigned>(ld) * 4u;
    __amdgpu_buffer_rsrc_t rq[3], rold[3], rout[3], rres[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        rq[c] = plane_rsrc(p.qin + c * plane, planeBytes);
        rold[c] = plane_rsrc((MODE == MODE_LSERK ? p.res : (MODE == MODE_COMBINE ? p.qbase : p.qin)) + c * plane, planeBytes);
        rout[c] = plane_rsrc((MODE == MODE_RHS ? p.rhs : p.qout) + c * plane, planeBytes);
        rres[c] = plane_rsrc((MODE == MODE_LSERK ? p.res : p.qin) + c * plane, planeBytes);
    }
    const __amdgpu_buffer_rsrc_t rgeo = plane_rsrc(NODAL ? p.geo : p.ageo, NODAL ? planeBytes : 13u * ld8),
                                 ridx = plane_rsrc(p.vmapP, 3u * Nfp * ld4);
    // nodal geometry: rx, sx, ry, sy planes of Np rows; nx, ny, Fscale planes of 3 Nfp rows
    __amdgpu_buffer_rsrc_t rmet[4], rfg[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) rmet[i] = plane_rsrc(NODAL ? p.geo + i * plane : p.ageo, NODAL ? planeBytes : 8u);
#pragma unroll
    for (int i = 0; i < 3; ++i)
        rfg[i] = plane_rsrc(NODAL ? p.fgeo + static_cast<long long>(i) * 3 * Nfp * ld : p.ageo, NODAL ? 3u * Nfp * ld8 : 8u);
    // received ghost records [ghost][field][node] (HALO); the descriptor is never used otherwise
    const __amdgpu_buffer_rsrc_t rrecv = plane_rsrc(HALO ? p.haloRecv : p.ageo, 0xffffffffu);

    // element of this lane in tile `tl` (padding lanes recompute the last element and store nothing)
    auto elementOf = [&](unsigned tl, bool& live) {
        const unsigned kTrue = static_cast<unsigned>(p.kbegin) + tl * 16u + j;
        live = kTrue <= kLast;
        return live ? kTrue : kLast;
    };
    // Row 4 t + q of element kk sits at byte offset (q ld + kk) 8 + t (4 ld 8). Loaded values are used as they come:
    // any fix-up of padding rows happens where a value is consumed, so that a load never has to be waited for early.
    auto loadStateRow = [&](unsigned kk, int t, double (&qs)[3][KV]) {
        const unsigned v8 = (q * static_cast<unsigned>(ld) + kk) * 8u;
#pragma unroll
        for (int c = 0; c < 3; ++c) qs[c][t] = bld_f64(rq[c], row_voffset<Np, KV>(t, q, v8), static_cast<unsigned>(4 * t) * ld8); // padding rows: 0
    };
    auto loadIndices = [&](unsigned kk, int (&ix)[3][KF]) {
        const unsigned v4 = (q * static_cast<unsigned>(ld) + kk) * 4u;
#pragma unroll
        for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int tf = 0; tf < KF; ++tf) // lanes beyond the face's nodes: out-of-range offset, value 0
                ix[f][tf] = bld_i32(ridx, (4 * tf + static_cast<int>(q) < Nfp) ? v4 : 0xfffffffcu, static_cast<unsigned>(f * Nfp + 4 * tf) * ld4);
    };
    auto loadGeometry = [&](unsigned kk, double (&gg)[13]) {
        if constexpr (!NODAL) {
#pragma unroll
            for (int i = 0; i < 13; ++i) gg[i] = bld_f64(rgeo, kk * 8u, static_cast<unsigned>(i) * ld8);
        }
    };
    auto loadMetricRow = [&](unsigned kk, int t, double (&mm)[4]) { // rx, sx, ry, sy at node 4 t + q
        const unsigned v8 = (q * static_cast<unsigned>(ld) + kk) * 8u;
#pragma unroll
        for (int i = 0; i < 4; ++i) mm[i] = bld_f64(rmet[i], row_voffset<Np, KV>(t, q, v8), static_cast<unsigned>(4 * t) * ld8);
    };
    auto loadFaceGeometry = [&](int f, unsigned kk, double (&fx)[3][KF], double (&fy)[3][KF], double (&fs)[3][KF]) {
        const unsigned v8 = (q * static_cast<unsigned>(ld) + kk) * 8u;
#pragma unroll
        for (int tf = 0; tf < KF; ++tf) {
            const unsigned vo = (4 * tf + static_cast<int>(q) < Nfp) ? v8 : 0xfffffff8u;
            const unsigned so = static_cast<unsigned>(f * Nfp + 4 * tf) * ld8;
            fx[f][tf] = bld_f64(rfg[0], vo, so);
            fy[f][tf] = bld_f64(rfg[1], vo, so);
            fs[f][tf] = bld_f64(rfg[2], vo, so);
        }
    };
    // neighbour traces of face f (face node n = 4 tf + q); lanes beyond the face read node 0 of element 0 and ignore it
    auto loadTraces = [&](int f, const int (&ix)[3][KF], double (&a)[3][KF], double (&b)[3][KF], double (&c3)[3][KF]) {
#pragma unroll
        for (int tf = 0; tf < KF; ++tf) {
            const int n = 4 * tf + static_cast<int>(q);
            const int id = n < Nfp ? ix[f][tf] : 0;
            const unsigned idp = static_cast<unsigned>(id < 0 ? -(id + 1) : id), o8 = idp * 8u;
            if constexpr (HALO) {
                const unsigned row = idp / static_cast<unsigned>(ld), slot = idp - row * static_cast<unsigned>(ld);
                if (slot >= static_cast<unsigned>(p.haloOwned)) { // the neighbour's record as it arrived: [field][node]
                    const unsigned rec8 = ((slot - static_cast<unsigned>(p.haloOwned)) * static_cast<unsigned>(p.haloRows) + row) * 8u;
                    a[f][tf] = bld_f64(rrecv, rec8, 0u);
                    b[f][tf] = bld_f64(rrecv, rec8, static_cast<unsigned>(Np) * 8u);
                    c3[f][tf] = bld_f64(rrecv, rec8, static_cast<unsigned>(2 * Np) * 8u);
                    continue;
                }
            }
            a[f][tf] = bld_f64(rq[0], o8, 0u);
            b[f][tf] = bld_f64(rq[1], o8, 0u);
            c3[f][tf] = bld_f64(rq[2], o8, 0u);
        }
    };

#ifdef BDG_PHASE_CLOCK
    // profiling build only (-DBDG_PHASE_CLOCK): cycles a wave spends in each phase of a tile, summed over its tiles
    unsigned long long phase[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp = 0;
#define BDG_STAMP(i)                                                          \
    {                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                    \
        const unsigned long long now = __builtin_readcyclecounter();          \
        phase[i] += now - stamp;                                              \
        stamp = now;                                                          \
        __builtin_amdgcn_sched_barrier(0);                                    \
    }
#else
#define BDG_STAMP(i)
#endif
    if (p.stagger > 0) { // workgroups start out of phase: (blk mod 8) * stagger kilocycles apart
        const unsigned naps = (blk & 7u) * static_cast<unsigned>(p.stagger);
        for (unsigned i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(16);
    }
    bool synced = false;
    auto waitBefore = [&](unsigned upcoming) { // `upcoming`: a tile of this wave whose data is about to be requested
        if constexpr (SYNC) {
            if (!synced && upcoming < tileEnd && upcoming >= static_cast<unsigned>(p.syncFirstTile)) {
                sync_wait(p.syncWait, p.syncWaitValue, p.syncError);
                synced = true;
            }
        }
    };
    waitBefore(tile);
    bool live;
    unsigned k = elementOf(tile, live);
    double qB[3][KV], geo[13], hP[3][KF], huP[3][KF], hvP[3][KF];
    int fidx[3][KF];
    double fnx[3][KF], fny[3][KF], fsc[3][KF];
    auto loadTile = [&](unsigned kk) { // everything a tile needs before its first product
        loadIndices(kk, fidx);
#pragma unroll
        for (int t = 0; t < KV; ++t) loadStateRow(kk, t, qB);
        loadGeometry(kk, geo);
#pragma unroll
        for (int f = 0; f < 3; ++f) loadTraces(f, fidx, hP, huP, hvP);
        if constexpr (NODAL) { // all face normals / scales
#pragma unroll
            for (int f = 0; f < 3; ++f) loadFaceGeometry(f, kk, fnx, fny, fsc);
        }
    };
    if constexpr (PF) loadTile(k); // the first tile; the following ones are requested piece by piece a tile ahead

#ifdef BDG_PHASE_CLOCK
    stamp = __builtin_readcyclecounter();
#endif
#pragma unroll 1
    for (;;) {
        const unsigned v8 = (q * static_cast<unsigned>(ld) + k) * 8u;
        const bool more = tile + tileStep < tileEnd;
        waitBefore(tile + tileStep); // the next tile is requested during this one
        bool liveN = false;
        const unsigned kN = more ? elementOf(tile + tileStep, liveN) : k;
        if constexpr (!PF) loadTile(k);
        // next tile, requested piece by piece below as this tile's registers fall free
        double qN[3][KV], geoN[13], hPN[3][KF], huPN[3][KF], hvPN[3][KF];
        int fidxN[3][KF];
        double fnxN[3][KF], fnyN[3][KF], fscN[3][KF];

        // ---- own state into the wave's LDS tile (face traces and the update read it back from there)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int t = 0; t < KV; ++t) {
                const int m = 4 * t + static_cast<int>(q);
                if (m < Np) sOps[sBase + (c * Np + m) * 16] = qB[c][t];
            }
        __builtin_amdgcn_wave_barrier(); // other lanes of this wave read these values back (DS operations of a wave execute in order)

        BDG_STAMP(0) // LDS copy
        // residual (LSERK) / base state (COMBINE) rows of THIS tile: requested during the first volume k-steps, consumed
        // by the update of their row block
        double oldv[3][KV];
        auto loadOldRow = [&](int t) {
            if constexpr (MODE != MODE_RHS) {
#pragma unroll
                for (int c = 0; c < 3; ++c) oldv[c][t] = bld_f64(rold[c], row_voffset<Np, KV>(t, q, v8), static_cast<unsigned>(4 * t) * ld8);
            }
        };
        mfma_acc_t acc[3][MT];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < MT; ++r) acc[c][r] = mfma_zero(); // not a literal 0: see mfma_zero

        // volume term: the six operands of k-step t (vector ALU), then its 6 MT matrix instructions; the state rows of
        // k-step t are dead by then and the same rows of the NEXT tile are requested in their place
        auto volumeOperands = [&](int t, double (&ab)[6]) {
            const int m = 4 * t + static_cast<int>(q);
            const bool pad = m >= Np;
            const double h = pad ? 1.0 : qB[0][t], hu = qB[1][t], hv = qB[2][t];
            const double r = fast_rcp(h);
            const double u = hu * r, v = hv * r;
            const double pr = halfg * h * h;
            const double F2 = hu * u + pr, G2 = hu * v, G3 = hv * v + pr;
            const double w = pad ? 0.0 : -1.0; // zero the padded rows of the operand
            if constexpr (NODAL) { // the five distinct flux functions themselves (metric terms come after the products)
                ab[0] = pad ? 0.0 : hu; ab[1] = pad ? 0.0 : hv; ab[2] = pad ? 0.0 : F2; ab[3] = pad ? 0.0 : G2;
                ab[4] = pad ? 0.0 : G3; ab[5] = 0.0;
            } else {
                const double rx = geo[0], sx = geo[1], ry = geo[2], sy = geo[3];
                ab[0] = w * (rx * hu + ry * hv); ab[1] = w * (sx * hu + sy * hv);
                ab[2] = w * (rx * F2 + ry * G2); ab[3] = w * (sx * F2 + sy * G2);
                ab[4] = w * (rx * G2 + ry * G3); ab[5] = w * (sx * G2 + sy * G3);
            }
        };
        // Pointwise work of the surface term (face node n = 4 tf + q of face f; '-' traces from the LDS tile, '+' traces
        // prefetched during the previous tile), one face node per lane at a time: it is spread over the volume k-steps so
        // that a face's neighbour-trace registers fall free early and the next tile's gathers go out one face at a time.
        // A face's operands s_c = Fscale/2 (e_c - lam d_c) are final once its last node is in.
        double sF[3][3][KF];
        double eF[3][KF], dF[3][KF], lamF = 0.0;
        auto faceNode = [&](int f, int tf) {
            const double nxf = NODAL ? fnx[f][tf] : geo[4 + f], nyf = NODAL ? fny[f][tf] : geo[7 + f];
            const int n = 4 * tf + static_cast<int>(q);
#pragma unroll
            for (int c = 0; c < 3; ++c) eF[c][tf] = dF[c][tf] = 0.0;
            if (n < Nfp) {
                const int m = fmask_rt<N>(f, n);
                const double hM = sOps[sBase + m * 16], huM = sOps[sBase + (Np + m) * 16], hvM = sOps[sBase + (2 * Np + m) * 16];
                const double hq = hP[f][tf];
                double huq = huP[f][tf], hvq = hvP[f][tf];
                if (fidx[f][tf] < 0) { // reflective wall: no normal flow
                    const double un = huM * nxf + hvM * nyf;
                    huq = huM - 2 * nxf * un;
                    hvq = hvM - 2 * nyf * un;
                }
                const double rM = fast_rcp(hM), rP = fast_rcp(hq);
                const double uM = huM * rM, vM = hvM * rM, uP = huq * rP, vP = hvq * rP;
                const double spdM = fast_sqrt(uM * uM + vM * vM) + fast_sqrt(g * hM);
                const double spdP = fast_sqrt(uP * uP + vP * vP) + fast_sqrt(g * hq);
                lamF = fmax(lamF, fmax(spdM, spdP));
                const double prM = halfg * hM * hM, prP = halfg * hq * hq;
                const double F2M = huM * uM + prM, G2M = huM * vM, G3M = hvM * vM + prM;
                const double F2P = huq * uP + prP, G2P = huq * vP, G3P = hvq * vP + prP;
                dF[0][tf] = hM - hq; dF[1][tf] = huM - huq; dF[2][tf] = hvM - hvq;
                eF[0][tf] = dF[1][tf] * nxf + dF[2][tf] * nyf;
                eF[1][tf] = (F2M - F2P) * nxf + (G2M - G2P) * nyf;
                eF[2][tf] = (G2M - G2P) * nxf + (G3M - G3P) * nyf;
            }
            if (tf == KF - 1) { // the face is complete: its speed (the face's nodes sit in the 4 lanes q of this element)
                double lam = fmax(lamF, __shfl_xor(lamF, 16));
                lam = fmax(lam, __shfl_xor(lam, 32));
#pragma unroll
                for (int t2 = 0; t2 < KF; ++t2) {
                    const double hfs = 0.5 * (NODAL ? fsc[f][t2] : geo[10 + f]);
#pragma unroll
                    for (int c = 0; c < 3; ++c) sF[f][c][t2] = hfs * (eF[c][t2] - lam * dF[c][t2]);
                }
                lamF = 0.0;
                // this face's '+' traces are dead: request the next tile's (requesting all three faces in the last k-steps
                // instead, so that they never wait for their indices, was slower: too many requests at once)
                if constexpr (PF) loadTraces(f, fidxN, hPN, huPN, hvPN);
                if constexpr (NODAL && PF) loadFaceGeometry(f, kN, fnxN, fnyN, fscN);
            }
        };
        constexpr int FACE_ITEMS = 3 * KF, PER_STEP = (FACE_ITEMS + KV - 1) / KV;

        if constexpr (!NODAL) {
            loadIndices(kN, fidxN);
            __builtin_amdgcn_sched_barrier(0);
            // residual rows: blocks 0 .. MT - 2 during the last k-steps (one row per step), the last block at the start of
            // the surface term -- each a good microsecond before its update, and no earlier (registers)
            constexpr int OLD_EARLY = HALO ? 0 : ((MT > 1 ? 4 * (MT - 1) : 0) < KV ? (MT > 1 ? 4 * (MT - 1) : 0) : KV); // (HALO: registers)
#pragma unroll
            for (int t = 0; t < KV; ++t) {
                double ab[6];
                volumeOperands(t, ab);
                loadStateRow(kN, t, qN);
                if (t >= KV - OLD_EARLY) loadOldRow(t - (KV - OLD_EARLY));
#pragma unroll
                for (int it = t * PER_STEP; it < (t + 1) * PER_STEP; ++it)
                    if (it < FACE_ITEMS) faceNode(it / KF, it % KF);
#pragma unroll
                for (int r2 = 0; r2 < MT; ++r2) {
                    const double Adr = sOps[O::OFF_DR + (r2 * KV + t) * 64 + lane];
                    const double Ads = sOps[O::OFF_DS + (r2 * KV + t) * 64 + lane];
                    acc[0][r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Adr, ab[0], acc[0][r2], 0, 0, 0);
                    acc[1][r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Adr, ab[2], acc[1][r2], 0, 0, 0);
                    acc[2][r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Adr, ab[4], acc[2][r2], 0, 0, 0);
                    acc[0][r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ads, ab[1], acc[0][r2], 0, 0, 0);
                    acc[1][r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ads, ab[3], acc[1][r2], 0, 0, 0);
                    acc[2][r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ads, ab[5], acc[2][r2], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                BDG_STAMP(1 + (t < 12 ? t : 11)) // k-step t
            }
#pragma unroll
            for (int t = OLD_EARLY; t < KV; ++t) loadOldRow(t);
        } else {
            // One 16-row block of output nodes at a time: av[2 i] = Dr s_i, av[2 i + 1] = Ds s_i for the five flux functions
            // s = (hu, hv, F2, G2, G3) (their values are formed again in every pass: a few dozen vector instructions against
            // 10 matrix instructions per k-step), then the block's rows R_c = -(rx Dr F_c + sx Ds F_c + ry Dr G_c + sy Ds G_c)
            // with the metric rows of the block's own nodes. The next tile's state is requested during the last pass.
            if constexpr (PF) loadIndices(kN, fidxN);
            constexpr int STEPS = MT * KV, PER = (FACE_ITEMS + STEPS - 1) / STEPS;
#pragma unroll
            for (int r = 0; r < MT; ++r) {
                double met[4][4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * r + e < KV) loadMetricRow(k, 4 * r + e, met[e]);
                mfma_acc_t av[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) av[i] = mfma_zero();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < KV; ++t) {
                    double fl[6];
                    volumeOperands(t, fl);
                    if (PF && r == MT - 1) loadStateRow(kN, t, qN);
#pragma unroll
                    for (int it = (r * KV + t) * PER; it < (r * KV + t + 1) * PER; ++it)
                        if (it < FACE_ITEMS) faceNode(it / KF, it % KF);
                    const double Adr = sOps[O::OFF_DR + (r * KV + t) * 64 + lane];
                    const double Ads = sOps[O::OFF_DS + (r * KV + t) * 64 + lane];
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        av[2 * i] = __builtin_amdgcn_mfma_f64_16x16x4f64(Adr, fl[i], av[2 * i], 0, 0, 0);
                        av[2 * i + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ads, fl[i], av[2 * i + 1], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (4 * r + e < KV) {
                        const double rx = met[e][0], sx = met[e][1], ry = met[e][2], sy = met[e][3]; // 0 on padding rows
                        // F = (hu, F2, G2), G = (hv, G2, G3); av: 0,1 hu | 2,3 hv | 4,5 F2 | 6,7 G2 | 8,9 G3
                        acc[0][r][e] = -(rx * av[0][e] + sx * av[1][e] + ry * av[2][e] + sy * av[3][e]);
                        acc[1][r][e] = -(rx * av[4][e] + sx * av[5][e] + ry * av[6][e] + sy * av[7][e]);
                        acc[2][r][e] = -(rx * av[6][e] + sx * av[7][e] + ry * av[8][e] + sy * av[9][e]);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < KV; ++t) loadOldRow(t); // (registers: not before the last pass is over)
        }

        BDG_STAMP(13) // volume term (rest)
        if constexpr (PF) loadGeometry(kN, geoN);
        __builtin_amdgcn_sched_barrier(0);

        // ---- stage update / output of the rows of block r: node m = 4 t + q is accumulator row 16 (t >> 2) + q + 4 (t & 3);
        //      stores of the padding rows carry an out-of-range vector offset (row_voffset) and are dropped, padding lanes
        //      store nothing. The block's own state comes back from the LDS tile in one batch.
        int sendRec[3] = {-1, -1, -1};
        auto loadSendRecords = [&]() { // (HALO) right before the update: nothing of it lives across the surface term
            if constexpr (HALO) {
                if (live) {
                    const unsigned b3 = (k - static_cast<unsigned>(p.kbegin)) * 3u;
                    sendRec[0] = p.haloSendOf[b3];
                    sendRec[1] = p.haloSendOf[b3 + 1];
                    sendRec[2] = p.haloSendOf[b3 + 2];
                }
            }
        };
        auto updateBlock = [&](int r) {
            if (!live) return;
            double own[3][4];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = 4 * (4 * r + e) + static_cast<int>(q);
                    own[c][e] = (4 * r + e < KV && m < Np) ? sOps[sBase + (c * Np + m) * 16] : 0.0;
                }
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int t = 4 * r + e;
                    if (t >= KV) continue;
                    const int m = 4 * t + static_cast<int>(q);
                    const unsigned soff = static_cast<unsigned>(4 * t) * ld8;
                    const double R = acc[c][r][e];
                    if constexpr (MODE == MODE_RHS) {
                        bst_f64(rout[c], row_voffset<Np, KV>(t, q, v8), soff, R);
                    } else if constexpr (MODE == MODE_LSERK) {
                        const double n1 = p.ca * oldv[c][t] + p.cc * R;
                        const double qn = own[c][e] + p.cb * n1;
                        bst_f64(rres[c], row_voffset<Np, KV>(t, q, v8), soff, n1);
                        if constexpr (SYNC) { // a ring tile hands its new state to the boundary launch: write-through
                            if (tile >= static_cast<unsigned>(p.syncFirstTile)) bst_f64_wt(rout[c], row_voffset<Np, KV>(t, q, v8), soff, qn);
                            else bst_f64(rout[c], row_voffset<Np, KV>(t, q, v8), soff, qn);
                        } else {
                            bst_f64(rout[c], row_voffset<Np, KV>(t, q, v8), soff, qn);
                        }
                        if constexpr (HALO) {
                            if (m < Np) {
#pragma unroll
                                for (int sr = 0; sr < 3; ++sr)
                                    if (sendRec[sr] >= 0)
                                        p.haloSend[static_cast<size_t>(sendRec[sr]) * p.haloRows + c * Np + m] = qn;
                            }
                        }
                    } else {
                        const double val = p.ca * oldv[c][t] + p.cb * own[c][e] + p.cc * R;
                        bst_f64(rout[c], row_voffset<Np, KV>(t, q, v8), soff, c == 0 ? val : sponge_relax(val, p.sponge));
                    }
                }
        };

        // ---- surface term: matrix instructions only (operands formed above). Straight-sided single-domain form: one
        //      block of 16 output rows at a time, so that a block's rows are updated and stored while the next block's
        //      products run and the stores of a tile are spread over the phase instead of arriving as one burst of 6 KV (a
        //      wave has at most 63 memory instructions in flight: the burst stalls on the first ones' acknowledgements).
        //      The other forms have no registers for it (a face's operands stay live until the last block): face by face,
        //      update at the end.
        constexpr bool BLOCKWISE = !NODAL && !HALO;
        if constexpr (BLOCKWISE) {
#pragma unroll
            for (int r = 0; r < MT; ++r) {
#pragma unroll
                for (int f = 0; f < 3; ++f)
#pragma unroll
                    for (int tf = 0; tf < KF; ++tf) {
                        const double Al = sOps[O::OFF_LIFT + ((r * 3 + f) * KF + tf) * 64 + lane];
                        acc[0][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(Al, sF[f][0][tf], acc[0][r], 0, 0, 0);
                        acc[1][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(Al, sF[f][1][tf], acc[1][r], 0, 0, 0);
                        acc[2][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(Al, sF[f][2][tf], acc[2][r], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
                updateBlock(r);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int f = 0; f < 3; ++f)
#pragma unroll
                for (int tf = 0; tf < KF; ++tf)
#pragma unroll
                    for (int r = 0; r < MT; ++r) {
                        const double Al = sOps[O::OFF_LIFT + ((r * 3 + f) * KF + tf) * 64 + lane];
                        acc[0][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(Al, sF[f][0][tf], acc[0][r], 0, 0, 0);
                        acc[1][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(Al, sF[f][1][tf], acc[1][r], 0, 0, 0);
                        acc[2][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(Al, sF[f][2][tf], acc[2][r], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!NFILT) {
                loadSendRecords();
#pragma unroll
                for (int r = 0; r < MT; ++r) updateBlock(r);
            }
        }
        BDG_STAMP(14) // surface term

        if constexpr (NFILT) { // RHS <- Filter * RHS: the accumulators are the operands of one more product
            mfma_acc_t fo[3][MT];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int r = 0; r < MT; ++r) fo[c][r] = mfma_zero();
#pragma unroll
            for (int t = 0; t < KV; ++t)
#pragma unroll
                for (int r = 0; r < MT; ++r) {
                    const double Af = sOps[O::DOUBLES + (r * KV + t) * 64 + lane];
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        fo[c][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(Af, acc[c][t >> 2][t & 3], fo[c][r], 0, 0, 0);
                }
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int r = 0; r < MT; ++r) acc[c][r] = fo[c][r];
#pragma unroll
            for (int r = 0; r < MT; ++r) updateBlock(r);
        }
        BDG_STAMP(15) // update and stores
        if constexpr (SYNC) {
            if (tile >= static_cast<unsigned>(p.syncFirstTile)) sync_signal_wave(p.syncSignal);
        }
        if (!more) break;
        tile += tileStep;
        k = kN;
        live = liveN;
        if constexpr (PF) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int t = 0; t < KV; ++t) qB[c][t] = qN[c][t];
#pragma unroll
            for (int i = 0; i < 13; ++i) geo[i] = geoN[i];
#pragma unroll
            for (int f = 0; f < 3; ++f)
#pragma unroll
                for (int tf = 0; tf < KF; ++tf) {
                    fidx[f][tf] = fidxN[f][tf];
                    hP[f][tf] = hPN[f][tf]; huP[f][tf] = huPN[f][tf]; hvP[f][tf] = hvPN[f][tf];
                    if constexpr (NODAL) { fnx[f][tf] = fnxN[f][tf]; fny[f][tf] = fnyN[f][tf]; fsc[f][tf] = fscN[f][tf]; }
                }
        }

    }
#ifdef BDG_PHASE_CLOCK
    if (p.phaseClock != nullptr && lane == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) p.phaseClock[(blockIdx.x * 4u + (threadIdx.x >> 6)) * 16u + i] = phase[i];
    }
#endif
#undef BDG_STAMP
}

} // namespace bdg_dev

namespace bdg_dev {

// ---------------------------------------------------------------------------------------------
// Partition-boundary strip of an exchanged LSERK4 stage (a few hundred elements between the received ghosts and the
// records to send): latency, not throughput, is what counts -- the strip kernel sits on the exchange chain
// (receive -> strip -> send) that every stage has to get through, and a tile's instruction stream on ONE wave takes
// 40-60 us (N=8: profiles/r02_rehearsal_n8.txt) whatever the schedule. Here a tile of 16 elements is shared by the
// three waves of a workgroup, one conserved field each: every wave forms the operands of its own field and issues a
// third of the matrix instructions, in the same order as sw2d_stage_mfma3_kernel (volume k-steps, then faces), and
// does its own field's stage update, halo staging included (ghost traces from the received records, new state to the
// send records). The wave speed of a face needs all three fields, so traces are loaded by all three waves (L1 hits).
// SYNC: the launch waits at its top for the previous interior launch's ring tiles (the boundary elements' interior
// neighbours, whose new state it reads and whose old state it overwrites) and signals once per workgroup at its end.
template <int N, bool SYNC = false>
__global__ __launch_bounds__(192, 2) void sw2d_strip_mfma3_kernel(const StageParams p) {
    using E = Elem<N>;
    using O = MfmaOps2<N>;
    constexpr int Np = E::Np, Nfp = E::Nfp, MT = O::MT, KV = O::KV, KF = O::KF;

    extern __shared__ double sOps[];
    bool staged = false; // the operator image is staged AFTER the first tile's loads are requested (below)
    if constexpr (SYNC) sync_wait(p.syncWait, p.syncWaitValue, p.syncError);

    const unsigned lane = threadIdx.x & 63u, q = lane >> 4, j = lane & 15u;
    const int c = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)); // field of this wave
    const unsigned ntiles = (static_cast<unsigned>(p.kend - p.kbegin) + 15u) / 16u;
    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld;
    const double g = p.g, halfg = 0.5 * p.g;
    const unsigned kLast = static_cast<unsigned>(p.kend) - 1u;
    const unsigned planeBytes = static_cast<unsigned>(plane * 8), ld8 = static_cast<unsigned>(ld) * 8u, ld4 = static_cast<unsigned>(ld) * 4u;
    __amdgpu_buffer_rsrc_t rq[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) rq[i] = plane_rsrc(p.qin + i * plane, planeBytes);
    const __amdgpu_buffer_rsrc_t rres = plane_rsrc(p.res + c * plane, planeBytes), rout = plane_rsrc(p.qout + c * plane, planeBytes);
    const __amdgpu_buffer_rsrc_t rgeo = plane_rsrc(p.ageo, 13u * ld8), ridx = plane_rsrc(p.vmapP, 3u * Nfp * ld4);
    const __amdgpu_buffer_rsrc_t rrecv = plane_rsrc(p.haloRecv, 0xffffffffu);

    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const unsigned kTrue = static_cast<unsigned>(p.kbegin) + tile * 16u + j;
        const bool live = kTrue <= kLast;
        const unsigned k = live ? kTrue : kLast;
        const unsigned k8 = k * 8u, v8 = (q * static_cast<unsigned>(ld) + k) * 8u, v4 = (q * static_cast<unsigned>(ld) + k) * 4u;

        // ---- everything this tile reads, requested up front (the gathers follow their indices)
        int fidx[3][KF];
#pragma unroll
        for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int tf = 0; tf < KF; ++tf)
                fidx[f][tf] = bld_i32(ridx, (4 * tf + static_cast<int>(q) < Nfp) ? v4 : 0xfffffffcu, static_cast<unsigned>(f * Nfp + 4 * tf) * ld4);
        double qB[3][KV], geo[13], oldv[KV];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int t = 0; t < KV; ++t) qB[i][t] = bld_f64(rq[i], row_voffset<Np, KV>(t, q, v8), static_cast<unsigned>(4 * t) * ld8);
#pragma unroll
        for (int i = 0; i < 13; ++i) geo[i] = bld_f64(rgeo, k8, static_cast<unsigned>(i) * ld8);
        auto requestResidual = [&] {
#pragma unroll
            for (int t = 0; t < KV; ++t) oldv[t] = bld_f64(rres, row_voffset<Np, KV>(t, q, v8), static_cast<unsigned>(4 * t) * ld8);
        };
        constexpr bool kLateFace = KV > 9; // N = 8: see requestFace
        if constexpr (!kLateFace) requestResidual();
        double hM[3][KF], huM[3][KF], hvM[3][KF], hP[3][KF], huP[3][KF], hvP[3][KF];
        // traces of one face: at N <= 7 all three faces are requested up front; at N = 8 (18 doubles per face) the third face
        // is requested when the first one has been worked on, which keeps the kernel within 256 registers: two workgroups
        // per CU, so that a strip of 24..47 tiles runs in ONE round on the CUs the interior launch leaves free
        auto requestFace = [&](int f) {
#pragma unroll
            for (int tf = 0; tf < KF; ++tf) {
                const int n = 4 * tf + static_cast<int>(q);
                const int m = n < Nfp ? fmask_rt<N>(f, n) : 0;
                const unsigned m8 = (static_cast<unsigned>(m) * static_cast<unsigned>(ld) + k) * 8u;
                hM[f][tf] = bld_f64(rq[0], m8, 0u);
                huM[f][tf] = bld_f64(rq[1], m8, 0u);
                hvM[f][tf] = bld_f64(rq[2], m8, 0u);
                const int id = n < Nfp ? fidx[f][tf] : 0;
                const unsigned idp = static_cast<unsigned>(id < 0 ? -(id + 1) : id), o8 = idp * 8u;
                const unsigned row = idp / static_cast<unsigned>(ld), slot = idp - row * static_cast<unsigned>(ld);
                if (slot >= static_cast<unsigned>(p.haloOwned)) { // the neighbour's record as it arrived: [field][node]
                    const unsigned rec8 = ((slot - static_cast<unsigned>(p.haloOwned)) * static_cast<unsigned>(p.haloRows) + row) * 8u;
                    hP[f][tf] = bld_f64(rrecv, rec8, 0u);
                    huP[f][tf] = bld_f64(rrecv, rec8, static_cast<unsigned>(Np) * 8u);
                    hvP[f][tf] = bld_f64(rrecv, rec8, static_cast<unsigned>(2 * Np) * 8u);
                } else {
                    hP[f][tf] = bld_f64(rq[0], o8, 0u);
                    huP[f][tf] = bld_f64(rq[1], o8, 0u);
                    hvP[f][tf] = bld_f64(rq[2], o8, 0u);
                }
            }
        };
        requestFace(0);
        if constexpr (!kLateFace) {
            requestFace(1);
            requestFace(2);
        }
        int sendRec[3] = {-1, -1, -1};
        if (live) {
            const unsigned b3 = (k - static_cast<unsigned>(p.kbegin)) * 3u;
            sendRec[0] = p.haloSendOf[b3];
            sendRec[1] = p.haloSendOf[b3 + 1];
            sendRec[2] = p.haloSendOf[b3 + 2];
        }
        if (!staged) { // (every workgroup has a first tile: the grid is the tile count; the flag is workgroup-uniform)
            stage_image<O::DOUBLES, 192>(sOps, p.opsAffine);
            __syncthreads();
            staged = true;
        }

        mfma_acc_t acc[MT];
#pragma unroll
        for (int r = 0; r < MT; ++r) acc[r] = mfma_zero(); // not a literal 0: see mfma_zero

        // ---- volume term: the two operands of this wave's field per k-step
        const double rx = geo[0], sx = geo[1], ry = geo[2], sy = geo[3];
#pragma unroll
        for (int t = 0; t < KV; ++t) {
            const int m = 4 * t + static_cast<int>(q);
            const bool pad = m >= Np;
            const double h = pad ? 1.0 : qB[0][t], hu = qB[1][t], hv = qB[2][t];
            const double r = fast_rcp(h);
            const double u = hu * r, v = hv * r;
            const double pr = halfg * h * h;
            const double F2 = hu * u + pr, G2 = hu * v, G3 = hv * v + pr;
            const double w = pad ? 0.0 : -1.0;
            const double Fc = c == 0 ? hu : (c == 1 ? F2 : G2), Gc = c == 0 ? hv : (c == 1 ? G2 : G3);
            const double a = w * (rx * Fc + ry * Gc), b = w * (sx * Fc + sy * Gc);
#pragma unroll
            for (int r2 = 0; r2 < MT; ++r2) {
                acc[r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(sOps[O::OFF_DR + (r2 * KV + t) * 64 + lane], a, acc[r2], 0, 0, 0);
                acc[r2] = __builtin_amdgcn_mfma_f64_16x16x4f64(sOps[O::OFF_DS + (r2 * KV + t) * 64 + lane], b, acc[r2], 0, 0, 0);
            }
        }

        if constexpr (kLateFace) { // the second face and the residual rows behind the first face's work (the other fields' state rows are dead by now)
            requestFace(1);
            requestResidual();
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- surface term
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            const double nxf = geo[4 + f], nyf = geo[7 + f], hfs = 0.5 * geo[10 + f];
            double e[KF], d[KF], lam = 0.0;
#pragma unroll
            for (int tf = 0; tf < KF; ++tf) {
                const int n = 4 * tf + static_cast<int>(q);
                e[tf] = d[tf] = 0.0;
                if (n < Nfp) {
                    const double hMv = hM[f][tf], huMv = huM[f][tf], hvMv = hvM[f][tf], hq = hP[f][tf];
                    double huq = huP[f][tf], hvq = hvP[f][tf];
                    if (fidx[f][tf] < 0) { // reflective wall: no normal flow
                        const double un = huMv * nxf + hvMv * nyf;
                        huq = huMv - 2 * nxf * un;
                        hvq = hvMv - 2 * nyf * un;
                    }
                    const double rM = fast_rcp(hMv), rP = fast_rcp(hq);
                    const double uM = huMv * rM, vM = hvMv * rM, uP = huq * rP, vP = hvq * rP;
                    const double spdM = fast_sqrt(uM * uM + vM * vM) + fast_sqrt(g * hMv);
                    const double spdP = fast_sqrt(uP * uP + vP * vP) + fast_sqrt(g * hq);
                    lam = fmax(lam, fmax(spdM, spdP));
                    const double prM = halfg * hMv * hMv, prP = halfg * hq * hq;
                    const double F2M = huMv * uM + prM, G2M = huMv * vM, G3M = hvMv * vM + prM;
                    const double F2P = huq * uP + prP, G2P = huq * vP, G3P = hvq * vP + prP;
                    const double d1 = hMv - hq, d2 = huMv - huq, d3 = hvMv - hvq;
                    d[tf] = c == 0 ? d1 : (c == 1 ? d2 : d3);
                    const double e1 = d2 * nxf + d3 * nyf;
                    const double e2 = (F2M - F2P) * nxf + (G2M - G2P) * nyf;
                    const double e3 = (G2M - G2P) * nxf + (G3M - G3P) * nyf;
                    e[tf] = c == 0 ? e1 : (c == 1 ? e2 : e3);
                }
            }
            lam = fmax(lam, __shfl_xor(lam, 16));
            lam = fmax(lam, __shfl_xor(lam, 32));
            if constexpr (kLateFace) {
                if (f == 0) {
                    requestFace(2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int tf = 0; tf < KF; ++tf) {
                const double s = hfs * (e[tf] - lam * d[tf]);
#pragma unroll
                for (int r = 0; r < MT; ++r)
                    acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(sOps[O::OFF_LIFT + ((r * 3 + f) * KF + tf) * 64 + lane], s, acc[r], 0, 0, 0);
            }
        }

        // ---- LSERK4 stage update of this wave's field, new state also to the element's send records
        if (live) {
#pragma unroll
            for (int t = 0; t < KV; ++t) {
                const int m = 4 * t + static_cast<int>(q);
                const unsigned soff = static_cast<unsigned>(4 * t) * ld8;
                const double own = c == 0 ? qB[0][t] : (c == 1 ? qB[1][t] : qB[2][t]);
                const double n1 = p.ca * oldv[t] + p.cc * acc[t >> 2][t & 3];
                const double qn = own + p.cb * n1;
                bst_f64(rres, row_voffset<Np, KV>(t, q, v8), soff, n1);
                if constexpr (SYNC) bst_f64_wt(rout, row_voffset<Np, KV>(t, q, v8), soff, qn); // read by the interior launch's ring tiles
                else bst_f64(rout, row_voffset<Np, KV>(t, q, v8), soff, qn);
                if (m < Np) {
#pragma unroll
                    for (int sr = 0; sr < 3; ++sr)
                        if (sendRec[sr] >= 0) p.haloSend[static_cast<size_t>(sendRec[sr]) * p.haloRows + c * Np + m] = qn;
                }
            }
        }
    }
    if constexpr (SYNC) sync_signal_workgroup(p.syncSignal);
}

} // namespace bdg_dev
