// sw2d_curved_kernel.hpp -- curved / over-integrated shallow-water RHS on the matrix cores.
//
// Reference: swhelpers.rhs.sw2dComputeRHS_curved (swhelpers/rhs.py:6-176) as driven by sw2d_curved.py:246-277.
// Per element, with q = (h, hu, hv, hN) at the Np nodes:
//     cub_q   = Vc q                                     (Ncub x Np) interpolation to the cubature points
//     MM_c    = DrT (W (rx F_c + ry G_c)) + DsT (W (sx F_c + sy G_c))           fluxes at the cubature points
//     MM_c   -= InterpT (gW * 1/2 ((F_cM + F_cP) nx + (G_cM + G_cP) ny + lam (q_cM - q_cP)))   at the Gauss face points
//     RHS_c   = (V V^T) (MM_c / J)    straight-sided elements;    U^-1 U^-T MM_c    elements listed in curvedEls
//     RHS_2,3 += Coriolis, drag, bed slope at the nodes
// Every one of these is a dense contraction over 4..200 points, batched over the elements: for a tile of 16
// elements it is a chain of v_mfma_f64_16x16x4_f64 products whose B operands and C results never leave
// registers. Lane l = (q = l >> 4, j = l & 15) works on element j of the tile; as a B operand it supplies
// contraction index q of a 4-deep step, and a C result hands it rows q, q+4, q+8, q+12 of a 16-row block -- which
// are again "index q of step reg" of the next product, so C of one product is fed straight back as B of the next
// (cubature values -> weighted fluxes -> DrT/DsT; Gauss fluxes -> InterpT; MM / J -> mass inverse), the A tiles
// being stored with their columns in that order. The pointwise physics runs in operand layout with no cross-lane
// traffic; only the per-face maximum of the wave speed crosses lanes (two xor shuffles per face).
//
// Launches per RHS evaluation:
//   sw2d_curved_gauss_kernel   q -> Gauss traces gq (Interp q) of every element, (4, GR, ld) planes
//   sw2d_curved_stage_kernel   everything above + RK update; neighbour traces gathered from gq through gmapP
//                              (any map: periodic rewiring included); elements of curvedEls leave their raw MM_c
//                              in a side buffer instead
//   sw2d_curved_fixup_kernel   the elements of curvedEls: their own inverse mass matrix (U^-1 U^-T of their Cholesky factor,
//                              formed at creation) applied as one dense product, sources, filter, update
// Tables are (rows, ld) planes, element index contiguous; a wave touches 128-byte row segments.
#pragma once
#include <hip/hip_runtime.h>

namespace bdg_dev {

typedef double cmfma_t __attribute__((ext_vector_type(4)));

enum { CMODE_RHS = 0, CMODE_LSERK = 1, CMODE_COMBINE = 2 };

// Zero accumulator the compiler cannot see through (a literal 0 as the C operand of a chain's first matrix instruction lets
// hipcc allocate a destination that overlaps the A operand, which the multi-pass instruction is still reading: see
// mfma_zero in sw2d_mfma3_kernel.hpp and tests/test_isa_hazards.py).
__device__ __forceinline__ cmfma_t cmfma_zero() {
    cmfma_t z{0.0, 0.0, 0.0, 0.0};
    asm volatile("" : "+v"(z));
    return z;
}

struct CurvedParams {
    const double* qin;    // 4 planes of Np*ld: h, hu, hv, hN
    const double* qbase;  // CMODE_COMBINE
    double* qout;         // CMODE_LSERK / CMODE_COMBINE (may alias qin or qbase: only own elements are read from them)
    double* res;          // CMODE_LSERK
    double* rhs;          // CMODE_RHS
    double* gq;           // Gauss traces, 4 planes of GR*ld (GR = 48*fb rows: every face padded to 16*fb rows)
    const double* cubG;   // 4 planes of CR*ld (CR = 16*ncb, zero padded): W rx, W ry, W sx, W sy
    const double* gaussG; // 3 planes of GR*ld (zero padded): nx, ny, W
    const int* gmapP;     // GR*ld: offset row*ld + k into a gq plane; wall Gauss nodes stored as -(offset+1)
    const int* gmapM;     // the same for the interior side, or nullptr when gmapM is the identity
    const double* rJ;     // Np*ld: 1 / J at the nodes
    const double* zx;     // Np*ld or nullptr
    const double* zy;
    const double* fcor;   // Np*ld or nullptr (then fconst)
    const double* cd;     // Np*ld or nullptr (then cdconst)
    double fconst, cdconst;
    const int* curvedSlot; // ld: index into the side buffer for elements of curvedEls, -1 otherwise (nullptr: none)
    double* mmSide;        // (numCurved, 4, Np): raw MM_c of the elements of curvedEls
    const double* minvSide; // (numCurved, Np, Np): their inverse mass matrices U^-1 U^-T (from cub_ctx.MMChol)
    const int* curvedEls;  // element slot of each side-buffer column
    int numCurved;          // columns of the side buffer the fix-up launch works on
    const int* slotList;    // nullptr: columns 0 .. numCurved - 1; else column slotList[i] (partitioned runs: the curved elements of a range)
    long long sideLd;
    const int* affineEl;    // ld: 1 for straight-sided elements whose cubature geometry is held compressed (nullptr: none)
    const double* cubAffine; // 4 rows of ld: the element's numbers c with (W rx, W ry, W sx, W sy)[i] = cubWref[i] * c
    const double* cubWref;   // CR: W at the cubature points of a reference straight element (zero padded)
    const double* ops;    // operator image (CurvedOps layout), 64 doubles per tile
    // nodal-trace form (sw2d_curved_nt_kernel.hpp); nullptr / unused in the first form
    const double* opsNT;     // operator image in CurvedOpsNT layout
    const int* nodeP;        // (3 * KE * 4, ld): offset row * ld + k of the neighbour's node at my face node (face f, node i) at row f * KE * 4 + i
    const int* faceFlags;    // ld: bit f set = face f is a wall; bit 3: straight element (elAffine holds its numbers)
    const int* faceNodes;    // 3 x 4 KE: node of face node i of face f (Fmask as Interp shows it; padding entries repeat the face's first node)
    int prioMode;            // nodal-trace kernel, two workgroups per CU: see the tile loop (0: every wave at priority 0)
    unsigned long long* phaseClock; // profiling builds (-DBDG_PHASE_CLOCK): 12 counts per wave of the nodal-trace kernel, else unused
    const double* elAffine;  // (14, ld), zeros on other elements: straight elements: W rx, W ry, W sx, W sy factors; nx, ny, W factor per face; 1 / J
    const double* gaussWref; // 16 fb: HALF the Gauss weights of a reference straight face (zero padded)
    int tileInterleave;      // 1: the waves of an XCD take its tiles side by side (default); 0: a contiguous run per wave
    const int* tileOrder;    // (ntiles) or nullptr: list position -> tile, every XCD's eighth of the list holding an eighth of the
                             // general (not all straight-sided) tiles: they cost more, and meshes keep them together
    const double* filt;   // (Np, Np) row-major filter for the fix-up kernel, or nullptr
    long long ld;
    int K;                // elements [kbegin, K) are worked on by the nodal-trace launch (kbegin = 0 in the first form)
    int kbegin;
    int gridReserve;      // nodal-trace launch: workgroup slots of the chip left free (for the launch that runs beside it)
    int ncb;              // 16-row blocks of cubature points
    int ncub;             // cubature points
    int ng;               // Gauss points per face
    int fb;               // 16-row blocks per face
    double g, ca, cb, cc;
};

// Operator image: zero-padded 16 x 4 A tiles, 64 doubles each, entry l of a tile = A[row l & 15][step column l >> 4].
//   Vc   [rb][t]       row 16 rb + i = cubature point,      column 4 t + s = node
//   DrT  [r][rb][reg]  row 16 r + i = node,                 column s <-> cubature point 16 rb + 4 reg + s   (Dr_cub^T)
//   DsT  likewise
//   IT   [r][gb][reg]  row 16 r + i = node,                 column s <-> Gauss row 16 b + 4 reg + s of face gb / fb   (-Interp^T)
//   M    [r][t]        row 16 r + i = node,                 column 4 t + s = node       V V^T
//   MF   [r][t]        Filter V V^T
//   F    [r][t]        Filter
//   GI   [gb][t]       row 16 b + i = Gauss row of a face,  column 4 t + s = node       Interp (Gauss kernel)
template <int N>
struct CurvedOps {
    static constexpr int Np = (N + 1) * (N + 2) / 2;
    static constexpr int KV = (Np + 3) / 4;
    static constexpr int MT = (Np + 15) / 16;
    __host__ __device__ static constexpr int offVc(int, int) { return 0; }
    __host__ __device__ static constexpr int offDrT(int ncb, int) { return ncb * KV; }
    __host__ __device__ static constexpr int offDsT(int ncb, int fb) { return offDrT(ncb, fb) + MT * ncb * 4; }
    __host__ __device__ static constexpr int offIT(int ncb, int fb) { return offDsT(ncb, fb) + MT * ncb * 4; }
    __host__ __device__ static constexpr int offM(int ncb, int fb) { return offIT(ncb, fb) + MT * 3 * fb * 4; }
    __host__ __device__ static constexpr int offMF(int ncb, int fb) { return offM(ncb, fb) + MT * KV; }
    __host__ __device__ static constexpr int offF(int ncb, int fb) { return offMF(ncb, fb) + MT * KV; }
    __host__ __device__ static constexpr int offGI(int ncb, int fb) { return offF(ncb, fb) + MT * KV; }
    __host__ __device__ static constexpr int tiles(int ncb, int fb) { return offGI(ncb, fb) + 3 * fb * KV; }
    // the stage kernel reads the whole image (the Gauss tiles GI for the element's own traces)
};

template <typename T>
__device__ __forceinline__ T cld_row(const T* row, unsigned byteOff) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(row) + byteOff);
}
template <typename T>
__device__ __forceinline__ void cst_row(T* row, unsigned byteOff, T v) {
    *reinterpret_cast<T*>(reinterpret_cast<char*>(row) + byteOff) = v;
}

// Tile range of this wave: XCD-aware (workgroups b and b + 8 share an XCD's L2), contiguous tiles per wave.
__device__ __forceinline__ void curved_wave_tiles(unsigned ntiles, unsigned& first, unsigned& last) {
    const unsigned nwg = gridDim.x, xcd = blockIdx.x % 8u, q8 = nwg / 8u, r8 = nwg % 8u;
    const unsigned blk = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + blockIdx.x / 8u;
    const unsigned wave = blk * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = nwg * (blockDim.x >> 6);
    const unsigned per = (ntiles + nwaves - 1u) / nwaves;
    first = wave * per;
    last = min(ntiles, first + per);
}

// ---- Gauss traces of every element: gq_c = Interp q_c, faces padded to 16*fb rows (padding rows are never written)
template <int N>
__global__ __launch_bounds__(256) void sw2d_curved_gauss_kernel(const CurvedParams p) {
    using O = CurvedOps<N>;
    constexpr int Np = O::Np, KV = O::KV;
    extern __shared__ double sOps[];
    const int fb = p.fb, ngb = 3 * fb, offGI = O::offGI(p.ncb, fb);
    for (int t = threadIdx.x; t < ngb * KV * 64; t += blockDim.x) sOps[t] = p.ops[static_cast<size_t>(offGI) * 64 + t];
    __syncthreads();

    const unsigned lane = threadIdx.x & 63u, q = lane >> 4, j = lane & 15u;
    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld, gplane = 48ll * fb * ld;
    const unsigned ntiles = (static_cast<unsigned>(p.K) + 15u) / 16u;
    unsigned tile, tileEnd;
    curved_wave_tiles(ntiles, tile, tileEnd);
    for (; tile < tileEnd; ++tile) {
        const unsigned kTrue = tile * 16u + j, kLast = static_cast<unsigned>(p.K) - 1u;
        const bool live = kTrue <= kLast;
        const unsigned k8 = (live ? kTrue : kLast) * 8u;
        double qB[4][KV];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < KV; ++t) {
                const int m = 4 * t + static_cast<int>(q);
                qB[c][t] = m < Np ? cld_row(p.qin + c * plane + m * ld, k8) : 0.0;
            }
        for (int gb = 0; gb < ngb; ++gb) {
            cmfma_t gv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) gv[c] = cmfma_zero();
#pragma unroll
            for (int t = 0; t < KV; ++t) {
                const double a = sOps[(gb * KV + t) * 64 + lane];
#pragma unroll
                for (int c = 0; c < 4; ++c) gv[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, qB[c][t], gv[c], 0, 0, 0);
            }
            const int b = gb % fb;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int local = 16 * b + static_cast<int>(q) + 4 * reg, gr = 16 * gb + static_cast<int>(q) + 4 * reg;
                if (live && local < p.ng) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) cst_row(p.gq + c * gplane + gr * ld, k8, gv[c][reg]);
                }
            }
        }
    }
}

// 1/x and sqrt(x) to ~1 ulp without the denormal / infinity fix-up paths of the IEEE sequences (water depth and
// speeds are well scaled): v_rcp_f64 / v_rsq_f64 + Newton steps, as in the straight-element kernels.
__device__ __forceinline__ double crcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double csqrt(double x) {
    x = fmax(x, 1e-290);
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    g = fma(fma(-g, g, x), h, g); // 0.5 ulp already: a second correction changes nothing (profiles/r02_rcp_rsq_accuracy.txt)
    return g;
}

// Shallow-water fluxes of one point (reference swhelpers/flux.py:1-22); u = hu * (1/h) formed once.
struct CurvedFlux {
    double F[4], G[4];
};
__device__ __forceinline__ CurvedFlux curved_fluxes(double h, double hu, double hv, double hN, double g) {
    const double rh = crcp(h);
    const double u = hu * rh, v = hv * rh;
    const double pr = 0.5 * g * h * h;
    CurvedFlux o;
    o.F[0] = hu;          o.G[0] = hv;
    o.F[1] = hu * u + pr; o.G[1] = hu * v;
    o.F[2] = hv * u;      o.G[2] = hv * v + pr;
    o.F[3] = hN * u;      o.G[3] = hN * v;
    return o;
}

// Momentum sources at a node (rhs.py:165-174): S2 = f hv - CD |u| u - g h zx,  S3 = -(f hu - CD |u| v) - g h zy.
__device__ __forceinline__ void curved_sources(const CurvedParams& p, double h, double hu, double hv, long long rowOff,
                                               unsigned k8, double& S2, double& S3) {
    const double rh = crcp(h);
    const double u = hu * rh, v = hv * rh;
    const double f = p.fcor ? cld_row(p.fcor + rowOff, k8) : p.fconst;
    const double cd = p.cd ? cld_row(p.cd + rowOff, k8) : p.cdconst;
    const double cdn = cd * csqrt(u * u + v * v);
    const double zx = p.zx ? cld_row(p.zx + rowOff, k8) : 0.0, zy = p.zy ? cld_row(p.zy + rowOff, k8) : 0.0;
    S2 = (f * hv - cdn * u) - p.g * h * zx;
    S3 = -(f * hu - cdn * v) - p.g * h * zy;
}

// Buffer addressing (as in sw2d_mfma3_kernel.hpp): wave-uniform descriptors, ONE per-lane byte offset per tile
// ((q ld + k) 8: lane (q, j) touches rows 4 t + q / 16 b + 4 reg + q of its element k) and the plane and row group as scalar
// offset, so no 64-bit address per load sits in vector registers. An access is dropped (loads return 0) when vector +
// scalar offset reaches the descriptor's size: lanes on padding rows get a vector offset of 0xfffffff8 for that.
typedef unsigned int cbdg_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cplane_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ double cbld_f64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ int cbld_i32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return static_cast<int>(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void cbst_f64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(cbdg_u32x2, v), r, voff, soff, 0);
}

// MODE: CMODE_*; FILTER: the result is Filter * RHS (the drivers filter the whole RHS, sources included);
// OPSLDS: operator tiles staged in LDS (else read from global memory / L2: images beyond the LDS budget);
// FB: 16-row blocks per face (1: NGauss <= 16, 2: NGauss <= 32); WAVES: waves per SIMD the register budget is set for.
//
// Two savings of HBM traffic on top of the first version:
//  * straight-sided elements (CurvedParams::affineEl; decided per tile with one ballot): W rx, W ry, W sx, W sy at the Ncub
//    points are 4 numbers per element times the rule's weights, so the four Ncub-row planes are not read for them;
//  * the element's own Gauss traces are formed here by the tile products of the Gauss kernel (same instructions, same
//    operands: bit-identical to what the neighbours read from gq) instead of being read back.
// MAPM: gmapM is not the identity (the interior traces are gathered through it like the exterior ones).
template <int N, int MODE, bool FILTER, bool OPSLDS, int FB, int WAVES, bool MAPM = false>
__global__ __launch_bounds__(256, WAVES) void sw2d_curved_stage_kernel(const CurvedParams p) {
    using O = CurvedOps<N>;
    constexpr int Np = O::Np, KV = O::KV, MT = O::MT;
    extern __shared__ double sOps[];
    const int ncb = p.ncb;
    constexpr int fb = FB;
    const int offDrT = O::offDrT(ncb, fb), offDsT = O::offDsT(ncb, fb), offIT = O::offIT(ncb, fb),
              offMass = FILTER ? O::offMF(ncb, fb) : O::offM(ncb, fb), offF = O::offF(ncb, fb), offGI = O::offGI(ncb, fb);
    // reference weights of the straight elements behind the image (a global read per cubature row would be a round trip
    // to L1 each, sixteen of them per tile in a row)
    const int wrefAt = OPSLDS ? O::tiles(ncb, fb) * 64 : 0;
    if constexpr (OPSLDS) {
        const int n = O::tiles(ncb, fb) * 64;
        // eight 16-byte pieces per thread in flight at a time (a plain copy loop is load, wait, store per iteration: 25 round
        // trips to L2 for a 50 KB image at the start of every workgroup)
        typedef double cbdg_f64x2 __attribute__((ext_vector_type(2)));
        const cbdg_f64x2* __restrict__ s2 = reinterpret_cast<const cbdg_f64x2*>(p.ops);
        cbdg_f64x2* __restrict__ d2 = reinterpret_cast<cbdg_f64x2*>(sOps);
        const int pairs = n / 2, nthreads = static_cast<int>(blockDim.x);
        for (int b = threadIdx.x; b < pairs; b += 8 * nthreads) {
            cbdg_f64x2 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (b + i * nthreads < pairs) v[i] = s2[b + i * nthreads];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (b + i * nthreads < pairs) d2[b + i * nthreads] = v[i];
        }
    }
    for (int t = threadIdx.x; t < 16 * ncb; t += blockDim.x) sOps[wrefAt + t] = p.cubWref ? p.cubWref[t] : 0.0;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u, q = lane >> 4, j = lane & 15u;
    auto A = [&](int tile) -> double {
        if constexpr (OPSLDS) return sOps[tile * 64 + static_cast<int>(lane)];
        else return p.ops[static_cast<size_t>(tile) * 64 + lane];
    };

    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld, gplane = 48ll * fb * ld,
                    cplane = 16ll * ncb * ld;
    const unsigned ld8 = static_cast<unsigned>(ld) * 8u, ld4 = static_cast<unsigned>(ld) * 4u;
    const unsigned planeB = static_cast<unsigned>(plane * 8), gplaneB = static_cast<unsigned>(gplane * 8),
                   cplaneB = static_cast<unsigned>(cplane * 8);
    // One descriptor per ARRAY (all its planes; the plane -- field c, table i -- is picked by the scalar offset). 35 per-plane
    // descriptors (140 scalar registers) were spilled to vector-register lanes and read back, four v_readlane per use. The
    // range check compares vector + scalar offset with the array's size (profiles/microbench/buffer_range_check.hip), so
    // an array stays below 4 GiB (checked at creation) and the padding rows of a plane are taken out by nodeOff below.
    // The cubature geometry keeps a descriptor per plane: its planes are the big ones.
    const unsigned ncoef = 1u + (p.zx ? 1u : 0u) + (p.zy ? 1u : 0u) + (p.fcor ? 1u : 0u) + (p.cd ? 1u : 0u);
    const __amdgpu_buffer_rsrc_t rq = cplane_rsrc(p.qin, 4u * planeB),
                                 rold = cplane_rsrc(MODE == CMODE_LSERK ? p.res : (MODE == CMODE_COMBINE ? p.qbase : p.qin), 4u * planeB),
                                 rout = cplane_rsrc(MODE == CMODE_RHS ? p.rhs : p.qout, 4u * planeB),
                                 rgq = cplane_rsrc(p.gq, 4u * gplaneB), rgg = cplane_rsrc(p.gaussG, 3u * gplaneB);
    __amdgpu_buffer_rsrc_t rcub[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) rcub[c] = cplane_rsrc(p.cubG + c * cplane, cplaneB);
    const __amdgpu_buffer_rsrc_t rmapP = cplane_rsrc(p.gmapP, static_cast<unsigned>(gplane * 4)),
                                 rmapM = cplane_rsrc(p.gmapM ? p.gmapM : p.gmapP, static_cast<unsigned>(gplane * 4)),
                                 rcoef = cplane_rsrc(p.rJ, ncoef * planeB), // rJ [, zx, zy, fcor, cd]: planes of ONE allocation
                                 raff = cplane_rsrc(p.cubAffine ? p.cubAffine : p.rJ, 4u * ld8);
    const unsigned soZx = p.zx ? static_cast<unsigned>((p.zx - p.rJ) * 8) : 0u, soZy = p.zy ? static_cast<unsigned>((p.zy - p.rJ) * 8) : 0u,
                   soFc = p.fcor ? static_cast<unsigned>((p.fcor - p.rJ) * 8) : 0u, soCd = p.cd ? static_cast<unsigned>((p.cd - p.rJ) * 8) : 0u;
    const double g = p.g;
    const unsigned ntiles = (static_cast<unsigned>(p.K) + 15u) / 16u;
    unsigned tile, tileEnd;
    curved_wave_tiles(ntiles, tile, tileEnd);
    for (; tile < tileEnd; ++tile) {
        const unsigned kTrue = tile * 16u + j, kLast = static_cast<unsigned>(p.K) - 1u;
        const bool live = kTrue <= kLast;
        const unsigned k = live ? kTrue : kLast; // padding lanes recompute the last element, store nothing
        const unsigned k8 = k * 8u, v8 = (q * static_cast<unsigned>(ld) + k) * 8u, v4 = v8 >> 1;
        // vector offset of node row 4 t + q (out of range on the padding rows of the last k-step)
        auto nodeOff = [&](int t) -> unsigned {
            if constexpr (Np % 4 != 0) {
                if (t == KV - 1) return (4 * (KV - 1) + static_cast<int>(q) < Np) ? v8 : 0xfffffff8u;
            }
            return v8;
        };

        // ---- own nodal state in operand layout: node m = 4 t + q (0 on padding rows)
        double qB[4][KV];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < KV; ++t) qB[c][t] = cbld_f64(rq, nodeOff(t), static_cast<unsigned>(c) * planeB + static_cast<unsigned>(4 * t) * ld8);
        // straight-sided tile? (one ballot; padding lanes repeat the last element)
        bool affTile = false;
        double ca[4] = {0.0, 0.0, 0.0, 0.0};
        if (p.cubAffine) {
            affTile = __all(p.affineEl[k] != 0);
            if (affTile) {
#pragma unroll
                for (int i = 0; i < 4; ++i) ca[i] = cbld_f64(raff, k8, static_cast<unsigned>(i) * ld8);
            }
        }

        // exterior-trace indices of all three faces, requested before the volume term: each face then has one dependent
        // memory round trip (the gathers), not two
        int idxP[3][FB][4];
#pragma unroll
        for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int b = 0; b < FB; ++b)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    idxP[f][b][reg] = cbld_i32(rmapP, v4, static_cast<unsigned>(16 * (f * FB + b) + 4 * reg) * ld4);

        cmfma_t acc[4][MT];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < MT; ++r) acc[c][r] = cmfma_zero();

        // ---- volume term, 16 cubature points at a time
        for (int rb = 0; rb < ncb; ++rb) {
            cmfma_t cv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) cv[c] = cmfma_zero();
#pragma unroll
            for (int t = 0; t < KV; ++t) {
                const double a = A(rb * KV + t);
#pragma unroll
                for (int c = 0; c < 4; ++c) cv[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, qB[c][t], cv[c], 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) { // 4 cubature points of this lane = contraction step reg of DrT / DsT
                const int row = 16 * rb + static_cast<int>(q) + 4 * reg;
                const unsigned so = static_cast<unsigned>(16 * rb + 4 * reg) * ld8;
                double wrx, wry, wsx, wsy;
                if (affTile) { // rule weight (times a reference Jacobian) of this point, the element's four numbers
                    const double w = sOps[wrefAt + row]; // zero on padding rows
                    wrx = w * ca[0]; wry = w * ca[1]; wsx = w * ca[2]; wsy = w * ca[3];
                } else {
                    wrx = cbld_f64(rcub[0], v8, so); wry = cbld_f64(rcub[1], v8, so);
                    wsx = cbld_f64(rcub[2], v8, so); wsy = cbld_f64(rcub[3], v8, so);
                }
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
                    const double aDr = A(offDrT + (r * ncb + rb) * 4 + reg), aDs = A(offDsT + (r * ncb + rb) * 4 + reg);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc[c][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(aDr, tr[c], acc[c][r], 0, 0, 0);
                        acc[c][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(aDs, ts[c], acc[c][r], 0, 0, 0);
                    }
                }
            }
        }

        // ---- surface term, face by face; Gauss row of this lane: 16 b + q + 4 reg of the face. One pass: the
        //      weighted central flux and jump of every point stay in registers until the face's speed is known.
        //      (Requesting face f + 1's traces and geometry while face f is worked on was tried: +66 spilled VGPRs at two
        //      waves per SIMD, 0.61 -> 0.67 ms per evaluation; not kept.)
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            double lam = 0.0;
            double ef[FB][4][4], dj[FB][4][4];
#pragma unroll
            for (int b = 0; b < FB; ++b) {
                const int gb = f * FB + b;
                // the element's own traces at this block's 16 Gauss rows: the Gauss kernel's products (identity gmapM)
                cmfma_t gM[4];
                if constexpr (!MAPM) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) gM[c] = cmfma_zero();
#pragma unroll
                    for (int t = 0; t < KV; ++t) {
                        const double a = A(offGI + gb * KV + t);
#pragma unroll
                        for (int c = 0; c < 4; ++c) gM[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, qB[c][t], gM[c], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int local = 16 * b + static_cast<int>(q) + 4 * reg;
                    const bool valid = local < p.ng;
                    const unsigned so8 = static_cast<unsigned>(16 * gb + 4 * reg) * ld8, so4 = static_cast<unsigned>(16 * gb + 4 * reg) * ld4;
                    const int idP = idxP[f][b][reg];
                    const unsigned oP = static_cast<unsigned>(idP < 0 ? -(idP + 1) : idP) * 8u;
                    double hM, huM, hvM, hNM;
                    if constexpr (MAPM) {
                        const unsigned oM = static_cast<unsigned>(cbld_i32(rmapM, v4, so4)) * 8u;
                        hM = cbld_f64(rgq, oM, 0u); huM = cbld_f64(rgq, oM, gplaneB);
                        hvM = cbld_f64(rgq, oM, 2u * gplaneB); hNM = cbld_f64(rgq, oM, 3u * gplaneB);
                    } else {
                        hM = gM[0][reg]; huM = gM[1][reg]; hvM = gM[2][reg]; hNM = gM[3][reg];
                    }
                    double hP = cbld_f64(rgq, oP, 0u), huP = cbld_f64(rgq, oP, gplaneB), hvP = cbld_f64(rgq, oP, 2u * gplaneB),
                           hNP = cbld_f64(rgq, oP, 3u * gplaneB);
                    const double nx = cbld_f64(rgg, v8, so8), ny = cbld_f64(rgg, v8, gplaneB + so8),
                                 hW = 0.5 * cbld_f64(rgg, v8, 2u * gplaneB + so8); // zero on padding rows
                    if (!valid) { hM = 1.0; hP = 1.0; huM = hvM = hNM = huP = hvP = hNP = 0.0; }
                    const double rM = crcp(hM), rP = crcp(hP);
                    // the wave speeds use the exterior velocity BEFORE the wall condition (rhs.py:81-85, :93-94)
                    const double uM = huM * rM, vM = hvM * rM, uP0 = huP * rP, vP0 = hvP * rP;
                    const double spdM = csqrt(uM * uM + vM * vM) + csqrt(g * hM);
                    const double spdP = csqrt(uP0 * uP0 + vP0 * vP0) + csqrt(g * hP);
                    lam = valid ? fmax(lam, fmax(spdM, spdP)) : lam;
                    if (idP < 0) { // reflective wall (rhs.py:87-88)
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
                }
            }
            lam = fmax(lam, __shfl_xor(lam, 16)); // the face's Gauss points sit in the 4 lanes q of this element
            lam = fmax(lam, __shfl_xor(lam, 32));
#pragma unroll
            for (int b = 0; b < FB; ++b)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    double sf[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) sf[c] = fma(lam, dj[b][reg][c], ef[b][reg][c]);
#pragma unroll
                    for (int r = 0; r < MT; ++r) {
                        const double a = A(offIT + (r * 3 * FB + f * FB + b) * 4 + reg);
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[c][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sf[c], acc[c][r], 0, 0, 0);
                    }
                }
        }

        // ---- mass inverse, sources, update. acc[c][t >> 2][t & 3] is node m = 4 t + q: the operand layout again.
        const int slot = p.curvedSlot ? p.curvedSlot[k] : -1;
        cmfma_t out[4][MT];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < MT; ++r) out[c][r] = cmfma_zero();
        double S2[KV], S3[KV];
#pragma unroll
        for (int t = 0; t < KV; ++t) {
            const int m = 4 * t + static_cast<int>(q);
            const unsigned so = static_cast<unsigned>(4 * t) * ld8, vo = nodeOff(t);
            const double rj = cbld_f64(rcoef, vo, so); // 0 on padding rows
            {   // momentum sources at the node (rhs.py:165-174): S2 = f hv - CD |u| u - g h zx, S3 = -(f hu - CD |u| v) - g h zy
                const double h = m < Np ? qB[0][t] : 1.0, hu = qB[1][t], hv = qB[2][t];
                const double rh = crcp(h);
                const double u = hu * rh, v = hv * rh;
                const double fco = p.fcor ? cbld_f64(rcoef, vo, soFc + so) : p.fconst, cdv = p.cd ? cbld_f64(rcoef, vo, soCd + so) : p.cdconst;
                const double cdn = cdv * csqrt(u * u + v * v);
                const double zx = p.zx ? cbld_f64(rcoef, vo, soZx + so) : 0.0, zy = p.zy ? cbld_f64(rcoef, vo, soZy + so) : 0.0;
                S2[t] = m < Np ? (fco * hv - cdn * u) - g * h * zx : 0.0;
                S3[t] = m < Np ? -(fco * hu - cdn * v) - g * h * zy : 0.0;
            }
#pragma unroll
            for (int r = 0; r < MT; ++r) {
                const double a = A(offMass + r * KV + t);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    out[c][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[c][t >> 2][t & 3] * rj, out[c][r], 0, 0, 0);
                if constexpr (FILTER) {
                    const double af = A(offF + r * KV + t);
                    out[1][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, S2[t], out[1][r], 0, 0, 0);
                    out[2][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, S3[t], out[2][r], 0, 0, 0);
                }
            }
        }
        if (live) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int t = 0; t < KV; ++t) {
                    const int m = 4 * t + static_cast<int>(q);
                    if (m >= Np) continue;
                    if (slot >= 0) { // element of curvedEls: its own mass matrix is applied by the fix-up kernel
                        p.mmSide[(static_cast<size_t>(slot) * 4 + c) * Np + m] = acc[c][t >> 2][t & 3];
                        continue;
                    }
                    double R = out[c][t >> 2][t & 3];
                    if constexpr (!FILTER) R += c == 1 ? S2[t] : (c == 2 ? S3[t] : 0.0);
                    const unsigned so = static_cast<unsigned>(c) * planeB + static_cast<unsigned>(4 * t) * ld8;
                    if constexpr (MODE == CMODE_RHS) {
                        cbst_f64(rout, v8, so, R);
                    } else if constexpr (MODE == CMODE_LSERK) {
                        const double n1 = p.ca * cbld_f64(rold, v8, so) + p.cc * R;
                        cbst_f64(rold, v8, so, n1); // the residual, in place
                        cbst_f64(rout, v8, so, qB[c][t] + p.cb * n1);
                    } else {
                        cbst_f64(rout, v8, so, p.ca * cbld_f64(rold, v8, so) + p.cb * qB[c][t] + p.cc * R);
                    }
                }
        }
    }
}

// ---- elements of curvedEls: RHS_c = U^-1 U^-T MM_c with the element's own Cholesky factor (rhs.py:157-162), then
// sources, filter and update as above. The inverse mass matrix Minv = U^-1 U^-T of every listed element is formed once at
// creation (minvSide, (Np, Np) per element, symmetric), so the two triangular solves -- 2 Np dependent steps per lane,
// 0.22 ms for 3000 elements at N = 8 in the first version -- are one dense product: lane i of an element's group of P lanes
// owns node i, reads column i of Minv (coalesced: row m of a symmetric matrix) and takes MM_c(m) from lane m by a
// shuffle; the same for Filter. 64 / P elements per wave (P = 16, 32 or 64 lanes >= Np).
template <int N, int MODE, bool FILTER>
__global__ __launch_bounds__(64) void sw2d_curved_fixup_kernel(const CurvedParams p) {
    constexpr int Np = (N + 1) * (N + 2) / 2, P = Np <= 16 ? 16 : (Np <= 32 ? 32 : 64), EPW = 64 / P;
    const int lane = static_cast<int>(threadIdx.x), e = lane / P, i = lane % P, first = e * P;
    const int slotTrue = static_cast<int>(blockIdx.x) * EPW + e;
    const bool has = slotTrue < p.numCurved, mine = has && i < Np;
    const int col = has ? slotTrue : p.numCurved - 1, ic = i < Np ? i : Np - 1; // (clamped: every lane reads valid memory)
    const int slot = p.slotList ? p.slotList[col] : col;
    const unsigned k = static_cast<unsigned>(p.curvedEls[slot]), k8 = k * 8u;
    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld;
    const double* __restrict__ Mi = p.minvSide + static_cast<size_t>(slot) * Np * Np;
    double x[4], y[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < 4; ++c) x[c] = p.mmSide[(static_cast<size_t>(slot) * 4 + c) * Np + ic];
    // this node's state and source tables (requested with the matrix rows)
    const long long rowOff = static_cast<long long>(ic) * ld;
    const double h = cld_row(p.qin + rowOff, k8), hu = cld_row(p.qin + plane + rowOff, k8), hv = cld_row(p.qin + 2 * plane + rowOff, k8);
    // the rows the update reads, requested with everything else (this kernel is a few waves per SIMD of pure latency)
    double own[4] = {h, hu, hv, 0.0}, old[4] = {0.0, 0.0, 0.0, 0.0};
    if constexpr (MODE != CMODE_RHS) {
        own[3] = cld_row(p.qin + 3 * plane + rowOff, k8);
        const double* __restrict__ ob = MODE == CMODE_LSERK ? p.res : p.qbase;
#pragma unroll
        for (int c = 0; c < 4; ++c) old[c] = cld_row(ob + c * plane + rowOff, k8);
    }
#pragma unroll 15
    for (int m = 0; m < Np; ++m) {
        const double a = Mi[m * Np + ic];
#pragma unroll
        for (int c = 0; c < 4; ++c) y[c] = fma(a, __shfl(x[c], first + m), y[c]);
    }
    {
        double S2, S3;
        curved_sources(p, h, hu, hv, rowOff, k8, S2, S3);
        y[1] += S2;
        y[2] += S3;
    }
    double r[4];
    if constexpr (FILTER) {
#pragma unroll
        for (int c = 0; c < 4; ++c) r[c] = 0.0;
#pragma unroll 5
        for (int m = 0; m < Np; ++m) {
            const double a = p.filt[ic * Np + m];
#pragma unroll
            for (int c = 0; c < 4; ++c) r[c] = fma(a, __shfl(y[c], first + m), r[c]);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) r[c] = y[c];
    }
    if (!mine) return;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const long long off = c * plane + rowOff;
        if constexpr (MODE == CMODE_RHS) {
            cst_row(p.rhs + off, k8, r[c]);
        } else if constexpr (MODE == CMODE_LSERK) {
            const double n1 = p.ca * old[c] + p.cc * r[c];
            cst_row(p.res + off, k8, n1);
            cst_row(p.qout + off, k8, own[c] + p.cb * n1);
        } else {
            cst_row(p.qout + off, k8, p.ca * old[c] + p.cb * own[c] + p.cc * r[c]);
        }
    }
}

// Launch table of one polynomial order (compiled per order in sw2d_curved_order.hip).
struct CurvedKernelTable {
    int order, Np, KV, MT;
    int (*opsTiles)(int ncb, int fb);                 // tiles of the whole image
    int (*stageTiles)(int ncb, int fb);               // tiles the stage kernel stages in LDS
    void (*opsOffsets)(int ncb, int fb, int* off);    // offVc, offDrT, offDsT, offIT, offM, offMF, offF, offGI
    hipError_t (*gauss)(const CurvedParams& p, hipStream_t stream);
    hipError_t (*stage)(int mode, bool filter, const CurvedParams& p, hipStream_t stream);
    hipError_t (*fixup)(int mode, bool filter, const CurvedParams& p, hipStream_t stream);
    // nodal-trace form
    int (*ntTiles)(int ncb, int fb);                  // tiles of its image
    void (*ntOffsets)(int ncb, int fb, int* off);     // VCH, SCH, KE, offSurf(0), offMass
    bool (*ntFits)(int ncb, int fb, bool filter);     // its LDS plan fits a workgroup
    hipError_t (*stageNT)(int mode, bool filter, const CurvedParams& p, hipStream_t stream);
};
const CurvedKernelTable* curved_kernel_table(int order); // nullptr if the order is not compiled in

} // namespace bdg_dev
