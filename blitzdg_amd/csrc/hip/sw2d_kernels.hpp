// sw2d_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the
// 2-D shallow-water nodal DG right-hand side fused with the Runge-Kutta stage
// update. Device-side restatement of the reference's
//   blitzdg::sw2d::computeRHS         src/sw2d-simple/main.cpp:181-356
//   LSERK4 stage update               src/advec1d/main.cpp:92-102 (include/LSERK4.hpp)
//   midpoint-RK2 + filter loop body   src/sw2d-simple/main.cpp:132-151
//   time-step / blow-up reductions    src/sw2d-simple/main.cpp:98-109,153-167
//
// Mapping to the hardware
//   * one wavefront lane owns one element: every (rows, K) table keeps the
//     reference's K-contiguous layout, so a wave's access to one nodal row is a
//     single contiguous 512-byte segment (rows are padded to a multiple of 64
//     elements, `ld`);
//   * all Np nodal values of the 3 fields and the flux columns live in registers;
//     Dr/Ds (interleaved), Lift and Filter are staged once per workgroup in LDS
//     and read with wave-uniform (broadcast) addresses, each entry feeding 5
//     (Dr/Ds) or 3 (Lift/Filter) FMAs;
//   * the neighbour trace q[vmapP] is a per-lane gather from the same planes the
//     neighbouring lanes/waves stream, served by L1/L2; the wall flag rides in
//     the sign bit of the gather index (no extra table);
//   * the stage update is fused into the same pass: q is double-buffered because
//     neighbours still need the old traces, the LSERK residual is updated in place.
// The kernel is HBM-bound at N <= 5 (about 3 flop/byte at N=4); there is no
// dense-GEMM reshaping and no MFMA here on purpose.
#pragma once
#include <hip/hip_runtime.h>

namespace bdg_dev {

template <int N>
struct Elem {
    static constexpr int Np = (N + 1) * (N + 2) / 2;
    static constexpr int Nfp = N + 1;
    static constexpr int NFN = 3 * Nfp;
    // Node numbering of the warp & blend lattice: rows of constant s, r ascending;
    // row j starts at j*(N+1) - j*(j-1)/2 and has N+1-j nodes.
    __host__ __device__ static constexpr int rowStart(int j) { return j * (N + 1) - j * (j - 1) / 2; }
    // Fmask(n, f): face 0 is s=-1 (first row), face 1 is r+s=0 (row ends), face 2
    // is r=-1 (row starts); ascending node index (reference buildNodes :692-727).
    __host__ __device__ static constexpr int fmask(int f, int n) {
        return f == 0 ? n : (f == 1 ? rowStart(n) + (N - n) : rowStart(n));
    }
    // LDS image, in doubles: [Dr,Ds interleaved | Lift | Filter]
    static constexpr int OFF_D = 0;
    static constexpr int OFF_LIFT = 2 * Np * Np;
    static constexpr int OFF_FILT = OFF_LIFT + Np * NFN;
    static constexpr int LDS_DOUBLES = OFF_FILT + Np * Np;
};

enum StageMode {
    MODE_RHS = 0,    // rhs = R(qin)
    MODE_LSERK = 1,  // res = ca*res + cc*R(qin); qout = qin + cb*res
    MODE_COMBINE = 2 // qout = ca*qbase + cb*qin + cc*R(qin)   (RK2 midpoint / Heun stages)
};

struct StageParams {
    const double* qin;   // 3 planes of Np*ld: h, hu, hv -- own values and neighbour traces
    const double* qbase; // MODE_COMBINE only
    double* qout;        // MODE_LSERK / MODE_COMBINE
    double* res;         // MODE_LSERK: 3 planes, updated in place
    double* rhs;         // MODE_RHS: 3 planes
    const double* geo;   // rx, sx, ry, sy: 4 planes of Np*ld
    const double* fgeo;  // nx, ny, Fscale: 3 planes of NFN*ld
    const int* vmapP;    // NFN*ld gather offsets n'*ld + k'; wall nodes stored as -(offset+1)
    const double* ops;   // global image of the LDS block (Elem<N>::LDS_DOUBLES doubles)
    const double* ageo;  // affine path: 13 planes of ld: rx, sx, ry, sy, nx[3], ny[3], Fscale[3]
    const double* opsAffine; // affine path: AffineOps<N> image (plain or pre-filtered operators)
    long long ld;        // padded element count (multiple of 64)
    int kbegin, kend;    // element slots [kbegin, kend) this launch updates
    double g;
    double ca, cb, cc;
    double one;          // 1.0 (run-time constant used to stop value reuse across phases)
    double sponge;       // MODE_COMBINE: momentum relaxation x /= (1 + sponge x^2) after the update (0: off)
    // Partition-boundary launches with the halo staging folded in (matrix-core kernel, HALO = true): neighbour
    // traces that live in ghost slots (>= haloOwned) are read from the received element-major records, and the
    // new state of each element is also written to its (up to three) records of the send buffer.
    const double* haloRecv; // (ghosts, haloRows) records as the neighbours packed them
    double* haloSend;       // (numSend, haloRows)
    const int* haloSendOf;  // 3 send-record indices per element of [kbegin, kend), -1 = none
    int haloOwned;          // first ghost slot
    int haloRows;           // doubles per record = fields * Np
    // Launches with one resident workgroup per CU for their whole duration (sw2d_stage_mfma3_kernel): at most this many
    // workgroups (0: one per CU). Interior launches of a partitioned run leave a few CUs to the partition-boundary
    // kernel on the exchange stream, which could not start beside 256 LDS-filling workgroups otherwise.
    int gridCap;
    // sw2d_stage_mfma3_kernel: 1 = the waves of an XCD take tiles side by side (stride = waves per XCD) instead of one
    // contiguous chunk of tiles per wave
    int tileInterleave;
    // sw2d_stage_mfma3_kernel: workgroups start (blk mod 8) * stagger kilocycles apart (0: together)
    int stagger;
    // profiling builds (-DBDG_PHASE_CLOCK): 16 cycle counts per wave (LDS copy, k-steps, surface, update), else unused
    unsigned long long* phaseClock;
    // In-kernel dependencies between the two chains of a partitioned stage (round 4; SYNC instances of the matrix-core kernels,
    // DESIGN.md section 4): the tiles from syncFirstTile on (interior launches: the ring of elements next to the partition
    // boundary, ordered last; strip launches: every tile, syncFirstTile = 0) wait until *syncWait >= syncWaitValue before
    // anything of theirs is read or written, and add one to *syncSignal when their stores are visible device-wide
    // (interior: one per ring tile, strip: one per workgroup). A wait that does not end within the spin bound sets
    // *syncError and goes on (the host reports it): no wave can hang the device.
    const unsigned long long* syncWait;
    unsigned long long syncWaitValue;
    unsigned long long* syncSignal;
    unsigned int* syncError;
    int syncFirstTile;
    unsigned int* syncSignalsOut; // host side only: the launch helper stores the number of signals the launch will add
    // host side only (the launch helpers of sw2d_order.hip): when set, the launch records this event through its own completion
    // signal (hipExtLaunchKernelGGL) instead of a separate record packet behind it, and sets *stopEventUsed; a launch path that
    // does not look at it leaves the flag alone and the caller records the event itself
    hipEvent_t stopEvent;
    bool* stopEventUsed;
};

// ---- hand-off between concurrently running kernels of one device (the guide's valid form, MI355X_MICROARCH.md, inter-workgroup
// visibility: per-XCD L2s are not coherent with each other and a CU's L1 is never refreshed by another CU's stores).
// Consumer: relaxed agent-scope polls of the counter, then ONE agent-scope acquire (invalidates this CU's L1; the waiting wave's
// own later loads are ordered behind it). Bounded: after kSyncSpinLimit polls the wave sets *err and goes on.
constexpr unsigned kSyncSpinLimit = 1u << 21;
__device__ __forceinline__ void sync_wait(const unsigned long long* ctr, unsigned long long want, unsigned int* err) {
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > kSyncSpinLimit) {
            if (err) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// Producer: every byte that is handed off is stored WRITE-THROUGH (sc1: st_row_wt / bst_f64_wt below, the guide's R1) -- such a
// store leaves the XCD's L2 for memory by itself, so no write-back of the whole L2 is needed before the signal. (First form of
// this round: plain stores + an agent-scope release fence per signal, i.e. buffer_wbl2 of an L2 that the interior launch keeps
// filling with dirty lines. 8-way rehearsal, per stage: N=4 0.0589 ms against 0.0567 with events, N=6 0.0552 against 0.0515 --
// the write-backs cost more than the queue waits they replaced. profiles/r04_rehearsal_experiments.txt.)
// One wave for its own stores: drain them (a store counts until it is acknowledged), then one lane adds to the counter.
__device__ __forceinline__ void sync_signal_wave(unsigned long long* ctr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63u) == 0u) __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// One workgroup for all its waves' stores (every wave calls this at the end of the kernel).
__device__ __forceinline__ void sync_signal_workgroup(unsigned long long* ctr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0u) __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// write-through store at a wave-uniform row pointer plus a per-lane byte offset (global_store_dwordx2 ... sc1)
__device__ __forceinline__ void st_row_wt(double* row, unsigned byteOff, double v) {
    __hip_atomic_store(reinterpret_cast<double*>(reinterpret_cast<char*>(row) + byteOff), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sponge-layer relaxation of the reference's variant-B driver (src/sw2d/main.cpp:223-224,234-235):
// hu /= (1 + sigma hu^2), applied to the momentum components after a stage update.
__device__ __forceinline__ double sponge_relax(double x, double sigma) {
    return sigma != 0.0 ? x / (1.0 + sigma * x * x) : x;
}
// The same without the test (x / (1 + 0 x^2) is x): for fully unrolled one-lane-per-element bodies, where a run-time test
// per value splits the kernel's one scheduling region and everything held in scalar registers is hoisted and spilled
// (DESIGN.md 3.6). Costs a division per value also when sigma is 0: used by the cross-check kernels only.
__device__ __forceinline__ double sponge_relax_no_test(double x, double sigma) { return x / (1.0 + sigma * x * x); }

template <int N, int MODE, bool FILTER>
__global__ __launch_bounds__(256) void sw2d_stage_kernel(const StageParams p) {
    using E = Elem<N>;
    constexpr int Np = E::Np, Nfp = E::Nfp, NFN = E::NFN;
    __shared__ double sOps[E::LDS_DOUBLES];
    for (int t = threadIdx.x; t < E::LDS_DOUBLES; t += blockDim.x) sOps[t] = p.ops[t];
    __syncthreads();

    // XCD-aware block remap: consecutive tiles of elements (which share faces,
    // hence gather lines) go to the same XCD's L2. Bijective for any grid size.
    const unsigned nwg = gridDim.x, xcd = blockIdx.x % 8u, q8 = nwg / 8u, r8 = nwg % 8u;
    const unsigned tile = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + blockIdx.x / 8u;
    const long long k = p.kbegin + static_cast<long long>(tile) * blockDim.x + threadIdx.x;
    if (k >= p.kend) return;

    const long long ld = p.ld, plane = static_cast<long long>(Np) * ld;
    const double* __restrict__ qh = p.qin + k;
    const double* __restrict__ qhu = qh + plane;
    const double* __restrict__ qhv = qhu + plane;

    // ---- own nodal values and neighbour gather indices
    double h[Np], hu[Np], hv[Np];
    int idx[NFN];
#pragma unroll
    for (int j = 0; j < NFN; ++j) idx[j] = p.vmapP[j * ld + k];
#pragma unroll
    for (int n = 0; n < Np; ++n) {
        h[n] = qh[n * ld];
        hu[n] = qhu[n * ld];
        hv[n] = qhv[n * ld];
    }
    // ---- neighbour ('+') traces
    double hP[NFN], huP[NFN], hvP[NFN];
#pragma unroll
    for (int j = 0; j < NFN; ++j) {
        const int o = idx[j] < 0 ? -(idx[j] + 1) : idx[j];
        hP[j] = p.qin[o];
        huP[j] = p.qin[plane + o];
        hvP[j] = p.qin[2 * plane + o];
    }

    // ---- volume fluxes: F1 = hu, G1 = hv, F2, G2 (= F3), G3
    const double g = p.g, halfg = 0.5 * p.g;
    double u[Np], v[Np], F2[Np], G2[Np], G3[Np];
#pragma unroll
    for (int n = 0; n < Np; ++n) {
        u[n] = hu[n] / h[n];
        v[n] = hv[n] / h[n];
        const double pr = halfg * h[n] * h[n];
        F2[n] = hu[n] * u[n] + pr;
        G2[n] = hu[n] * v[n];
        G3[n] = hv[n] * v[n] + pr;
    }

    // ---- surface terms: s_c[j] = Fscale * 0.5 * ((F_c^- - F_c^+) nx + (G_c^- - G_c^+) ny - lambda (q_c^- - q_c^+))
    double s1[NFN], s2[NFN], s3[NFN];
    const double* __restrict__ fg = p.fgeo + k;
    const long long fplane = static_cast<long long>(NFN) * ld;
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        double lam = 0.0;
        double dh[Nfp], dhu[Nfp], dhv[Nfp], e1[Nfp], e2[Nfp], e3[Nfp];
#pragma unroll
        for (int n = 0; n < Nfp; ++n) {
            const int j = f * Nfp + n, m = E::fmask(f, n);
            const double nxj = fg[j * ld], nyj = fg[fplane + j * ld];
            const double hM = h[m], huM = hu[m], hvM = hv[m];
            double hPj = hP[j], huPj = huP[j], hvPj = hvP[j];
            if (idx[j] < 0) { // reflective wall: no normal flow
                const double un = huM * nxj + hvM * nyj;
                huPj = huM - 2 * nxj * un;
                hvPj = hvM - 2 * nyj * un;
            }
            const double uP = huPj / hPj, vP = hvPj / hPj;
            const double prP = halfg * hPj * hPj;
            const double F2P = huPj * uP + prP, G2P = huPj * vP, G3P = hvPj * vP + prP;
            const double spdM = sqrt(u[m] * u[m] + v[m] * v[m]) + sqrt(g * hM);
            const double spdP = sqrt(uP * uP + vP * vP) + sqrt(g * hPj);
            lam = fmax(lam, fmax(spdM, spdP));
            dh[n] = hM - hPj;
            dhu[n] = huM - huPj;
            dhv[n] = hvM - hvPj;
            e1[n] = dhu[n] * nxj + dhv[n] * nyj;
            e2[n] = (F2[m] - F2P) * nxj + (G2[m] - G2P) * nyj;
            e3[n] = (G2[m] - G2P) * nxj + (G3[m] - G3P) * nyj;
        }
#pragma unroll
        for (int n = 0; n < Nfp; ++n) {
            const int j = f * Nfp + n;
            const double fs = fg[2 * fplane + j * ld];
            s1[j] = fs * (0.5 * (e1[n] - lam * dh[n]));
            s2[j] = fs * (0.5 * (e2[n] - lam * dhu[n]));
            s3[j] = fs * (0.5 * (e3[n] - lam * dhv[n]));
        }
    }

    // ---- per output node: flux divergence + lifted surface term
    const double* __restrict__ sD = sOps + E::OFF_D;
    const double* __restrict__ sL = sOps + E::OFF_LIFT;
    const double* __restrict__ sF = sOps + E::OFF_FILT;
    const double* __restrict__ gp = p.geo + k;
    double R1[Np], R2[Np], R3[Np];
#pragma unroll
    for (int i = 0; i < Np; ++i) {
        double rF1 = 0, sF1 = 0, rG1 = 0, sG1 = 0, rF2 = 0, sF2 = 0, rG2 = 0, sG2 = 0, rG3 = 0, sG3 = 0;
#pragma unroll
        for (int m = 0; m < Np; ++m) {
            const double dr = sD[2 * (i * Np + m)], ds = sD[2 * (i * Np + m) + 1];
            rF1 += dr * hu[m]; sF1 += ds * hu[m];
            rG1 += dr * hv[m]; sG1 += ds * hv[m];
            rF2 += dr * F2[m]; sF2 += ds * F2[m];
            rG2 += dr * G2[m]; sG2 += ds * G2[m];
            rG3 += dr * G3[m]; sG3 += ds * G3[m];
        }
        double l1 = 0, l2 = 0, l3 = 0;
#pragma unroll
        for (int j = 0; j < NFN; ++j) {
            const double lj = sL[i * NFN + j];
            l1 += lj * s1[j];
            l2 += lj * s2[j];
            l3 += lj * s3[j];
        }
        const double rx = gp[i * ld], sx = gp[plane + i * ld], ry = gp[2 * plane + i * ld],
                     sy = gp[3 * plane + i * ld];
        double r1 = -(rx * rF1 + sx * sF1);
        r1 += -(ry * rG1 + sy * sG1);
        double r2 = -(rx * rF2 + sx * sF2);
        r2 += -(ry * rG2 + sy * sG2);
        double r3 = -(rx * rG2 + sx * sG2);
        r3 += -(ry * rG3 + sy * sG3);
        R1[i] = r1 + l1;
        R2[i] = r2 + l2;
        R3[i] = r3 + l3;
    }

    // ---- modal filter (RK2 driver of the reference applies Filt to every RHS)
    if constexpr (FILTER) {
        double T1[Np], T2[Np], T3[Np];
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            double a = 0, b = 0, c = 0;
#pragma unroll
            for (int m = 0; m < Np; ++m) {
                const double fm = sF[i * Np + m];
                a += fm * R1[m];
                b += fm * R2[m];
                c += fm * R3[m];
            }
            T1[i] = a; T2[i] = b; T3[i] = c;
        }
#pragma unroll
        for (int i = 0; i < Np; ++i) { R1[i] = T1[i]; R2[i] = T2[i]; R3[i] = T3[i]; }
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
            const double n1 = a * rs[i * ld] + dt * R1[i];
            const double n2 = a * rs[plane + i * ld] + dt * R2[i];
            const double n3 = a * rs[2 * plane + i * ld] + dt * R3[i];
            rs[i * ld] = n1;
            rs[plane + i * ld] = n2;
            rs[2 * plane + i * ld] = n3;
            o[i * ld] = h[i] + b * n1;
            o[plane + i * ld] = hu[i] + b * n2;
            o[2 * plane + i * ld] = hv[i] + b * n3;
        }
    } else {
        const double* __restrict__ qb = p.qbase + k;
        double* __restrict__ o = p.qout + k;
        const double a = p.ca, b = p.cb, c = p.cc;
#pragma unroll
        for (int i = 0; i < Np; ++i) {
            o[i * ld] = a * qb[i * ld] + b * h[i] + c * R1[i];
            o[plane + i * ld] = sponge_relax_no_test(a * qb[plane + i * ld] + b * hu[i] + c * R2[i], p.sponge);
            o[2 * plane + i * ld] = sponge_relax_no_test(a * qb[2 * plane + i * ld] + b * hv[i] + c * R3[i], p.sponge);
        }
    }
}

// ------------------------------------------------------------------ reductions

// Per-block partial maxima for the adaptive time step and the blow-up check:
//   out[2*b]   = max over face nodes of |Fscale| * (sqrt(u^2+v^2) + sqrt(g h)) at the '-' node
//   out[2*b+1] = max |h - H| (or |h|), NaN-propagating
// Contraction is off so the value is bit-identical to the host formula
// (reference src/sw2d-simple/main.cpp:159-167; max is order independent).
template <int N>
__global__ __launch_bounds__(256) void sw2d_dt_kernel(const double* __restrict__ q, const double* __restrict__ fscale,
                                                      const double* __restrict__ H, long long ld, int K, double g,
                                                      double* __restrict__ out) {
#pragma clang fp contract(off)
    using E = Elem<N>;
    constexpr int Np = E::Np, Nfp = E::Nfp;
    const long long plane = static_cast<long long>(Np) * ld;
    const long long k = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    double fmaxv = 0.0, emax = 0.0;
    bool bad = false;
    if (k < K) {
        double spd[Np];
#pragma unroll
        for (int n = 0; n < Np; ++n) {
            const double h = q[n * ld + k], hu = q[plane + n * ld + k], hv = q[2 * plane + n * ld + k];
            const double u = hu / h, v = hv / h;
            spd[n] = sqrt(u * u + v * v) + sqrt(g * h);
            const double eta = H ? h - H[n * ld + k] : h;
            const double ae = fabs(eta);
            if (ae != ae) bad = true;
            emax = fmax(emax, ae);
        }
#pragma unroll
        for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int n = 0; n < Nfp; ++n) {
                const int j = f * Nfp + n;
                const double val = fabs(fscale[j * ld + k]) * spd[E::fmask(f, n)];
                if (val != val) bad = true;
                fmaxv = fmax(fmaxv, val);
            }
    }
    __shared__ double sA[256], sB[256];
    __shared__ int sBad;
    if (threadIdx.x == 0) sBad = 0;
    __syncthreads();
    if (bad) sBad = 1;
    sA[threadIdx.x] = fmaxv;
    sB[threadIdx.x] = emax;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (static_cast<int>(threadIdx.x) < s) {
            sA[threadIdx.x] = fmax(sA[threadIdx.x], sA[threadIdx.x + s]);
            sB[threadIdx.x] = fmax(sB[threadIdx.x], sB[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double nan = __builtin_nan("");
        out[2 * blockIdx.x] = sBad ? nan : sA[0];
        out[2 * blockIdx.x + 1] = sBad ? nan : sB[0];
    }
}

// ------------------------------------------------------------------ output step
// Primitive output fields of the drivers (eta = h - H, u = hu/h, v = hv/h; reference
// src/sw2d-simple/main.cpp:123-131, src/sw2d/main.cpp:203-210), optionally interpolated to the
// equispaced lattice of each element with the (Np, Np) matrix of splitElements
// (src/TriangleNodesProvisioner.cpp:1154-1180) before they leave the device. Contraction is off and
// the sum runs over ascending m so the values equal the host's splitElements bit for bit.
template <int N>
__global__ __launch_bounds__(256) void sw2d_output_kernel(const double* __restrict__ q, const double* __restrict__ H,
                                                          const double* __restrict__ M, double* __restrict__ out,
                                                          long long ld, int K, int which) {
#pragma clang fp contract(off)
    using E = Elem<N>;
    constexpr int Np = E::Np;
    const long long plane = static_cast<long long>(Np) * ld;
    const long long k = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= K) return;
    double val[Np];
#pragma unroll
    for (int m = 0; m < Np; ++m) {
        const double h = q[m * ld + k];
        if (which == 0) val[m] = H ? h - H[m * ld + k] : h;
        else val[m] = q[which * plane + m * ld + k] / h;
    }
    if (!M) {
#pragma unroll
        for (int i = 0; i < Np; ++i) out[i * ld + k] = val[i];
        return;
    }
#pragma unroll 1
    for (int i = 0; i < Np; ++i) {
        const double* __restrict__ row = M + i * Np;
        double acc = 0.0;
#pragma unroll
        for (int m = 0; m < Np; ++m) acc = acc + row[m] * val[m];
        out[i * ld + k] = acc;
    }
}

} // namespace bdg_dev
