// sw2d_device.hip -- device-resident sw2d solver behind the C ABI
// (include/blitzdg_hip.h, group 2). Owns the HBM image of the DG tables and the
// state, launches the fused stage kernels (sw2d_kernels.hpp) on its own stream.
//
// HBM layout (fp64 planes of `ld` = K rounded up to 64 elements per nodal row):
//   qA, qB      3*Np*ld   state (h, hu, hv), double-buffered across stages
//   res         3*Np*ld   LSERK4 residual
//   aux         3*Np*ld   RHS output / RK2 intermediate stage
//   geo         4*Np*ld   rx, sx, ry, sy
//   fgeo        9*Nfp*ld  nx, ny, Fscale
//   vmapP       3*Nfp*ld  int32 gather offsets (wall flag in the sign bit)
//   ops         Dr|Ds interleaved, Lift, Filter (copied to LDS by every workgroup)
// Elements may be renumbered internally (BDG_SW2D_REORDER); all I/O is in the
// caller's numbering.
#include "../host/capi_internal.hpp"
#include "../host/parallel_for.hpp"
#include "blitzdg/LSERK4.hpp"
#include "rccl_api.hpp"
#include "sw2d_launch.hpp"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <queue>
#include <stdexcept>
#include <string>
#include <vector>

namespace bdg_dev {

const KernelTable* kernel_table_order1();
const KernelTable* kernel_table_order2();
const KernelTable* kernel_table_order3();
const KernelTable* kernel_table_order4();
const KernelTable* kernel_table_order5();
const KernelTable* kernel_table_order6();
const KernelTable* kernel_table_order7();
const KernelTable* kernel_table_order8();

const KernelTable* kernel_table(int order) {
    switch (order) {
    case 1: return kernel_table_order1();
    case 2: return kernel_table_order2();
    case 3: return kernel_table_order3();
    case 4: return kernel_table_order4();
    case 5: return kernel_table_order5();
    case 6: return kernel_table_order6();
    case 7: return kernel_table_order7();
    case 8: return kernel_table_order8();
    default: return nullptr;
    }
}

// Final reduction of the per-block partials of sw2d_dt_kernel (single block).
__global__ __launch_bounds__(256) void sw2d_reduce_kernel(const double* __restrict__ part, int nblocks,
                                                          double* __restrict__ out2) {
    __shared__ double sA[256], sB[256];
    __shared__ int sBad;
    if (threadIdx.x == 0) sBad = 0;
    __syncthreads();
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) {
        const double x = part[2 * i], y = part[2 * i + 1];
        if (x != x || y != y) sBad = 1;
        a = fmax(a, x);
        b = fmax(b, y);
    }
    sA[threadIdx.x] = a;
    sB[threadIdx.x] = b;
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
        out2[0] = sBad ? nan : sA[0];
        out2[1] = sBad ? nan : sB[0];
    }
}

// (rows, K) caller-order image <-> padded, optionally renumbered device planes.
// perm[k_caller] = device slot (NULL = identity). One thread per (row, element).
template <typename T>
__global__ void scatter_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, int rows, int K, long long ld,
                                    const int* __restrict__ perm) {
    const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= static_cast<long long>(rows) * K) return;
    const long long r = t / K, k = t % K;
    dst[r * ld + (perm ? perm[k] : k)] = src[t];
}

template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, int rows, int K, long long ld,
                                   const int* __restrict__ perm) {
    const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= static_cast<long long>(rows) * K) return;
    const long long r = t / K, k = t % K;
    dst[t] = src[r * ld + (perm ? perm[k] : k)];
}

// Halo exchange staging: element-major buffers of 3*Np doubles per element.
//   pack:   buf[i*rows + r] = q[r*ld + slots[i]]        (owned elements a neighbour rank needs)
//   unpack: q[r*ld + first + i] = buf[i*rows + r]       (ghost elements, stored after the owned ones)
__global__ void halo_pack_kernel(const double* __restrict__ q, double* __restrict__ buf, const int* __restrict__ slots,
                                 int count, int rows, long long ld) {
    const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= static_cast<long long>(count) * rows) return;
    const long long r = t / count, i = t % count; // consecutive lanes read consecutive elements of one row
    buf[i * rows + r] = q[r * ld + slots[i]];
}

__global__ void halo_unpack_kernel(double* __restrict__ q, const double* __restrict__ buf, int first, int count,
                                   int rows, long long ld) {
    const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= static_cast<long long>(count) * rows) return;
    const long long r = t / count, i = t % count;
    q[r * ld + first + i] = buf[i * rows + r];
}

// ---- bandwidth probes (measurement aids, not part of the solver)
// STREAM triad a = b + s*c on 16-byte lanes: the practical HBM roof of this device.
__global__ __launch_bounds__(256) void triad_kernel(double2* __restrict__ a, const double2* __restrict__ b,
                                                    const double2* __restrict__ c, double s, size_t n2) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n2; i += stride) {
        const double2 x = b[i], y = c[i];
        a[i] = make_double2(x.x + s * y.x, x.y + s * y.y);
    }
}

// Same rows, same 8-byte-per-lane accesses and same read/write mix as one fused LSERK stage of
// the affine kernel (reads: state 3Np rows, residual 3Np rows, 13 geometry rows, 3Nfp index rows;
// writes: residual and state, 6Np rows), with no gathers and almost no arithmetic: the time of
// this launch is the memory-system floor for the stage kernel's own access pattern.
__global__ __launch_bounds__(256) void stage_traffic_probe_kernel(const double* __restrict__ qin, double* __restrict__ qout,
                                                                 double* __restrict__ res, const double* __restrict__ ageo,
                                                                 const int* __restrict__ vmapP, int rows, int idxRows,
                                                                 long long ld, int K) {
    const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= static_cast<unsigned>(K)) return;
    double acc = 0.0;
    for (int r = 0; r < 13; ++r) acc += ageo[r * ld + k];
    for (int r = 0; r < idxRows; ++r) acc += vmapP[r * ld + k];
    for (int r = 0; r < rows; ++r) {
        const double q = qin[r * ld + k], o = res[r * ld + k];
        const double n = 0.5 * o + 1e-300 * (q + acc);
        res[r * ld + k] = n;
        qout[r * ld + k] = q + 1e-300 * n;
    }
}

} // namespace bdg_dev

using bdg_detail::arg_error;
using bdg_detail::guard;
using bdg_detail::hip_error;
using bdg_detail::unstable_error;

namespace {

void hipCheck(hipError_t e, const char* what) {
    if (e != hipSuccess) throw hip_error(std::string(what) + ": " + hipGetErrorString(e));
}

using bdg_rccl::ncclCheck;
using bdg_rccl::rccl;
using bdg_rccl::RcclApi;

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    void alloc(size_t count, size_t& total) {
        release();
        if (count == 0) return;
        hipCheck(hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)), "hipMalloc");
        n = count;
        total += count * sizeof(T);
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
};

} // namespace

namespace {
std::vector<double> matmulHost(const double* A, const double* B, int n, int c);
}

struct bdg_sw2d {
    const bdg_dev::KernelTable* kt = nullptr;
    int N = 0, Np = 0, Nfp = 0, NFN = 0, K = 0, device = 0;
    long long ld = 0;
    double g = 9.81;
    bool hasFilter = false, hasH = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    size_t bytes = 0;
    DevBuf<double> qA, qB, res, aux, geo, fgeo, ops, Hbuf, stage, partials, red2;
    DevBuf<double> ageo, opsAffine, opsAffineFiltered; // affine-geometry fast path
    DevBuf<double> opsMfma, opsMfmaFiltered;           // same operators in MFMA A-operand layout
    DevBuf<double> opsMfma2, opsMfma2Filtered;         // ... with the lift tiles padded per face
    DevBuf<double> opsMfma2Src, opsMfma2SrcFiltered;   // ... followed by the source-term tiles F' (variants C/D, N >= 6)
    bool mfmaSources = false;
    bool affine = false;
    // variant D (reference swhelpers/rhs.py:178-311): optional tracer field and source terms
    int nf = 3;
    bool variantD = false;
    bdg_dev::VdParams vd{};
    DevBuf<double> zxBuf, zyBuf, fcorBuf, opsVd, opsVdFiltered;
    DevBuf<double> opsVn, filterRows, vnRaw; // variants B / C / D on per-node geometry (sw2d_vn_kernel.hpp)
    // variant B (reference src/sw2d/main.cpp:279-484): depth + star states, open boundary, global LF speed, sources
    bool variantB = false;
    bdg_dev::VbParams vb{};
    DevBuf<double> HxBuf, HyBuf, spongeBuf, lamBuf, vbPartials;
    DevBuf<double> outM; // (Np, Np) lattice interpolation of the output step
    DevBuf<int> obcBuf;
    double tideAmp = 0.0, tidePeriod = 1.0, tideRamp = 0.0;
    double timeNow = 0.0;  // model time of the resident state (tide phase)
    std::vector<double> hostDr, hostDs, hostLift, hostFilter; // kept for operator images built after creation
    int affineVariant = 0; // 0: unrolled, register-resident state; 2/3: unrolled, streamed state at 2/3 waves
                           // per SIMD; 1: rolled, one field per wave; 4: rolled, three fields per lane; 5: matrix cores (MFMA f64),
                           // whole tile unrolled; 6: matrix cores, face-by-face / chunked schedule at 2 waves per SIMD
    DevBuf<int> vmapP, perm, istage, sendSlots, haloSendOf;
    bool haloFusable = false;  // every sent element is a partition-boundary element with at most three records
    int numInterior = 0, numOwned = 0, numSend = 0; // element partition: [interior | boundary | ghost]
    // native halo exchange (RCCL over xGMI): one send and one receive range per neighbour rank
    struct Peer { int rank, sendStart, sendCount, recvStart, recvCount; };
    std::vector<Peer> peers;
    ncclComm_t comm = nullptr;
    int commRank = 0, commWorld = 1;
    hipStream_t commStream = nullptr;
    hipEvent_t evA[2] = {nullptr, nullptr}, evB[2] = {nullptr, nullptr};
    hipEvent_t evPacked[2] = {nullptr, nullptr}, evCopied[2] = {nullptr, nullptr}; // in-process group transport
    bool localGroup = false;
    DevBuf<double> sendBuf, recvBuf, scalarBuf;
    // In-kernel dependencies between the two chains of an exchanged stage (round 4): two device counters --
    // [0] ring tiles of interior launches finished, [1] workgroups of boundary launches finished, [2] a word a bounded wait
    // sets when it gives up -- and what the host expects them to reach after the launches issued so far
    DevBuf<unsigned long long> syncBuf;
    unsigned long long expectRing = 0, expectStrip = 0;
    int ringBegin = 0;   // first interior element that has a partition-boundary neighbour (elements are ordered [deep | ring | boundary | ghost]
                         // by halo.build_plan; any other order only makes more tiles wait)
    double* fscaleNodal = nullptr; // (NFN, ld) per-node Fscale plane (time-step reduction)
    double* qcur = nullptr;  // current state
    double* qalt = nullptr;  // the other buffer
    long long stageCount = 0; // LSERK stage counter (stage index = count % 5)
    std::vector<int> permHost; // caller element -> device slot (empty = identity)

    ~bdg_sw2d() {
        if (comm) (void)rccl().CommDestroy(comm);
        for (hipEvent_t e : {evA[0], evA[1], evB[0], evB[1], evPacked[0], evPacked[1], evCopied[0], evCopied[1]})
            if (e) (void)hipEventDestroy(e);
        if (commStream) (void)hipStreamDestroy(commStream);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (stream) (void)hipStreamDestroy(stream);
    }

    void use() const { hipCheck(hipSetDevice(device), "hipSetDevice"); }
    size_t planeSize() const { return static_cast<size_t>(Np) * static_cast<size_t>(ld); }
    const int* permDev() const { return perm.p; }

    // host (rows, K) caller order -> device planes (padded / renumbered)
    void uploadRows(const double* host, double* dev, int rows) {
        const size_t n = static_cast<size_t>(rows) * K;
        hipCheck(hipMemcpyAsync(stage.p, host, n * sizeof(double), hipMemcpyHostToDevice, stream), "H2D copy");
        const unsigned grid = static_cast<unsigned>((n + 255) / 256);
        hipLaunchKernelGGL((bdg_dev::scatter_rows_kernel<double>), dim3(grid), dim3(256), 0, stream, stage.p, dev, rows,
                           K, ld, permDev());
        hipCheck(hipGetLastError(), "scatter_rows_kernel");
        hipCheck(hipStreamSynchronize(stream), "upload sync"); // staging buffer is reused
    }
    void downloadRows(const double* dev, double* host, int rows) {
        const size_t n = static_cast<size_t>(rows) * K;
        const unsigned grid = static_cast<unsigned>((n + 255) / 256);
        hipLaunchKernelGGL((bdg_dev::gather_rows_kernel<double>), dim3(grid), dim3(256), 0, stream, dev, stage.p, rows,
                           K, ld, permDev());
        hipCheck(hipGetLastError(), "gather_rows_kernel");
        hipCheck(hipMemcpyAsync(host, stage.p, n * sizeof(double), hipMemcpyDeviceToHost, stream), "D2H copy");
        hipCheck(hipStreamSynchronize(stream), "download sync");
    }

    bdg_dev::StageParams baseParams() const {
        bdg_dev::StageParams p{};
        p.geo = geo.p;
        p.fgeo = fgeo.p;
        p.vmapP = vmapP.p;
        p.ops = ops.p;
        p.ageo = ageo.p;
        p.opsAffine = opsAffine.p;
        p.ld = ld;
        p.kbegin = 0;
        p.kend = numOwned;
        p.g = g;
        p.one = 1.0;
        return p;
    }

    // One fused pass. The affine path takes the filter through pre-multiplied operators.
    // `on`: stream to launch on (default: the compute stream).
    void launchStage(int mode, bool filter, bdg_dev::StageParams& p, const char* what, hipStream_t on = nullptr) {
        if (filter && !hasFilter) throw arg_error("filter requested but the solver was created without a Filter matrix");
        hipStream_t st = on ? on : stream;
        // Small launches (a rank's share of a many-way split, its partition-boundary strip) are bound by
        // the latency of one wavefront, not by HBM: the matrix-core kernel gives each wave 16 elements
        // instead of 64 (measured at N=4: 750 elements 7 us vs 19 us; 125 k elements 48 us vs 56 us;
        // 250 k elements 136 us vs 109 us -- DESIGN.md section 4).
        int variant = affineVariant;
        // BDG_SW2D_SMALL_LAUNCH=n pins the crossover (A/B runs; the partition-boundary strip keeps the table's value: halosFold)
        static const int smallPinned = [] { const char* e = std::getenv("BDG_SW2D_SMALL_LAUNCH"); return e ? std::atoi(e) : -1; }();
        const int smallLaunch = (smallPinned >= 0 && p.kbegin == 0) ? smallPinned : kSmallLaunch[N];
        if (!variantForced && affine && N <= 5 && p.kend - p.kbegin < smallLaunch) variant = 5;
        if (p.syncSignal && !(affine && !variantB && !variantD && (variant == 5 || variant == 7)))
            throw std::logic_error("in-kernel stage dependencies were requested for a launch whose kernel has no SYNC instance");
        if (!affine && (variantB || variantD)) {
            // per-node geometry tables: the general (rolled) form of variants B / C / D
            buildNodalVariantOps();
            const double* filt = filter ? filterRows.p : nullptr;
            if (variantB) {
                vb.tide = tideAt(timeNow);
                vb.lam = lamBuf.p;
                vb.lamNext = nullptr;
                lamStateFor = nullptr;
                hipCheck(kt->stageVn(mode, 2, p, vd, vb, opsVn.p, filt, vnRaw.p, vbPartials.p, lamBuf.p, !lamExternal, st), what);
            } else {
                bdg_dev::VdParams v = vd;
                v.cbase = 0;
                hipCheck(kt->stageVn(mode, 1, p, v, vb, opsVn.p, filt, vnRaw.p, nullptr, nullptr, false, st), what);
            }
        } else if (variantB) {
            vb.tide = tideAt(timeNow);
            if (lamExternal) {
                // partitioned run: the global speed of this state was reduced over all ranks into lamBuf beforehand
                // (globalSpeedOf); every kernel family runs without its own speed pass
                vb.lam = lamBuf.p;
                vb.lamNext = nullptr;
                lamStateFor = nullptr;
                if (fastSources) {
                    p.opsAffine = opsAffine.p;
                    hipCheck(kt->stageVb(mode, p, vb, vbPartials.p, lamBuf.p, 2, filter ? filterT.p : nullptr, st), what);
                } else if (mfmaSources) {
                    bdg_dev::PhysParams ph{};
                    ph.sx = vb.Hx; ph.sy = vb.Hy; ph.fconst = vb.fcor; ph.cd = vb.cd;
                    ph.slope = 1.0; ph.dragSign = -1.0;
                    ph.H = vb.H; ph.obc = vb.obc; ph.lam = lamBuf.p; ph.spongeField = vb.sponge; ph.tide = vb.tide;
                    p.opsAffine = filter ? opsMfma2SrcFiltered.p : opsMfma2Src.p;
                    hipCheck(kt->stageMfma2Src(mode, p, ph, variantBStateOnce() ? (6 | srcIdentity(filter)) : 2, st), what);
                } else {
                    p.opsAffine = filter ? opsVdFiltered.p : opsVd.p;
                    hipCheck(kt->stageVb(mode, p, vb, vbPartials.p, lamBuf.p, 6, nullptr, st), what);
                }
            } else if (fastSources) {
                // The unrolled kernel also reduces the global speed of the state it writes (for the tide value
                // the next evaluation is expected to see). If this launch reads exactly that state at exactly
                // that tide over the whole mesh, the separate speed pass is skipped.
                p.opsAffine = opsAffine.p;
                const bool whole = p.kbegin == 0 && p.kend == numOwned;
                const bool reuse = whole && lamStateFor == p.qin && lamTideFor == vb.tide && !std::getenv("BDG_SW2D_SPEED_PASS");
                const int cur = lamSlot, nxt = lamSlot ^ 1;
                vb.lam = reuse ? lamPair.p + cur : lamBuf.p;
                vb.lamNext = nullptr;
                lamStateFor = nullptr;
                if (whole && mode != bdg_dev::MODE_RHS) {
                    hipCheck(hipMemsetAsync(lamPair.p + nxt, 0, sizeof(double), st), "hipMemset");
                    vb.lamNext = reinterpret_cast<unsigned long long*>(lamPair.p + nxt);
                    vb.tideNext = tideAt(nextEvalTime);
                    lamStateFor = p.qout;
                    lamTideFor = vb.tideNext;
                    lamSlot = nxt;
                }
                hipCheck(kt->stageVb(mode, p, vb, vbPartials.p, lamBuf.p, reuse ? 2 : 1, filter ? filterT.p : nullptr, st), what);
            } else if (mfmaSources) {
                // N >= 6: speed pass, then the matrix-core kernel with variant B's surface term and sources
                vb.lam = lamBuf.p;
                hipCheck(kt->stageVb(mode, p, vb, vbPartials.p, lamBuf.p, 4, nullptr, st), what);
                bdg_dev::PhysParams ph{};
                ph.sx = vb.Hx; ph.sy = vb.Hy; ph.fconst = vb.fcor; ph.cd = vb.cd;
                ph.slope = 1.0; ph.dragSign = -1.0;             // src/sw2d/main.cpp:461-478
                ph.H = vb.H; ph.obc = vb.obc; ph.lam = lamBuf.p; ph.spongeField = vb.sponge; ph.tide = vb.tide;
                p.opsAffine = filter ? opsMfma2SrcFiltered.p : opsMfma2Src.p;
                hipCheck(kt->stageMfma2Src(mode, p, ph, variantBStateOnce() ? (6 | srcIdentity(filter)) : 2, st), what);
            } else {
                p.opsAffine = filter ? opsVdFiltered.p : opsVd.p;
                hipCheck(kt->stageVb(mode, p, vb, vbPartials.p, lamBuf.p, 0, nullptr, st), what);
            }
        } else if (variantD && fastSources) {
            // three conserved fields on the unrolled kernel with the sources folded in, the tracer (if
            // any) by its own one-field-per-wave launch reading the same input state
            bdg_dev::PhysParams ph{};
            if (vd.sources) {
                ph.sx = vd.zx; ph.sy = vd.zy; ph.fcor = vd.fcor;
                ph.fconst = vd.fconst; ph.cd = vd.cd;
                ph.slope = -1.0; ph.dragSign = 1.0;   // swhelpers/rhs.py:300-309
            }
            // filtered RHS: plain operators, Filter applied to flux terms + sources at the end
            ph.fmat = filter ? filterT.p : nullptr;
            p.opsAffine = opsAffine.p;
            if (nf == 4 && !std::getenv("BDG_SW2D_TRACER_PASS")) {
                hipCheck(kt->stageAffineSrc(mode, p, ph, 1, st), what);    // the tracer rides in the same pass
            } else {
                hipCheck(kt->stageAffineSrc(mode, p, ph, 0, st), what);
                if (nf == 4) { // own pass; the tracer has no sources: pre-filtered operators as in variant A
                    p.opsAffine = filter ? opsAffineFiltered.p : opsAffine.p;
                    hipCheck(kt->stageTracer(mode, p, st), what);
                }
            }
        } else if (variantD && mfmaSources) {
            // N >= 6: three conserved fields with sources on the matrix cores, then the tracer pass
            bdg_dev::PhysParams ph{};
            if (vd.sources) {
                ph.sx = vd.zx; ph.sy = vd.zy; ph.fcor = vd.fcor;
                ph.fconst = vd.fconst; ph.cd = vd.cd;
                ph.slope = -1.0; ph.dragSign = 1.0;   // swhelpers/rhs.py:300-309
            }
            p.opsAffine = filter ? opsMfma2SrcFiltered.p : opsMfma2Src.p;
            // state-once schedule where it exists (sw2d_mfma3src_kernel.hpp: N = 5, 6, 7 with and without the tracer, N = 8 three fields;
            // BDG_SW2D_SOURCES_TWO_WAVE=1 keeps the two-waves-per-SIMD kernels below for A/B runs and cross-checks)
            const bool stateOnceSrc = !std::getenv("BDG_SW2D_SOURCES_TWO_WAVE");
            if (stateOnceSrc && kt->mfma3SrcFields >= nf && !std::getenv("BDG_SW2D_TRACER_PASS") &&
                static_cast<long long>(nf) * Np * ld * 8 <= 4294967295LL) {
                hipCheck(kt->stageMfma2Src(mode, p, ph, (nf == 4 ? 5 : 4) | srcIdentity(filter), st), what);
            } else if (stateOnceSrc && kt->mfma3TracerPhase && nf == 4 && !std::getenv("BDG_SW2D_TRACER_PASS") &&
                       static_cast<long long>(4) * Np * ld * 8 <= 4294967295LL) {
                // N = 8: the tracer equation as a second phase of every tile, from the state tile still in LDS (one launch, state read once)
                hipCheck(kt->stageMfma2Src(mode, p, ph, 7 | srcIdentity(filter), st), what);
            } else if (stateOnceSrc && kt->mfma3SrcFields == 3 && nf == 4 && static_cast<long long>(3) * Np * ld * 8 <= 4294967295LL) {
                // N = 8: three conserved fields with sources on the state-once schedule, the tracer in its own pass
                hipCheck(kt->stageMfma2Src(mode, p, ph, 4 | srcIdentity(filter), st), what);
                p.opsAffine = filter ? opsMfma2Filtered.p : opsMfma2.p;
                hipCheck(kt->stageMfma2Src(mode, p, ph, 1, st), what);
            } else if (nf == 4 && kt->mfmaMT <= 2 && !std::getenv("BDG_SW2D_TRACER_PASS")) {
                hipCheck(kt->stageMfma2Src(mode, p, ph, 3, st), what);     // N <= 6: the tracer rides in the same pass
            } else {
                hipCheck(kt->stageMfma2Src(mode, p, ph, 0, st), what);
                if (nf == 4) {
                    p.opsAffine = filter ? opsMfma2Filtered.p : opsMfma2.p;
                    hipCheck(kt->stageMfma2Src(mode, p, ph, 1, st), what);
                }
            }
        } else if (variantD) {
            p.opsAffine = filter ? opsVdFiltered.p : opsVd.p;
            hipCheck(kt->stageVd(mode, p, vd, st), what);
        } else if (affine && variant == 5) {
            p.opsAffine = filter ? opsMfmaFiltered.p : opsMfma.p;
            hipCheck(kt->stageMfma(mode, p, st), what);
        } else if (affine && variant == 7) {
            p.opsAffine = filter ? opsMfma2Filtered.p : opsMfma2.p;
            hipCheck(kt->stageMfma3(mode, p, st), what);
        } else if (affine && variant == 6) {
            p.opsAffine = filter ? opsMfma2Filtered.p : opsMfma2.p;
            hipCheck(kt->stageMfma2(mode, p, st), what);
        } else if (affine) {
            p.opsAffine = filter ? opsAffineFiltered.p : opsAffine.p;
            hipCheck(kt->stageAffine(mode, variant, p, st), what);
        } else if (nodalMfma) {
            // per-node geometry on the matrix cores (state-once schedule), every order
            p.opsAffine = filter ? opsMfma2NodalFilter.p : opsMfma2.p;
            hipCheck(kt->stageMfma3Nodal(mode, filter, p, st), what);
        } else {
            hipCheck(kt->stage(mode, filter, p, st), what);
        }
    }
    // variant B's stage kernel on the state-once schedule where it exists (sw2d_mfma3src_kernel.hpp, PHYS = 2);
    // BDG_SW2D_SOURCES_TWO_WAVE=1 keeps the two-waves-per-SIMD kernel
    bool variantBStateOnce() const {
        return kt->mfma3SrcFields >= 3 && !std::getenv("BDG_SW2D_SOURCES_TWO_WAVE") &&
               static_cast<long long>(3) * Np * ld * 8 <= 4294967295LL;
    }
    DevBuf<double> opsMfma2NodalFilter; // plain MfmaOps2 image + MT*KV Filter tiles
    bool nodalMfma = false;   // non-affine tables: matrix-core kernel (default) instead of the N <= 6 vector kernel
    bool fastSources = false; // variants B/C/D on the unrolled kernels instead of the rolled ones
    bool lamExternal = false; // variant B, partitioned: lamBuf holds the all-rank speed of the state about to be evaluated
    // up to this order the unrolled source-term kernels are used, above it the matrix-core ones
    static constexpr int kUnrolledSourcesMaxOrder = 4;
    DevBuf<double> filterT;   // [m][i] = Filter[i][m], for filtered source terms
    // :352  hP = HM + amp cos(om t) 1/2 (tanh(ramp (t - T)) + 1)
    double tideAt(double t) const {
        const double om = 2.0 * M_PI / tidePeriod;
        return tideAmp * std::cos(om * t) * 0.5 * (std::tanh(tideRamp * (t - tidePeriod)) + 1);
    }
    DevBuf<double> lamPair;            // two accumulators for the fused next-evaluation speed (alternating)
    int lamSlot = 0;
    const double* lamStateFor = nullptr; // state buffer the accumulated speed belongs to (nullptr: none)
    double lamTideFor = 0.0;
    double nextEvalTime = 0.0;         // model time of the evaluation that will follow the current launch
    // elements below which the matrix-core kernel is the faster one, per order (measured crossovers:
    // N=2 near 10 k, N=3 near 125 k, N=4 between 125 k and 250 k; N=1 never ahead; N=5 runs on
    // the matrix cores at every size)
    // (round 4, after the matrix-core kernel of N <= 4 got all its requests ahead of its first product -- kernel ms, unrolled / matrix cores,
    // profiles/r04_rehearsal_experiments.txt: N=4 125 k elements 0.0543 / 0.0452, 250 k 0.1009 / 0.1142; N=3 125 k 0.0365 / 0.0313, 250 k 0.0612 /
    // 0.0704; N=2 125 k 0.0189 / 0.0283: N=3's crossover moves up to where N=4's is)
    static constexpr int kSmallLaunch[6] = {0, 4000, 10000, 160000, 160000, 0};
    bool variantForced = false;                 // BDG_SW2D_AFFINE_VARIANT given
    // resident-workgroup kernels, interior launch of a partitioned run: CUs left to the boundary kernel (a strip of a few
    // hundred elements = 4..8 four-wave workgroups); N=8, 8-way rehearsal: 0.087 -> see profiles/r02_rehearsal.txt
    static constexpr int kInteriorGridCap = 244;
    // ... and at N >= 5 the strip kernel's workgroups (one 16-element tile each, three waves; two fit a CU, none fits beside an
    // interior workgroup's 120 KB of LDS) need free CUs, or the strip -- which sits on the exchange chain -- runs in many rounds.
    // Round 3 left one CU per strip tile (cap 218-232). Round 4, with the chains meeting inside the kernels, swept the cap in
    // the 8-way rehearsal (profiles/r04_rehearsal_experiments.txt; ms per stage): N=5 244: 0.0580, 250: 0.0521; N=6 244: 0.0486,
    // 248: 0.0464, 252: 0.0584 (four free CUs: the strip's 32 tiles take four rounds); N=7 240: 0.0473, 246: 0.0401, 250: 0.0400;
    // N=8 238: 0.0606 (1930 tiles are three rounds of a tile per wave on 952 waves, two on 968 and more), 242: 0.0546,
    // 244: 0.0540, 246: 0.0542. The interior wants every CU it can get; the strip needs about a round's worth of slots for its
    // tiles and RCCL's kernel a CU: eight free CUs, twelve at N = 8 (its strip tiles carry three row blocks).
    // BDG_SW2D_INTERIOR_CAP=n pins the cap.
    int interiorGridCap() const {
        static const int pinned = [] { const char* e = std::getenv("BDG_SW2D_INTERIOR_CAP"); return e ? std::atoi(e) : 0; }();
        if (pinned > 0) return pinned;
        if (N < 5) return kInteriorGridCap;
        return N >= 8 ? 244 : 248;
    }

    void launchRhs(const double* qin, double* out, bool filter) {
        if (filter && !hasFilter) throw arg_error("filter requested but the solver was created without a Filter matrix");
        bdg_dev::StageParams p = baseParams();
        p.qin = qin;
        p.rhs = out;
        launchStage(bdg_dev::MODE_RHS, filter, p, "sw2d stage kernel <RHS>");
    }

    // part: 0 = interior elements only (no ghost dependency; state not advanced),
    //       1 = partition-boundary elements, then advance; 2 = all owned elements, then advance.
    // done: an event to record when this launch has finished -- through the launch itself where its helper can (one packet
    // on the queue instead of two), by a record behind it otherwise
    void launchLserkStage(int part = 2, hipStream_t on = nullptr, bool advance = true, hipEvent_t done = nullptr, bool flagSync = false) {
        const int s = static_cast<int>(stageCount % blitzdg::LSERK4::numStages);
        bdg_dev::StageParams p = baseParams();
        bool recorded = false;
        unsigned signals = 0;
        if (flagSync && part == 0) { // ring tiles wait for every boundary launch issued so far and signal for themselves
            p.syncWait = syncBuf.p + 1; p.syncWaitValue = expectStrip;
            p.syncSignal = syncBuf.p; p.syncError = reinterpret_cast<unsigned int*>(syncBuf.p + 2);
            p.syncFirstTile = ringBegin / 16;
            p.syncSignalsOut = &signals;
        }
        static const bool extLaunch = [] { const char* e = std::getenv("BDG_SW2D_EXT_LAUNCH"); return !e || e[0] != '0'; }();
        if (done && extLaunch) { p.stopEvent = done; p.stopEventUsed = &recorded; }
        if (part == 0) {
            p.kend = numInterior;
            p.gridCap = interiorGridCap(); // the boundary kernel runs beside this launch (exchange stream)
        }
        if (part == 1) p.kbegin = numInterior;
        p.qin = qcur;
        p.qout = qalt;
        p.res = res.p;
        p.ca = blitzdg::LSERK4::rk4a[s];
        p.cb = blitzdg::LSERK4::rk4b[s];
        p.cc = dtStage;
        // model time (the tide phase of variant B) is frozen over the five stages and moves on after the last
        const bool lastOfStep = s == blitzdg::LSERK4::numStages - 1;
        nextEvalTime = lastOfStep ? timeNow + dtStage : timeNow;
        launchStage(bdg_dev::MODE_LSERK, false, p, "sw2d stage kernel <LSERK>", on);
        expectRing += signals;
        if (done && !recorded) hipCheck(hipEventRecord(done, on ? on : stream), "hipEventRecord");
        if (part == 0 || !advance) return;
        std::swap(qcur, qalt);
        ++stageCount;
        if (lastOfStep) timeNow += dtStage;
    }
    double dtStage = 0.0;

    // `state`: the planes whose boundary elements are packed / whose ghost columns are filled (default: the current state)
    void launchPack(double* buf, hipStream_t on = nullptr, const double* state = nullptr) {
        if (numSend == 0) return;
        const int rows = nf * Np;
        const long long n = static_cast<long long>(numSend) * rows;
        hipLaunchKernelGGL(bdg_dev::halo_pack_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
                           on ? on : stream, state ? state : qcur, buf, sendSlots.p, numSend, rows, ld);
        hipCheck(hipGetLastError(), "halo_pack_kernel");
    }
    void launchUnpack(const double* buf, hipStream_t on = nullptr, double* state = nullptr) {
        const int ghosts = K - numOwned;
        if (ghosts == 0) return;
        const int rows = nf * Np;
        const long long n = static_cast<long long>(ghosts) * rows;
        hipLaunchKernelGGL(bdg_dev::halo_unpack_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
                           on ? on : stream, state ? state : qcur, buf, numOwned, ghosts, rows, ld);
        hipCheck(hipGetLastError(), "halo_unpack_kernel");
    }

    // ---- plain (not overlapped) exchange for the steppers whose RHS needs more than the neighbours' traces of the
    //      previous stage: every evaluation of the midpoint / Heun schemes reads another state, and variant B also
    //      needs ONE Lax-Friedrichs speed over all ranks (reference src/sw2d/main.cpp:414) before any element starts.
    void exchangeGhostsOf(double* state) {
        if (!comm) throw arg_error("no communicator: call bdg_sw2d_comm_init first");
        const int rows = nf * Np;
        launchPack(sendBuf.p, stream, state);
        if (!peers.empty()) {
            RcclApi& nc = rccl();
            ncclCheck(nc.GroupStart(), "ncclGroupStart");
            for (const Peer& pr : peers) {
                if (pr.recvCount > 0)
                    ncclCheck(nc.Recv(recvBuf.p + static_cast<size_t>(pr.recvStart) * rows, static_cast<size_t>(pr.recvCount) * rows,
                                      ncclDouble, pr.rank, comm, stream), "ncclRecv");
                if (pr.sendCount > 0)
                    ncclCheck(nc.Send(sendBuf.p + static_cast<size_t>(pr.sendStart) * rows, static_cast<size_t>(pr.sendCount) * rows,
                                      ncclDouble, pr.rank, comm, stream), "ncclSend");
            }
            ncclCheck(nc.GroupEnd(), "ncclGroupEnd");
        }
        launchUnpack(recvBuf.p, stream, state);
    }
    // variant B: speed of `state` over this rank's owned elements (their '+' traces include the ghosts just received),
    // then the maximum over all ranks, left in lamBuf on the device: one 8-byte all-reduce per RHS evaluation
    void globalSpeedOf(const double* state) {
        bdg_dev::StageParams p = baseParams();
        p.qin = state;
        vb.tide = tideAt(timeNow);
        vb.lam = lamBuf.p;
        hipCheck(kt->stageVb(bdg_dev::MODE_RHS, p, vb, vbPartials.p, lamBuf.p, 4, nullptr, stream), "sw2d_vb_speed_kernel");
        if (comm && commWorld > 1)
            ncclCheck(rccl().AllReduce(lamBuf.p, lamBuf.p, 1, ncclDouble, ncclMax, comm, stream), "ncclAllReduce");
    }
    // one evaluation of a partitioned run: ghosts of `state`, the all-rank speed (variant B), then fn launches the stage
    template <typename Fn>
    void evaluateExchanged(double* state, Fn&& fn) {
        exchangeGhostsOf(state);
        if (variantB) {
            globalSpeedOf(state);
            lamExternal = true;
        }
        try {
            fn();
        } catch (...) {
            lamExternal = false;
            throw;
        }
        lamExternal = false;
    }

    // True when the partition-boundary launch of an exchanged stage can do the halo staging itself (three-field
    // straight-sided solver whose boundary strip runs on a matrix-core kernel).
    bool halosFold() const {
        if (!haloFusable || !affine || variantB || variantD || variantForced || std::getenv("BDG_SW2D_HALO_KERNELS")) return false;
        return N >= 5 || numOwned - numInterior < kSmallLaunch[N]; // the strip runs on a matrix-core kernel
    }
    // True when the two chains of an exchanged stage can meet through device counters polled inside the kernels instead of
    // through events on the queues (BDG_SW2D_EVENT_SYNC=1 keeps the events): folded halo staging, and both launches of a stage
    // on kernels that have a SYNC instance -- the interior on the matrix-core kernel of its order, the strip on the latency form.
    bool flagSyncUsable() const {
        const char* eventsPinned = std::getenv("BDG_SW2D_EVENT_SYNC"); // (read per call: the tests switch it within one process)
        if ((eventsPinned && eventsPinned[0] != '0') || !syncBuf.p || !halosFold()) return false;
        if (std::getenv("BDG_SW2D_STRIP_THROUGHPUT") || std::getenv("BDG_SW2D_HALO_VARIANT")) return false;
        if (N >= 5) return affineVariant == 7 && (numOwned - numInterior + 15) / 16 <= 1024;
        // N <= 4: only while the interior share is small enough for the matrix-core kernel (8-way split of C3). A SYNC instance of
        // the unrolled kernel (the wait in front of its one big basic block, ring waves storing write-through) was built and
        // measured in the 2- and 4-way rehearsal: 0.332 / 0.164 ms per stage against 0.220 / 0.121 with the events -- the branch
        // at the top costs that kernel its load batching (profiles/r04_rehearsal_experiments.txt); larger shares keep the events.
        static const int smallPinned = [] { const char* e = std::getenv("BDG_SW2D_SMALL_LAUNCH"); return e ? std::atoi(e) : -1; }();
        return numInterior > 0 && affineVariant == 0 && numInterior < (smallPinned >= 0 ? smallPinned : kSmallLaunch[N]);
    }
    // a bounded in-kernel wait that gave up (sync_wait) left a mark: report it the next time the host looks at the device
    // unfiltered evaluation on a state-once kernel with sources: F' is the identity, the sources are added pointwise (IDF instance;
    // BDG_SW2D_SOURCES_PRODUCT=1 keeps the products with the identity tiles for A/B runs and cross-checks -- bit-identical)
    static int srcIdentity(bool filter) {
        const char* product = std::getenv("BDG_SW2D_SOURCES_PRODUCT"); // (read per call: the cross-check switches it within one process)
        return (!filter && !(product && product[0] != '0')) ? bdg_dev::kSrcIdentity : 0;
    }
    bool takeSyncMark() { // true (and the mark cleared) if a wait of this device gave up since the last look
        if (!syncBuf.p) return false;
        unsigned long long mark = 0;
        hipCheck(hipMemcpy(&mark, syncBuf.p + 2, sizeof(mark), hipMemcpyDeviceToHost), "sync word download");
        if (mark != 0) hipCheck(hipMemset(syncBuf.p + 2, 0, sizeof(mark)), "hipMemset");
        return mark != 0;
    }
    static const char* syncErrorText() {
        return "an in-kernel wait between the interior and the partition-boundary launch of an exchanged stage "
               "timed out: the results of this run are not valid (BDG_SW2D_EVENT_SYNC=1 restores event waits)";
    }
    void checkSyncError() {
        if (takeSyncMark()) throw std::runtime_error(syncErrorText());
    }
    // LSERK4 stage of the partition-boundary elements: reads ghost traces from recv, writes send records
    // ringExpected (flagSync): the ring-tile count the interior launch of the PREVIOUS stage brings the counter to
    void launchBoundaryStageFolded(hipStream_t on, const double* recv, double* send, hipEvent_t done = nullptr, bool flagSync = false,
                                   unsigned long long ringExpected = 0) {
        const int st = static_cast<int>(stageCount % blitzdg::LSERK4::numStages);
        bdg_dev::StageParams p = baseParams();
        bool recorded = false;
        unsigned signals = 0;
        if (flagSync) {
            p.syncWait = syncBuf.p; p.syncWaitValue = ringExpected;
            p.syncSignal = syncBuf.p + 1; p.syncError = reinterpret_cast<unsigned int*>(syncBuf.p + 2);
            p.syncFirstTile = 0;
            p.syncSignalsOut = &signals;
        }
        static const bool extLaunch = [] { const char* e = std::getenv("BDG_SW2D_EXT_LAUNCH"); return !e || e[0] != '0'; }();
        if (done && extLaunch) { p.stopEvent = done; p.stopEventUsed = &recorded; }
        p.kbegin = numInterior;
        p.qin = qcur; p.qout = qalt; p.res = res.p;
        p.ca = blitzdg::LSERK4::rk4a[st]; p.cb = blitzdg::LSERK4::rk4b[st]; p.cc = dtStage;
        p.haloRecv = recv; p.haloSend = send; p.haloSendOf = haloSendOf.p;
        p.haloOwned = numOwned; p.haloRows = nf * Np;
        if (N >= 5) { // the kernel family the interior (and a single-domain run) uses: bit-identical arithmetic
            p.opsAffine = opsMfma2.p;
            static const char* haloVariant = std::getenv("BDG_SW2D_HALO_VARIANT"); // A/B switch: "6" = two-waves schedule
            const bool two = affineVariant == 6 || (haloVariant && haloVariant[0] == '6');
            hipCheck(two ? kt->stageMfma2Halo(p, on) : kt->stageMfma3Halo(p, on), "sw2d boundary stage kernel <LSERK, halo>");
        } else {
            p.opsAffine = opsMfma.p;
            hipCheck(kt->stageMfmaHalo(p, on), "sw2d boundary stage kernel <LSERK, halo>");
        }
        expectStrip += signals;
        if (done && !recorded) hipCheck(hipEventRecord(done, on), "hipEventRecord");
        std::swap(qcur, qalt);
        ++stageCount;
        if (st == blitzdg::LSERK4::numStages - 1) timeNow += dtStage;
    }

    // LSERK4 stages of a partitioned run, all on the device, as two concurrent chains:
    //   compute stream  A:  wait B(s-1) -> [interior elements of stage s] -> signal A(s)
    //   exchange stream B:  pack(s) -> grouped ncclSend/ncclRecv with every neighbour -> unpack(s)
    //                       -> wait A(s-1) -> [partition-boundary elements of stage s] -> signal B(s)
    // The only true dependency loop is boundary -> transfer -> boundary on chain B; the interior
    // elements (99 % of the work) run beside it and only meet it one stage later:
    //   interior(s) reads boundary elements' state of stage s-1 and overwrites the buffer boundary(s-1)
    //   read, hence waits for B(s-1); boundary(s) reads interior neighbours' state of stage s-1 and
    //   overwrites slots interior(s-1) read, hence waits for A(s-1). Within B, stream order protects
    //   the send / receive buffers and the ghost slots. Events alternate by stage parity so that a
    //   wait issued for stage s-1 is not re-armed by the record of stage s.
    void launchLserkStagesExchanged(double dt, int numStages) {
        if (!comm) throw arg_error("no communicator: call bdg_sw2d_comm_init first");
        if (numStages <= 0) return;
        const int rows = nf * Np;
        dtStage = dt;
        if (variantB) { // the all-rank speed has to exist before any element of the stage starts: no overlap to be had
            for (int i = 0; i < numStages; ++i) evaluateExchanged(qcur, [&] { launchLserkStage(2); });
            return;
        }
        // chain B starts after everything already queued on A (state upload, earlier steps)
        hipCheck(hipEventRecord(evA[1], stream), "hipEventRecord");
        hipCheck(hipStreamWaitEvent(commStream, evA[1], 0), "hipStreamWaitEvent");
        bool haveA = false, haveB = false;
        // With the staging folded into the boundary kernel, chain B is: exchange -> boundary kernel (which reads
        // the received records and writes the next send records); only the very first exchange needs a pack.
        const bool fold = halosFold();
        // Round 4: where both launches of a stage have a SYNC instance the two chains meet INSIDE the kernels. The elements are
        // ordered [deep interior | ring | boundary | ghost]; only the ring -- the interior elements next to the partition
        // boundary -- reads what boundary(s-1) wrote and overwrites what it read, and only the boundary elements read what the ring
        // tiles of interior(s-1) wrote. So interior(s)'s ring tiles (its last ones) poll a counter that boundary(s-1)'s workgroups
        // raise, boundary(s) polls the counter interior(s-1)'s ring tiles raise, and neither queue carries a wait or a record
        // for the other any more: the interior launches run back to back on A, exchange and boundary launch back to back on B.
        // No deadlock: a launch only ever waits for a launch that was queued earlier on the other stream and whose own waits
        // are for launches queued earlier still (down to counts that already hold when the loop starts); the waiting ring tiles
        // are a few dozen waves, so the boundary launch and RCCL's kernel always find free compute units (interiorGridCap keeps
        // them free at N >= 5), and every wait is bounded (sync_wait) -- a wave that gives up marks the run invalid.
        const bool flags = fold && flagSyncUsable();
        if (fold) launchPack(sendBuf.p, commStream);
        for (int i = 0; i < numStages; ++i) {
            const int cur = i & 1, prev = cur ^ 1;
            const bool last = i == numStages - 1;
            const unsigned long long ringBefore = expectRing; // what interior(i-1) and everything before it brings the counter to
            // ---- chain A
            if (!flags && haveB) hipCheck(hipStreamWaitEvent(stream, evB[prev], 0), "hipStreamWaitEvent");
            launchLserkStage(0, nullptr, true, flags ? nullptr : evA[cur], flags);
            // ---- chain B
            if (!fold) launchPack(sendBuf.p, commStream);
            if (!peers.empty()) {
                RcclApi& nc = rccl();
                ncclCheck(nc.GroupStart(), "ncclGroupStart");
                for (const Peer& pr : peers) {
                    if (pr.recvCount > 0)
                        ncclCheck(nc.Recv(recvBuf.p + static_cast<size_t>(pr.recvStart) * rows,
                                          static_cast<size_t>(pr.recvCount) * rows, ncclDouble, pr.rank, comm, commStream),
                                  "ncclRecv");
                    if (pr.sendCount > 0)
                        ncclCheck(nc.Send(sendBuf.p + static_cast<size_t>(pr.sendStart) * rows,
                                          static_cast<size_t>(pr.sendCount) * rows, ncclDouble, pr.rank, comm, commStream),
                                  "ncclSend");
                }
                ncclCheck(nc.GroupEnd(), "ncclGroupEnd");
            }
            if (!fold) launchUnpack(recvBuf.p, commStream);
            if (!flags && haveA) hipCheck(hipStreamWaitEvent(commStream, evA[prev], 0), "hipStreamWaitEvent");
            if (fold) launchBoundaryStageFolded(commStream, recvBuf.p, sendBuf.p, (!flags || last) ? evB[cur] : nullptr, flags, ringBefore);
            else launchLserkStage(1, commStream, true, evB[cur]); // partition-boundary elements, advance
            haveA = haveB = true;
        }
        // later work on A (dt reduction, downloads, plain stages) sees the last boundary update
        hipCheck(hipStreamWaitEvent(stream, evB[(numStages - 1) & 1], 0), "hipStreamWaitEvent");
    }

    // max (or min) of one double over all ranks, through the device
    double allReduceScalar(double v, bool takeMax, bool sum = false) {
        if (!comm || commWorld == 1) return v;
        hipCheck(hipMemcpyAsync(scalarBuf.p, &v, sizeof(double), hipMemcpyHostToDevice, stream), "H2D copy");
        ncclCheck(rccl().AllReduce(scalarBuf.p, scalarBuf.p + 1, 1, ncclDouble, sum ? ncclSum : (takeMax ? ncclMax : ncclMin),
                                   comm, stream), "ncclAllReduce");
        double out = 0.0;
        hipCheck(hipMemcpyAsync(&out, scalarBuf.p + 1, sizeof(double), hipMemcpyDeviceToHost, stream), "D2H copy");
        hipCheck(hipStreamSynchronize(stream), "allreduce sync");
        return out;
    }

    // q1 = q + dt/2 R(q);  q = q + dt R(q1)   (reference src/sw2d-simple/main.cpp:132-151)
    void launchRk2Step(double dt, bool filter) {
        if (filter && !hasFilter) throw arg_error("filter requested but the solver was created without a Filter matrix");
        bdg_dev::StageParams p = baseParams();
        p.qin = qcur; p.qbase = qcur; p.qout = aux.p;
        p.ca = 1.0; p.cb = 0.0; p.cc = 0.5 * dt;
        nextEvalTime = timeNow;
        launchStage(bdg_dev::MODE_COMBINE, filter, p, "sw2d stage kernel <COMBINE>");
        p.qin = aux.p; p.qbase = qcur; p.qout = qalt;
        p.ca = 1.0; p.cb = 0.0; p.cc = dt;
        nextEvalTime = timeNow + dt;
        launchStage(bdg_dev::MODE_COMBINE, filter, p, "sw2d stage kernel <COMBINE>");
        std::swap(qcur, qalt);
        timeNow += dt;
    }

    // The same two schemes in a partitioned run: each evaluation first refreshes the ghost columns of the state it reads
    // (and, variant B, reduces the Lax-Friedrichs speed over all ranks).
    void launchRk2StepExchanged(double dt, bool filter) {
        if (filter && !hasFilter) throw arg_error("filter requested but the solver was created without a Filter matrix");
        bdg_dev::StageParams p = baseParams();
        p.qin = qcur; p.qbase = qcur; p.qout = aux.p;
        p.ca = 1.0; p.cb = 0.0; p.cc = 0.5 * dt;
        nextEvalTime = timeNow;
        evaluateExchanged(qcur, [&] { launchStage(bdg_dev::MODE_COMBINE, filter, p, "sw2d stage kernel <COMBINE>"); });
        p.qin = aux.p; p.qbase = qcur; p.qout = qalt;
        p.ca = 1.0; p.cb = 0.0; p.cc = dt;
        nextEvalTime = timeNow + dt;
        evaluateExchanged(aux.p, [&] { launchStage(bdg_dev::MODE_COMBINE, filter, p, "sw2d stage kernel <COMBINE>"); });
        std::swap(qcur, qalt);
        timeNow += dt;
    }
    void launchSspRk2StepExchanged(double dt, bool filter, double spongeCoeff) {
        bdg_dev::StageParams p = baseParams();
        p.sponge = spongeCoeff;
        p.qin = qcur; p.qbase = qcur; p.qout = aux.p;
        p.ca = 1.0; p.cb = 0.0; p.cc = dt;
        nextEvalTime = timeNow;
        evaluateExchanged(qcur, [&] { launchStage(bdg_dev::MODE_COMBINE, filter, p, "sw2d stage kernel <COMBINE>"); });
        p.qin = aux.p; p.qbase = qcur; p.qout = qalt;
        p.ca = 0.5; p.cb = 0.5; p.cc = 0.5 * dt;
        nextEvalTime = timeNow + dt;
        evaluateExchanged(aux.p, [&] { launchStage(bdg_dev::MODE_COMBINE, filter, p, "sw2d stage kernel <COMBINE>"); });
        std::swap(qcur, qalt);
        timeNow += dt;
    }

    // SSP-RK2 (Heun) of the reference's variant-B driver (src/sw2d/main.cpp:211-235), sponge optional:
    //   q1 = sponge(q + dt R(q));   q = sponge(1/2 (q + q1 + dt R(q1)))
    void launchSspRk2Step(double dt, bool filter, double spongeCoeff) {
        bdg_dev::StageParams p = baseParams();
        p.sponge = spongeCoeff;
        p.qin = qcur; p.qbase = qcur; p.qout = aux.p;
        p.ca = 1.0; p.cb = 0.0; p.cc = dt;
        nextEvalTime = timeNow;                               // second evaluation: same time level (:225)
        launchStage(bdg_dev::MODE_COMBINE, filter, p, "sw2d stage kernel <COMBINE>");
        p.qin = aux.p; p.qbase = qcur; p.qout = qalt;
        p.ca = 0.5; p.cb = 0.5; p.cc = 0.5 * dt;
        nextEvalTime = timeNow + dt;                          // first evaluation of the next step
        launchStage(bdg_dev::MODE_COMBINE, filter, p, "sw2d stage kernel <COMBINE>");
        std::swap(qcur, qalt);
        timeNow += dt;
    }

    // Row-wise operator image of the per-node-geometry form of variants B / C / D (VnOps: row i = Dr[i][.], Ds[i][.] interleaved,
    // then Lift[i][.]), the Filter rows of its second pass and the scratch planes of the unfiltered rows
    void buildNodalVariantOps() {
        if (opsVn.p) return;
        const int row = 2 * Np + NFN;
        std::vector<double> img(static_cast<size_t>(row) * Np);
        for (int i = 0; i < Np; ++i) {
            for (int m = 0; m < Np; ++m) {
                img[static_cast<size_t>(i) * row + 2 * m] = hostDr[static_cast<size_t>(i) * Np + m];
                img[static_cast<size_t>(i) * row + 2 * m + 1] = hostDs[static_cast<size_t>(i) * Np + m];
            }
            for (int j = 0; j < NFN; ++j) img[static_cast<size_t>(i) * row + 2 * Np + j] = hostLift[static_cast<size_t>(i) * NFN + j];
        }
        opsVn.alloc(img.size(), bytes);
        hipCheck(hipMemcpy(opsVn.p, img.data(), img.size() * sizeof(double), hipMemcpyHostToDevice), "nodal variant ops upload");
        if (!hostFilter.empty()) {
            filterRows.alloc(hostFilter.size(), bytes);
            hipCheck(hipMemcpy(filterRows.p, hostFilter.data(), hostFilter.size() * sizeof(double), hipMemcpyHostToDevice), "filter upload");
            vnRaw.alloc(static_cast<size_t>(nf) * planeSize(), bytes);
        }
    }

    // [m][i]{Dr'[i][m], Ds'[i][m], F'[i][m]} + Lift' images (F' = I or Filter) for the rolled
    // one-field-per-wave kernels (variants B and D)
    void buildSourceOps() {
        if (opsVd.p) return;
        auto image = [&](const double* Dr, const double* Ds, const double* F, const double* Lift) {
            std::vector<double> img(static_cast<size_t>(3) * Np * Np + static_cast<size_t>(NFN) * Np);
            for (int m = 0; m < Np; ++m)
                for (int i = 0; i < Np; ++i) {
                    const size_t o = 3 * (static_cast<size_t>(m) * Np + i);
                    img[o] = Dr[i * Np + m];
                    img[o + 1] = Ds[i * Np + m];
                    img[o + 2] = F ? F[i * Np + m] : (i == m ? 1.0 : 0.0);
                }
            for (int j = 0; j < NFN; ++j)
                for (int i = 0; i < Np; ++i)
                    img[static_cast<size_t>(3) * Np * Np + static_cast<size_t>(j) * Np + i] = Lift[i * NFN + j];
            return img;
        };
        const std::vector<double> plain = image(hostDr.data(), hostDs.data(), nullptr, hostLift.data());
        opsVd.alloc(plain.size(), bytes);
        hipCheck(hipMemcpy(opsVd.p, plain.data(), plain.size() * sizeof(double), hipMemcpyHostToDevice), "source ops upload");
        if (!hostFilter.empty()) {
            const std::vector<double> FDr = matmulHost(hostFilter.data(), hostDr.data(), Np, Np),
                                      FDs = matmulHost(hostFilter.data(), hostDs.data(), Np, Np),
                                      FL = matmulHost(hostFilter.data(), hostLift.data(), Np, NFN);
            const std::vector<double> filt = image(FDr.data(), FDs.data(), hostFilter.data(), FL.data());
            opsVdFiltered.alloc(filt.size(), bytes);
            hipCheck(hipMemcpy(opsVdFiltered.p, filt.data(), filt.size() * sizeof(double), hipMemcpyHostToDevice),
                     "filtered source ops upload");
        }
    }

    // Operator image of the matrix-core kernel with source terms (variants B/C/D at N >= 6): MfmaOps2 layout --
    // Dr tiles [r][t], Ds tiles, lift tiles [r][f][tf] -- followed by MT*KV tiles of F' (identity, or Filter with
    // the other operators pre-multiplied by it). Lane l of a tile holds A[16r + (l&15)][4t + (l>>4)].
    void buildMfma2SourceOps() {
        if (opsMfma2Src.p) return;
        const int MT = kt->mfmaMT, KV = kt->mfmaKV, KF = kt->mfma2KF;
        auto image = [&](const double* Dr, const double* Ds, const double* Lift, const double* F) {
            const size_t tilesD = static_cast<size_t>(MT) * KV * 64, offF = static_cast<size_t>(kt->mfma2OpsDoubles);
            std::vector<double> img(offF + tilesD, 0.0);
            for (int r = 0; r < MT; ++r)
                for (int l = 0; l < 64; ++l) {
                    const int i = 16 * r + (l & 15);
                    if (i >= Np) continue;
                    for (int t = 0; t < KV; ++t) {
                        const int k = 4 * t + (l >> 4);
                        if (k >= Np) continue;
                        const size_t at = (static_cast<size_t>(r) * KV + t) * 64 + l;
                        img[at] = Dr[i * Np + k];
                        img[tilesD + at] = Ds[i * Np + k];
                        img[offF + at] = F ? F[i * Np + k] : (i == k ? 1.0 : 0.0);
                    }
                    for (int f = 0; f < 3; ++f)
                        for (int tf = 0; tf < KF; ++tf) {
                            const int n = 4 * tf + (l >> 4);
                            if (n < Nfp)
                                img[2 * tilesD + ((static_cast<size_t>(r) * 3 + f) * KF + tf) * 64 + l] = Lift[i * NFN + f * Nfp + n];
                        }
                }
            return img;
        };
        const std::vector<double> plain = image(hostDr.data(), hostDs.data(), hostLift.data(), nullptr);
        opsMfma2Src.alloc(plain.size(), bytes);
        hipCheck(hipMemcpy(opsMfma2Src.p, plain.data(), plain.size() * sizeof(double), hipMemcpyHostToDevice),
                 "mfma2 source ops upload");
        if (!hostFilter.empty()) {
            const std::vector<double> FDr = matmulHost(hostFilter.data(), hostDr.data(), Np, Np),
                                      FDs = matmulHost(hostFilter.data(), hostDs.data(), Np, Np),
                                      FL = matmulHost(hostFilter.data(), hostLift.data(), Np, NFN);
            const std::vector<double> filt = image(FDr.data(), FDs.data(), FL.data(), hostFilter.data());
            opsMfma2SrcFiltered.alloc(filt.size(), bytes);
            hipCheck(hipMemcpy(opsMfma2SrcFiltered.p, filt.data(), filt.size() * sizeof(double), hipMemcpyHostToDevice),
                     "filtered mfma2 source ops upload");
        }
    }

    const double* uploadPlane(const double* host, DevBuf<double>& buf) {
        if (!host) return nullptr;
        if (!buf.p) buf.alloc(planeSize(), bytes);
        hipCheck(hipMemsetAsync(buf.p, 0, buf.n * sizeof(double), stream), "hipMemset");
        uploadRows(host, buf.p, Np);
        return buf.p;
    }

    // Returns {max |Fscale|*spd, max |eta|}; NaN if any entry is NaN.
    void reduceDt(double out[2]) {
        const int nblocks = (numOwned + 255) / 256;
        hipCheck(kt->dt(qcur, fscaleNodal, hasH ? Hbuf.p : nullptr, ld, numOwned, g, partials.p,
                        stream), "sw2d_dt_kernel");
        hipLaunchKernelGGL(bdg_dev::sw2d_reduce_kernel, dim3(1), dim3(256), 0, stream, partials.p, nblocks, red2.p);
        hipCheck(hipGetLastError(), "sw2d_reduce_kernel");
        hipCheck(hipMemcpyAsync(out, red2.p, 2 * sizeof(double), hipMemcpyDeviceToHost, stream), "D2H copy");
        hipCheck(hipStreamSynchronize(stream), "reduce sync");
    }
};

namespace {

// Breadth-first (Cuthill-McKee style) renumbering from the face-neighbour graph:
// neighbours end up within O(sqrt(K)) slots of each other, so the trace gather of
// a wave hits lines its own or nearby waves stream. perm[k] = device slot.
std::vector<int> bfsOrder(const int* vmapP, int K, int Np, int Nfp) {
    std::vector<int> perm(K, -1);
    int next = 0;
    std::vector<int> frontier, nextFrontier;
    for (int seed = 0; seed < K; ++seed) {
        if (perm[seed] >= 0) continue;
        perm[seed] = next++;
        frontier.assign(1, seed);
        while (!frontier.empty()) {
            nextFrontier.clear();
            for (int k : frontier)
                for (int f = 0; f < 3; ++f) {
                    const int k2 = vmapP[(static_cast<size_t>(k) * 3 + f) * Nfp] / Np;
                    if (k2 >= 0 && k2 < K && perm[k2] < 0) {
                        perm[k2] = next++;
                        nextFrontier.push_back(k2);
                    }
                }
            frontier.swap(nextFrontier);
        }
    }
    return perm;
}

// True when the metric terms are constant per element and the face terms constant per
// face (straight-sided elements), to round-off: then one value per element/face suffices.
// The tables themselves carry round-off from Dr*x on small elements (relative ~1e-16 * |Dr| / h:
// 5e-11 at N=8 on a 500x250-cell box), so the test is 1e-8: far above that noise, far below any
// real curvature.
bool geometryIsAffine(const bdg_sw2d_desc& d, int Np, int Nfp, int K) {
    const double tol = 1e-8;
    std::atomic<bool> ok{true};
    blitzdg::detail::parallelFor(K, [&](int k) {
        const double scale = std::fabs(d.rx[k]) + std::fabs(d.sx[k]) + std::fabs(d.ry[k]) + std::fabs(d.sy[k]);
        for (int n = 1; n < Np; ++n) {
            const size_t o = static_cast<size_t>(n) * K + k;
            if (std::fabs(d.rx[o] - d.rx[k]) > tol * scale || std::fabs(d.sx[o] - d.sx[k]) > tol * scale ||
                std::fabs(d.ry[o] - d.ry[k]) > tol * scale || std::fabs(d.sy[o] - d.sy[k]) > tol * scale)
                ok = false;
        }
        for (int f = 0; f < 3; ++f) {
            const size_t o0 = static_cast<size_t>(f) * Nfp * K + k;
            for (int n = 1; n < Nfp; ++n) {
                const size_t o = o0 + static_cast<size_t>(n) * K;
                if (std::fabs(d.nx[o] - d.nx[o0]) > tol || std::fabs(d.ny[o] - d.ny[o0]) > tol ||
                    std::fabs(d.Fscale[o] - d.Fscale[o0]) > tol * std::fabs(d.Fscale[o0]))
                    ok = false;
            }
        }
    });
    return ok;
}

// C = A (n x n) * B (n x c), row-major.
std::vector<double> matmulHost(const double* A, const double* B, int n, int c) {
    std::vector<double> C(static_cast<size_t>(n) * c, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < c; ++j) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += A[i * n + k] * B[k * c + j];
            C[static_cast<size_t>(i) * c + j] = s;
        }
    return C;
}

// AffineOps<N> image: [m][i]{Dr[i][m], Ds[i][m]} then [j][i] Lift[i][j].
std::vector<double> affineOpsImage(const double* Dr, const double* Ds, const double* Lift, int Np, int NFN) {
    std::vector<double> img(static_cast<size_t>(2) * Np * Np + static_cast<size_t>(NFN) * Np);
    for (int m = 0; m < Np; ++m)
        for (int i = 0; i < Np; ++i) {
            img[2 * (static_cast<size_t>(m) * Np + i)] = Dr[i * Np + m];
            img[2 * (static_cast<size_t>(m) * Np + i) + 1] = Ds[i * Np + m];
        }
    for (int j = 0; j < NFN; ++j)
        for (int i = 0; i < Np; ++i) img[static_cast<size_t>(2) * Np * Np + static_cast<size_t>(j) * Np + i] = Lift[i * NFN + j];
    return img;
}

bdg_sw2d* createSolver(const bdg_sw2d_desc& d) {
    if (d.order < 1 || d.order > BDG_SW2D_MAX_ORDER)
        throw arg_error("bdg_sw2d_create: order must be 1.." + std::to_string(BDG_SW2D_MAX_ORDER) +
                        " for the register-resident kernels");
    if (d.num_elements < 1) throw arg_error("bdg_sw2d_create: num_elements must be >= 1");
    if (!d.Dr || !d.Ds || !d.Lift || !d.rx || !d.sx || !d.ry || !d.sy || !d.nx || !d.ny || !d.Fscale || !d.vmapP)
        throw arg_error("bdg_sw2d_create: a required table pointer is NULL");
    if (d.num_wall < 0 || (d.num_wall > 0 && !d.mapW)) throw arg_error("bdg_sw2d_create: bad wall-node list");

    const bdg_dev::KernelTable* kt = bdg_dev::kernel_table(d.order);
    if (!kt) throw arg_error("bdg_sw2d_create: no kernels compiled for this order");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        throw hip_error("bdg_sw2d_create: no HIP device available (the sw2d path has no CPU fallback)");
    if (d.device < 0 || d.device >= ndev) throw arg_error("bdg_sw2d_create: device ordinal out of range");

    auto s = std::unique_ptr<bdg_sw2d>(new bdg_sw2d());
    s->kt = kt;
    s->N = d.order; s->Np = kt->Np; s->Nfp = kt->Nfp; s->NFN = 3 * kt->Nfp; s->K = d.num_elements;
    s->device = d.device;
    s->g = d.g;
    if (d.num_fields != 0 && d.num_fields != 3 && d.num_fields != 4)
        throw arg_error("bdg_sw2d_create: num_fields must be 3 or 4");
    s->nf = d.num_fields == 4 ? 4 : 3;
    s->variantD = s->nf == 4 || d.sources != 0;
    s->ld = (static_cast<long long>(s->K) + 63) / 64 * 64;
    s->numInterior = s->numOwned = s->K;
    const int Np = s->Np, Nfp = s->Nfp, NFN = s->NFN, K = s->K;
    const long long ld = s->ld;
    // Lane addresses are a wave-uniform row pointer plus a 32-bit unsigned BYTE offset (8 * node offset).
    if (static_cast<long long>(Np) * ld * 8 > 4294967295LL)
        throw arg_error("bdg_sw2d_create: Np*K exceeds 2^29 nodes (32-bit byte offsets within a field): partition the "
                        "mesh over more devices");

    // ---- validate the index tables on the host before anything touches the GPU
    const size_t nFaceNodes = static_cast<size_t>(NFN) * K;
    if (d.vmapM)
        for (int k = 0; k < K; ++k)
            for (int f = 0; f < 3; ++f)
                for (int n = 0; n < Nfp; ++n)
                    if (d.vmapM[(static_cast<size_t>(k) * 3 + f) * Nfp + n] != kt->fmask(f, n) + Np * k)
                        throw arg_error("bdg_sw2d_create: vmapM does not follow the warp&blend face-node ordering "
                                        "(Fmask) these kernels are specialised for");
    const long long totalNodes = static_cast<long long>(Np) * K;
    for (size_t i = 0; i < nFaceNodes; ++i)
        if (d.vmapP[i] < 0 || d.vmapP[i] >= totalNodes) throw arg_error("bdg_sw2d_create: vmapP entry out of range");
    for (int i = 0; i < d.num_wall; ++i)
        if (d.mapW[i] < 0 || static_cast<size_t>(d.mapW[i]) >= nFaceNodes)
            throw arg_error("bdg_sw2d_create: wall-node index out of range");

    // Element numbering: the trace gather is cheap only when face neighbours sit within an L2-sized
    // window of slots. Forced by BDG_SW2D_REORDER, suppressed by BDG_SW2D_KEEP_ORDER, otherwise
    // decided from the mean neighbour distance (shuffled 10^6-triangle box: 1.11 ms as given,
    // 0.37 ms renumbered; natural order: 0.38 ms either way).
    bool reorder = (d.flags & BDG_SW2D_REORDER) != 0;
    if (!reorder && !(d.flags & BDG_SW2D_KEEP_ORDER)) {
        std::atomic<long long> total{0};        // integers: the same sum whatever the number of workers
        blitzdg::detail::parallelChunks(K, [&](int kBegin, int kEnd) {
            long long mine = 0;
            for (int k = kBegin; k < kEnd; ++k)
                for (int f = 0; f < 3; ++f)
                    mine += std::llabs(static_cast<long long>(d.vmapP[(static_cast<size_t>(k) * 3 + f) * Nfp] / Np) - k);
            total += mine;
        });
        const double sum = static_cast<double>(total.load());
        reorder = sum / (3.0 * K) > 4.0 * std::sqrt(static_cast<double>(K));
    }
    if (reorder) s->permHost = bfsOrder(d.vmapP, K, Np, Nfp);
    s->affine = !(d.flags & BDG_SW2D_NODAL_GEOMETRY) && geometryIsAffine(d, Np, Nfp, K);
    // Non-affine tables: the matrix-core kernel with per-node geometry (every order). BDG_SW2D_NODAL_VECTOR=1 keeps the
    // round-1 vector kernel (one lane per element, N <= 6) for A/B measurements and cross-checks.
    s->nodalMfma = !s->affine && !(std::getenv("BDG_SW2D_NODAL_VECTOR") && kt->ldsDoubles != 0);

    s->use();
    hipCheck(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking), "hipStreamCreate");
    hipCheck(hipEventCreate(&s->ev0), "hipEventCreate");
    hipCheck(hipEventCreate(&s->ev1), "hipEventCreate");

    const size_t plane3 = static_cast<size_t>(s->nf) * s->planeSize();
    s->qA.alloc(plane3, s->bytes);
    s->qB.alloc(plane3, s->bytes);
    s->res.alloc(plane3, s->bytes);
    s->aux.alloc(plane3, s->bytes);
    // Per-node geometry planes and the LDS operator image belong to the nodal kernels; the affine
    // path keeps 13 values per element (below) plus the per-node Fscale plane, which only the
    // time-step reduction reads (so that dt stays bit-identical to the reference formula).
    const size_t fplane = static_cast<size_t>(NFN) * ld;
    if (!s->affine) {
        s->geo.alloc(4 * s->planeSize(), s->bytes);
        s->fgeo.alloc(3 * fplane, s->bytes);
        s->ops.alloc(kt->ldsDoubles, s->bytes);
    } else {
        s->fgeo.alloc(fplane, s->bytes);
    }
    s->fscaleNodal = s->affine ? s->fgeo.p : s->fgeo.p + 2 * fplane;
    s->vmapP.alloc(static_cast<size_t>(NFN) * ld, s->bytes);
    s->stage.alloc(static_cast<size_t>(std::max(Np, NFN)) * K, s->bytes);
    s->istage.alloc(static_cast<size_t>(NFN) * K, s->bytes);
    s->partials.alloc(2 * static_cast<size_t>((K + 255) / 256), s->bytes);
    s->red2.alloc(2, s->bytes);
    if (!s->permHost.empty()) {
        s->perm.alloc(K, s->bytes);
        hipCheck(hipMemcpy(s->perm.p, s->permHost.data(), sizeof(int) * K, hipMemcpyHostToDevice), "perm upload");
    }
    for (auto* b : {&s->qA, &s->qB, &s->res, &s->aux, &s->geo, &s->fgeo})
        if (b->p) hipCheck(hipMemsetAsync(b->p, 0, b->n * sizeof(double), s->stream), "hipMemset");
    hipCheck(hipMemsetAsync(s->vmapP.p, 0, s->vmapP.n * sizeof(int), s->stream), "hipMemset");
    s->qcur = s->qA.p;
    s->qalt = s->qB.p;

    s->hasFilter = d.Filter != nullptr;
    if (!s->affine && kt->ldsDoubles > 0) { // (orders 7, 8 have no vector kernel: matrix cores only)
        // ---- operator image for the nodal kernels: [Dr,Ds interleaved | Lift | Filter]
        std::vector<double> img(kt->ldsDoubles, 0.0);
        for (int i = 0; i < Np * Np; ++i) {
            img[2 * i] = d.Dr[i];
            img[2 * i + 1] = d.Ds[i];
        }
        std::copy(d.Lift, d.Lift + static_cast<size_t>(Np) * NFN, img.begin() + 2 * Np * Np);
        if (d.Filter) std::copy(d.Filter, d.Filter + static_cast<size_t>(Np) * Np, img.begin() + 2 * Np * Np + Np * NFN);
        hipCheck(hipMemcpyAsync(s->ops.p, img.data(), img.size() * sizeof(double), hipMemcpyHostToDevice, s->stream),
                 "ops upload");
        hipCheck(hipStreamSynchronize(s->stream), "ops sync");
    }

    // ---- geometry planes
    if (!s->affine) {
        const size_t pl = s->planeSize();
        s->uploadRows(d.rx, s->geo.p, Np);
        s->uploadRows(d.sx, s->geo.p + pl, Np);
        s->uploadRows(d.ry, s->geo.p + 2 * pl, Np);
        s->uploadRows(d.sy, s->geo.p + 3 * pl, Np);
        s->uploadRows(d.nx, s->fgeo.p, NFN);
        s->uploadRows(d.ny, s->fgeo.p + fplane, NFN);
    }
    s->uploadRows(d.Fscale, s->fscaleNodal, NFN);

    // ---- affine fast path: one metric value per element, one normal/scale per face
    // Measured on MI355X: the fully unrolled vector kernel wins up to N=4; from
    // N=5 on its basic block outgrows the register files and the matrix-core kernel is fastest
    // (N=5, 640 k elements: 0.454 ms unrolled, 0.405 ms matrix cores).
    // From N = 5 the default is the state-once matrix-core schedule (variant 7; measured against variant 6 at
    // 640 k / 500 k / 250 k / 250 k elements: N=5 0.36 vs 0.42 ms, N=6 0.33 vs 0.40, N=7 0.24 vs 0.28, N=8 0.30 vs 0.39).
    s->affineVariant = s->N <= 4 ? 0 : 7;
    if (const char* e = std::getenv("BDG_SW2D_AFFINE_VARIANT")) {
        const int v = std::atoi(e);
        if (v >= 0 && v <= 9) {
            s->affineVariant = v;
            s->variantForced = true;
        }
    }
    // the same operators as zero-padded 16x4 MFMA A tiles: lane l of tile (r, t) holds A[16r + (l&15)][4t + (l>>4)]
    auto mfmaImage = [&](const double* Dr, const double* Ds, const double* Lift) {
        const int MT = kt->mfmaMT, KV = kt->mfmaKV, KS = kt->mfmaKS;
        std::vector<double> img(static_cast<size_t>(kt->mfmaOpsDoubles), 0.0);
        auto fill = [&](size_t off, const double* A, int cols, int KT) {
            for (int r = 0; r < MT; ++r)
                for (int t = 0; t < KT; ++t)
                    for (int l = 0; l < 64; ++l) {
                        const int i = 16 * r + (l & 15), k = 4 * t + (l >> 4);
                        if (i < Np && k < cols) img[off + (static_cast<size_t>(r) * KT + t) * 64 + l] = A[i * cols + k];
                    }
        };
        fill(0, Dr, Np, KV);
        fill(static_cast<size_t>(MT) * KV * 64, Ds, Np, KV);
        fill(static_cast<size_t>(2) * MT * KV * 64, Lift, NFN, KS);
        return img;
    };
    // face-by-face variant: lift tile (r, f, tf), lane l = Lift[16r + (l&15)][f*Nfp + 4tf + (l>>4)]
    auto mfma2Image = [&](const double* Dr, const double* Ds, const double* Lift) {
        const int MT = kt->mfmaMT, KV = kt->mfmaKV, KF = kt->mfma2KF;
        std::vector<double> img(static_cast<size_t>(kt->mfma2OpsDoubles), 0.0);
        const std::vector<double> first = mfmaImage(Dr, Ds, Lift);
        std::copy(first.begin(), first.begin() + static_cast<size_t>(2) * MT * KV * 64, img.begin());
        const size_t off = static_cast<size_t>(2) * MT * KV * 64;
        for (int r = 0; r < MT; ++r)
            for (int f = 0; f < 3; ++f)
                for (int tf = 0; tf < KF; ++tf)
                    for (int l = 0; l < 64; ++l) {
                        const int i = 16 * r + (l & 15), n = 4 * tf + (l >> 4);
                        if (i < Np && n < Nfp)
                            img[off + ((static_cast<size_t>(r) * 3 + f) * KF + tf) * 64 + l] = Lift[i * NFN + f * Nfp + n];
                    }
        return img;
    };
    if (s->nodalMfma) { // per-node geometry on the matrix cores: the MfmaOps2 image, plain and pre-filtered
        const std::vector<double> img2 = mfma2Image(d.Dr, d.Ds, d.Lift);
        s->opsMfma2.alloc(img2.size(), s->bytes);
        hipCheck(hipMemcpy(s->opsMfma2.p, img2.data(), img2.size() * sizeof(double), hipMemcpyHostToDevice), "mfma2 ops upload");
        if (d.Filter) { // the filter follows the metric terms: plain operators, then the Filter's own tiles (r, t)
            std::vector<double> imgF = img2;
            const int MT = kt->mfmaMT, KV = kt->mfmaKV;
            imgF.resize(img2.size() + static_cast<size_t>(MT) * KV * 64, 0.0);
            for (int r = 0; r < MT; ++r)
                for (int t = 0; t < KV; ++t)
                    for (int l = 0; l < 64; ++l) {
                        const int i = 16 * r + (l & 15), k = 4 * t + (l >> 4);
                        if (i < Np && k < Np) imgF[img2.size() + (static_cast<size_t>(r) * KV + t) * 64 + l] = d.Filter[i * Np + k];
                    }
            s->opsMfma2NodalFilter.alloc(imgF.size(), s->bytes);
            hipCheck(hipMemcpy(s->opsMfma2NodalFilter.p, imgF.data(), imgF.size() * sizeof(double), hipMemcpyHostToDevice),
                     "nodal filter ops upload");
        }
    }
    if (s->affine) {
        s->ageo.alloc(13 * static_cast<size_t>(ld), s->bytes);
        hipCheck(hipMemsetAsync(s->ageo.p, 0, s->ageo.n * sizeof(double), s->stream), "hipMemset");
        s->uploadRows(d.rx, s->ageo.p, 1);           // row 0 of each (Np, K) table
        s->uploadRows(d.sx, s->ageo.p + ld, 1);
        s->uploadRows(d.ry, s->ageo.p + 2 * ld, 1);
        s->uploadRows(d.sy, s->ageo.p + 3 * ld, 1);
        for (int f = 0; f < 3; ++f) {                 // first node of each face
            const size_t row = static_cast<size_t>(f) * Nfp * K;
            s->uploadRows(d.nx + row, s->ageo.p + (4 + f) * ld, 1);
            s->uploadRows(d.ny + row, s->ageo.p + (7 + f) * ld, 1);
            s->uploadRows(d.Fscale + row, s->ageo.p + (10 + f) * ld, 1);
        }
        const std::vector<double> plain = affineOpsImage(d.Dr, d.Ds, d.Lift, Np, NFN);
        s->opsAffine.alloc(plain.size(), s->bytes);
        hipCheck(hipMemcpy(s->opsAffine.p, plain.data(), plain.size() * sizeof(double), hipMemcpyHostToDevice),
                 "affine ops upload");
        {
            const std::vector<double> img2 = mfma2Image(d.Dr, d.Ds, d.Lift);
            s->opsMfma2.alloc(img2.size(), s->bytes);
            hipCheck(hipMemcpy(s->opsMfma2.p, img2.data(), img2.size() * sizeof(double), hipMemcpyHostToDevice), "mfma2 ops upload");
        }
        {
            const std::vector<double> img = mfmaImage(d.Dr, d.Ds, d.Lift);
            s->opsMfma.alloc(img.size(), s->bytes);
            hipCheck(hipMemcpy(s->opsMfma.p, img.data(), img.size() * sizeof(double), hipMemcpyHostToDevice), "mfma ops upload");
        }
        if (d.Filter) {
            // Filter * (Dr, Ds, Lift): the filtered RHS of an affine element is linear in these.
            const std::vector<double> FDr = matmulHost(d.Filter, d.Dr, Np, Np), FDs = matmulHost(d.Filter, d.Ds, Np, Np),
                                      FL = matmulHost(d.Filter, d.Lift, Np, NFN);
            {
                const std::vector<double> img2 = mfma2Image(FDr.data(), FDs.data(), FL.data());
                s->opsMfma2Filtered.alloc(img2.size(), s->bytes);
                hipCheck(hipMemcpy(s->opsMfma2Filtered.p, img2.data(), img2.size() * sizeof(double), hipMemcpyHostToDevice),
                         "filtered mfma2 ops upload");
            }
            {
                const std::vector<double> img = mfmaImage(FDr.data(), FDs.data(), FL.data());
                s->opsMfmaFiltered.alloc(img.size(), s->bytes);
                hipCheck(hipMemcpy(s->opsMfmaFiltered.p, img.data(), img.size() * sizeof(double), hipMemcpyHostToDevice),
                         "filtered mfma ops upload");
            }
            const std::vector<double> filt = affineOpsImage(FDr.data(), FDs.data(), FL.data(), Np, NFN);
            s->opsAffineFiltered.alloc(filt.size(), s->bytes);
            hipCheck(hipMemcpy(s->opsAffineFiltered.p, filt.data(), filt.size() * sizeof(double),
                               hipMemcpyHostToDevice), "filtered affine ops upload");
        }
    }

    // ---- variant D: operator image with a third row per (m, i) (identity or Filter) + source tables
    s->hostDr.assign(d.Dr, d.Dr + static_cast<size_t>(Np) * Np);
    s->hostDs.assign(d.Ds, d.Ds + static_cast<size_t>(Np) * Np);
    s->hostLift.assign(d.Lift, d.Lift + static_cast<size_t>(Np) * NFN);
    if (d.Filter) s->hostFilter.assign(d.Filter, d.Filter + static_cast<size_t>(Np) * Np);
    if (s->variantD) {
        s->buildSourceOps();
        if (s->N > bdg_sw2d::kUnrolledSourcesMaxOrder) {
            s->buildMfma2SourceOps();
            s->mfmaSources = !std::getenv("BDG_SW2D_ROLLED_SOURCES");
        }
        s->vd.nf = s->nf;
        s->vd.sources = d.sources ? 1 : 0;
        s->vd.fconst = d.coriolis_const;
        s->vd.cd = d.drag;
        auto plane = [&](const double* host, DevBuf<double>& buf) -> const double* {
            if (!host) return nullptr;
            buf.alloc(s->planeSize(), s->bytes);
            hipCheck(hipMemsetAsync(buf.p, 0, buf.n * sizeof(double), s->stream), "hipMemset");
            s->uploadRows(host, buf.p, Np);
            return buf.p;
        };
        if (d.sources) {
            s->vd.zx = plane(d.zx, s->zxBuf);
            s->vd.zy = plane(d.zy, s->zyBuf);
            s->vd.fcor = plane(d.coriolis, s->fcorBuf);
        }
        // Measured at C3 (10^6 triangles, N=4): rolled kernel 0.87 ms (3 fields) / 1.16 ms (4 fields) per
        // stage; unrolled kernel + tracer launch: see DESIGN.md section 3.
        s->fastSources = s->N <= bdg_sw2d::kUnrolledSourcesMaxOrder && !std::getenv("BDG_SW2D_ROLLED_SOURCES");
        if (s->fastSources && d.Filter) {
            std::vector<double> ft(static_cast<size_t>(Np) * Np);
            for (int m = 0; m < Np; ++m)
                for (int i = 0; i < Np; ++i) ft[static_cast<size_t>(m) * Np + i] = d.Filter[i * Np + m];
            s->filterT.alloc(ft.size(), s->bytes);
            hipCheck(hipMemcpy(s->filterT.p, ft.data(), ft.size() * sizeof(double), hipMemcpyHostToDevice), "filter upload");
        }
    }

    // ---- gather offsets: reference numbering (n' + Np*k') -> n'*ld + slot(k'), as (NFN, K) rows;
    //      wall nodes are stored as -(offset+1).
    {
        std::vector<int> rows(nFaceNodes);
        const int* perm = s->permHost.empty() ? nullptr : s->permHost.data();
        for (int k = 0; k < K; ++k)
            for (int j = 0; j < NFN; ++j) {
                const int v = d.vmapP[static_cast<size_t>(k) * NFN + j];
                const int n2 = v % Np, k2 = v / Np;
                rows[static_cast<size_t>(j) * K + k] = static_cast<int>(n2 * ld + (perm ? perm[k2] : k2));
            }
        for (int i = 0; i < d.num_wall; ++i) {
            const int w = d.mapW[i], k = w / NFN, j = w % NFN;
            int& e = rows[static_cast<size_t>(j) * K + k];
            if (e >= 0) e = -(e + 1);
        }
        hipCheck(hipMemcpyAsync(s->istage.p, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice, s->stream),
                 "vmapP upload");
        const unsigned grid = static_cast<unsigned>((rows.size() + 255) / 256);
        hipLaunchKernelGGL((bdg_dev::scatter_rows_kernel<int>), dim3(grid), dim3(256), 0, s->stream, s->istage.p,
                           s->vmapP.p, NFN, K, ld, s->permDev());
        hipCheck(hipGetLastError(), "scatter_rows_kernel<int>");
        hipCheck(hipStreamSynchronize(s->stream), "vmapP sync");
    }
    s->istage.release();
    return s.release();
}

void requireSolver(const bdg_sw2d* s, const char* fn) {
    if (!s) throw arg_error(std::string(fn) + ": solver handle is NULL");
}

} // namespace

extern "C" {

int bdg_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int bdg_sw2d_create(const bdg_sw2d_desc* desc, bdg_sw2d** out) {
    return guard([&] {
        if (!desc || !out) throw arg_error("bdg_sw2d_create: NULL argument");
        *out = createSolver(*desc);
    });
}

int bdg_sw2d_create_from_nodes(const bdg_trinodes* nodes, double g, int device, int flags, bdg_sw2d** out) {
    return guard([&] {
        if (!nodes || !out) throw arg_error("bdg_sw2d_create_from_nodes: NULL argument");
        const blitzdg::TriangleNodesProvisioner& p = nodes->prov;
        bdg_sw2d_desc d{};
        d.order = p.get_NOrder();
        d.num_elements = p.get_NumElements();
        d.Dr = p.get_Dr().data(); d.Ds = p.get_Ds().data(); d.Lift = p.get_Lift().data();
        d.Filter = nodes->hasFilter ? p.get_Filter().data() : nullptr;
        d.rx = p.get_rx().data(); d.sx = p.get_sx().data(); d.ry = p.get_ry().data(); d.sy = p.get_sy().data();
        d.nx = p.get_nx().data(); d.ny = p.get_ny().data(); d.Fscale = p.get_Fscale().data();
        d.vmapM = p.get_vmapM().data(); d.vmapP = p.get_vmapP().data();
        const auto& bc = p.get_bcMap();
        const auto it = bc.find(blitzdg::BCTag::Wall);
        if (it != bc.end()) { d.mapW = it->second.data(); d.num_wall = static_cast<int>(it->second.size()); }
        d.g = g; d.device = device; d.flags = flags;
        *out = createSolver(d);
    });
}

void bdg_sw2d_destroy(bdg_sw2d* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    delete s;
}

int bdg_sw2d_set_state(bdg_sw2d* s, const double* h, const double* hu, const double* hv) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_set_state");
        if (s->nf != 3) throw arg_error("bdg_sw2d_set_state: this solver has 4 fields, use bdg_sw2d_set_state4");
        if (!h || !hu || !hv) throw arg_error("bdg_sw2d_set_state: NULL field");
        s->use();
        const size_t pl = s->planeSize();
        s->uploadRows(h, s->qcur, s->Np);
        s->uploadRows(hu, s->qcur + pl, s->Np);
        s->uploadRows(hv, s->qcur + 2 * pl, s->Np);
        hipCheck(hipMemsetAsync(s->res.p, 0, s->res.n * sizeof(double), s->stream), "hipMemset");
        s->stageCount = 0;
        s->lamStateFor = nullptr; // a speed accumulated for the previous contents of this buffer is void
        hipCheck(hipStreamSynchronize(s->stream), "set_state sync");
    });
}

int bdg_sw2d_get_state(bdg_sw2d* s, double* h, double* hu, double* hv) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_get_state");
        if (s->nf != 3) throw arg_error("bdg_sw2d_get_state: this solver has 4 fields, use bdg_sw2d_get_state4");
        if (!h || !hu || !hv) throw arg_error("bdg_sw2d_get_state: NULL field");
        s->use();
        const size_t pl = s->planeSize();
        s->downloadRows(s->qcur, h, s->Np);
        s->downloadRows(s->qcur + pl, hu, s->Np);
        s->downloadRows(s->qcur + 2 * pl, hv, s->Np);
        s->checkSyncError();
    });
}

int bdg_sw2d_output_fields(bdg_sw2d* s, const double* IM, double* eta, double* u, double* v) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_output_fields");
        s->use();
        if (IM) {
            if (!s->outM.p) s->outM.alloc(static_cast<size_t>(s->Np) * s->Np, s->bytes);
            hipCheck(hipMemcpyAsync(s->outM.p, IM, s->outM.n * sizeof(double), hipMemcpyHostToDevice, s->stream), "H2D copy");
        }
        double* targets[3] = {eta, u, v};
        for (int which = 0; which < 3; ++which) {
            if (!targets[which]) continue;
            // aux is scratch between steps (RHS output / RK2 intermediate)
            hipCheck(s->kt->output(s->qcur, s->hasH ? s->Hbuf.p : nullptr, IM ? s->outM.p : nullptr, s->aux.p, s->ld,
                                   s->numOwned, which, s->stream), "sw2d_output_kernel");
            s->downloadRows(s->aux.p, targets[which], s->Np);
        }
    });
}

int bdg_sw2d_output_tracer(bdg_sw2d* s, const double* IM, double* tracer) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_output_tracer");
        if (s->nf != 4) throw arg_error("bdg_sw2d_output_tracer: the solver has no tracer field");
        if (!tracer) throw arg_error("bdg_sw2d_output_tracer: NULL output");
        s->use();
        if (IM) {
            if (!s->outM.p) s->outM.alloc(static_cast<size_t>(s->Np) * s->Np, s->bytes);
            hipCheck(hipMemcpyAsync(s->outM.p, IM, s->outM.n * sizeof(double), hipMemcpyHostToDevice, s->stream), "H2D copy");
        }
        // which = 3: field 3 divided by h, i.e. the concentration N = hN / h the reference script writes out
        hipCheck(s->kt->output(s->qcur, nullptr, IM ? s->outM.p : nullptr, s->aux.p, s->ld, s->numOwned, 3, s->stream),
                 "sw2d_output_kernel");
        s->downloadRows(s->aux.p, tracer, s->Np);
    });
}

int bdg_sw2d_set_bathymetry(bdg_sw2d* s, const double* H) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_set_bathymetry");
        s->use();
        if (!H) { s->hasH = false; return; }
        if (!s->Hbuf.p) s->Hbuf.alloc(s->planeSize(), s->bytes);
        hipCheck(hipMemsetAsync(s->Hbuf.p, 0, s->Hbuf.n * sizeof(double), s->stream), "hipMemset");
        s->uploadRows(H, s->Hbuf.p, s->Np);
        s->hasH = true;
    });
}

int bdg_sw2d_enable_variant_b(bdg_sw2d* s, const bdg_sw2d_vb_desc* d) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_enable_variant_b");
        if (!d || !d->H || !d->Hx || !d->Hy) throw arg_error("bdg_sw2d_enable_variant_b: H, Hx and Hy are required");
        if (s->nf != 3 || s->variantD)
            throw arg_error("bdg_sw2d_enable_variant_b: the solver was created with tracer / variant-D sources");
        // partitioned solvers: allowed -- their steppers are the *_exchanged ones, which reduce the global speed over all
        // ranks before each evaluation (evaluateExchanged); the in-process group transport has no such reduction
        if (s->localGroup) throw arg_error("bdg_sw2d_enable_variant_b: not available with the in-process group transport");
        if (d->num_out < 0 || (d->num_out > 0 && !d->mapO)) throw arg_error("bdg_sw2d_enable_variant_b: bad open-boundary list");
        if (!(d->tide_period > 0.0) && d->num_out > 0) throw arg_error("bdg_sw2d_enable_variant_b: tide_period must be > 0");
        const size_t nFaceNodes = static_cast<size_t>(s->NFN) * s->K;
        for (int i = 0; i < d->num_out; ++i)
            if (d->mapO[i] < 0 || static_cast<size_t>(d->mapO[i]) >= nFaceNodes)
                throw arg_error("bdg_sw2d_enable_variant_b: open-boundary node index out of range");
        s->use();
        s->buildSourceOps();
        s->fastSources = s->N <= bdg_sw2d::kUnrolledSourcesMaxOrder && !std::getenv("BDG_SW2D_ROLLED_SOURCES");
        if (s->N > bdg_sw2d::kUnrolledSourcesMaxOrder && !std::getenv("BDG_SW2D_ROLLED_SOURCES")) {
            s->buildMfma2SourceOps();
            s->mfmaSources = true;
        }
        if (s->fastSources && !s->hostFilter.empty() && !s->filterT.p) {
            const int Np = s->Np;
            std::vector<double> ft(static_cast<size_t>(Np) * Np);
            for (int m = 0; m < Np; ++m)
                for (int i = 0; i < Np; ++i) ft[static_cast<size_t>(m) * Np + i] = s->hostFilter[static_cast<size_t>(i) * Np + m];
            s->filterT.alloc(ft.size(), s->bytes);
            hipCheck(hipMemcpy(s->filterT.p, ft.data(), ft.size() * sizeof(double), hipMemcpyHostToDevice), "filter upload");
        }
        if (!s->Hbuf.p) s->Hbuf.alloc(s->planeSize(), s->bytes);
        hipCheck(hipMemsetAsync(s->Hbuf.p, 0, s->Hbuf.n * sizeof(double), s->stream), "hipMemset");
        // padding lanes are never computed, but give them a positive depth anyway
        s->uploadRows(d->H, s->Hbuf.p, s->Np);
        s->hasH = true;
        s->vb.H = s->Hbuf.p;
        s->vb.Hx = s->uploadPlane(d->Hx, s->HxBuf);
        s->vb.Hy = s->uploadPlane(d->Hy, s->HyBuf);
        s->vb.sponge = s->uploadPlane(d->sponge, s->spongeBuf);
        // open-boundary face nodes as one bit mask per element, in device slots
        std::vector<int> mask(static_cast<size_t>(s->ld), 0);
        for (int i = 0; i < d->num_out; ++i) {
            const int k = d->mapO[i] / s->NFN, j = d->mapO[i] % s->NFN;
            mask[s->permHost.empty() ? k : s->permHost[k]] |= 1 << j;
        }
        if (!s->obcBuf.p) s->obcBuf.alloc(static_cast<size_t>(s->ld), s->bytes);
        hipCheck(hipMemcpy(s->obcBuf.p, mask.data(), mask.size() * sizeof(int), hipMemcpyHostToDevice), "open-boundary upload");
        s->vb.obc = s->obcBuf.p;
        if (!s->lamBuf.p) s->lamBuf.alloc(1, s->bytes);
        if (!s->lamPair.p) s->lamPair.alloc(2, s->bytes);
        s->lamStateFor = nullptr;
        if (!s->vbPartials.p) s->vbPartials.alloc(static_cast<size_t>((s->K + 255) / 256), s->bytes);
        s->vb.lam = s->lamBuf.p;
        s->vb.fcor = d->coriolis;
        s->vb.cd = d->drag;
        s->tideAmp = d->tide_amplitude;
        s->tidePeriod = d->tide_period > 0.0 ? d->tide_period : 1.0;
        s->tideRamp = d->tide_ramp;
        s->variantB = true;
    });
}

int bdg_sw2d_set_time(bdg_sw2d* s, double t) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_set_time");
        s->timeNow = t;
    });
}

int bdg_sw2d_get_time(const bdg_sw2d* s, double* t) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_get_time");
        if (!t) throw arg_error("bdg_sw2d_get_time: NULL argument");
        *t = s->timeNow;
    });
}

int bdg_sw2d_global_speed(bdg_sw2d* s, double* lam) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_global_speed");
        if (!s->variantB || !lam) throw arg_error("bdg_sw2d_global_speed: variant B is not enabled");
        s->use();
        hipCheck(hipMemcpyAsync(lam, s->vb.lam ? s->vb.lam : s->lamBuf.p, sizeof(double), hipMemcpyDeviceToHost, s->stream),
                 "D2H copy");
        hipCheck(hipStreamSynchronize(s->stream), "sync");
    });
}

int bdg_sw2d_rhs(bdg_sw2d* s, const double* h, const double* hu, const double* hv, double* r1, double* r2,
                 double* r3, int filter) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_rhs");
        if (s->nf != 3) throw arg_error("bdg_sw2d_rhs: this solver has 4 fields, use bdg_sw2d_rhs4");
        if (!h || !hu || !hv || !r1 || !r2 || !r3) throw arg_error("bdg_sw2d_rhs: NULL field");
        s->use();
        const size_t pl = s->planeSize();
        // the inactive state buffer is scratch between steps
        s->uploadRows(h, s->qalt, s->Np);
        s->uploadRows(hu, s->qalt + pl, s->Np);
        s->uploadRows(hv, s->qalt + 2 * pl, s->Np);
        s->launchRhs(s->qalt, s->aux.p, filter != 0);
        s->downloadRows(s->aux.p, r1, s->Np);
        s->downloadRows(s->aux.p + pl, r2, s->Np);
        s->downloadRows(s->aux.p + 2 * pl, r3, s->Np);
    });
}

int bdg_sw2d_num_fields(const bdg_sw2d* s) { return s ? s->nf : -1; }

int bdg_sw2d_set_state4(bdg_sw2d* s, const double* h, const double* hu, const double* hv, const double* hN) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_set_state4");
        if (s->nf != 4) throw arg_error("bdg_sw2d_set_state4: the solver was created with 3 fields");
        const double* f[4] = {h, hu, hv, hN};
        s->use();
        for (int c = 0; c < 4; ++c) {
            if (!f[c]) throw arg_error("bdg_sw2d_set_state4: NULL field");
            s->uploadRows(f[c], s->qcur + c * s->planeSize(), s->Np);
        }
        hipCheck(hipMemsetAsync(s->res.p, 0, s->res.n * sizeof(double), s->stream), "hipMemset");
        s->stageCount = 0;
        hipCheck(hipStreamSynchronize(s->stream), "set_state sync");
    });
}

int bdg_sw2d_get_state4(bdg_sw2d* s, double* h, double* hu, double* hv, double* hN) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_get_state4");
        if (s->nf != 4) throw arg_error("bdg_sw2d_get_state4: the solver was created with 3 fields");
        double* f[4] = {h, hu, hv, hN};
        s->use();
        for (int c = 0; c < 4; ++c) {
            if (!f[c]) throw arg_error("bdg_sw2d_get_state4: NULL field");
            s->downloadRows(s->qcur + c * s->planeSize(), f[c], s->Np);
        }
    });
}

int bdg_sw2d_rhs4(bdg_sw2d* s, const double* h, const double* hu, const double* hv, const double* hN, double* r1,
                  double* r2, double* r3, double* r4, int filter) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_rhs4");
        if (s->nf != 4) throw arg_error("bdg_sw2d_rhs4: the solver was created with 3 fields");
        const double* in[4] = {h, hu, hv, hN};
        double* out[4] = {r1, r2, r3, r4};
        s->use();
        for (int c = 0; c < 4; ++c) {
            if (!in[c] || !out[c]) throw arg_error("bdg_sw2d_rhs4: NULL field");
            s->uploadRows(in[c], s->qalt + c * s->planeSize(), s->Np);
        }
        s->launchRhs(s->qalt, s->aux.p, filter != 0);
        for (int c = 0; c < 4; ++c) s->downloadRows(s->aux.p + c * s->planeSize(), out[c], s->Np);
    });
}

int bdg_sw2d_lserk4_stages(bdg_sw2d* s, double dt, int num_stages) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_lserk4_stages");
        if (num_stages < 0) throw arg_error("bdg_sw2d_lserk4_stages: num_stages < 0");
        s->use();
        s->dtStage = dt;
        for (int i = 0; i < num_stages; ++i) s->launchLserkStage();
    });
}

int bdg_sw2d_step_lserk4(bdg_sw2d* s, double dt, int num_steps) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_step_lserk4");
        if (num_steps < 0) throw arg_error("bdg_sw2d_step_lserk4: num_steps < 0");
        if (s->stageCount % blitzdg::LSERK4::numStages != 0)
            throw arg_error("bdg_sw2d_step_lserk4: a previous step was left part-way through its stages");
        s->use();
        s->dtStage = dt;
        for (int i = 0; i < num_steps * blitzdg::LSERK4::numStages; ++i) s->launchLserkStage();
    });
}

int bdg_sw2d_step_rk2(bdg_sw2d* s, double dt, int num_steps, int filter) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_step_rk2");
        if (num_steps < 0) throw arg_error("bdg_sw2d_step_rk2: num_steps < 0");
        s->use();
        for (int i = 0; i < num_steps; ++i) s->launchRk2Step(dt, filter != 0);
    });
}

int bdg_sw2d_step_ssprk2(bdg_sw2d* s, double dt, int num_steps, int filter, double sponge_coeff) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_step_ssprk2");
        if (num_steps < 0) throw arg_error("bdg_sw2d_step_ssprk2: num_steps < 0");
        s->use();
        for (int i = 0; i < num_steps; ++i) s->launchSspRk2Step(dt, filter != 0, sponge_coeff);
    });
}

int bdg_sw2d_step_rk2_exchanged(bdg_sw2d* s, double dt, int num_steps, int filter) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_step_rk2_exchanged");
        if (num_steps < 0) throw arg_error("bdg_sw2d_step_rk2_exchanged: num_steps < 0");
        s->use();
        for (int i = 0; i < num_steps; ++i) s->launchRk2StepExchanged(dt, filter != 0);
    });
}

int bdg_sw2d_step_ssprk2_exchanged(bdg_sw2d* s, double dt, int num_steps, int filter, double sponge_coeff) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_step_ssprk2_exchanged");
        if (num_steps < 0) throw arg_error("bdg_sw2d_step_ssprk2_exchanged: num_steps < 0");
        s->use();
        for (int i = 0; i < num_steps; ++i) s->launchSspRk2StepExchanged(dt, filter != 0, sponge_coeff);
    });
}

int bdg_sw2d_compute_dt(bdg_sw2d* s, double cfl, double* dt, double* eta_max) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_compute_dt");
        s->use();
        double r[2];
        s->reduceDt(r);
        if (eta_max) *eta_max = r[1];
        if (dt) *dt = cfl / ((s->N + 1) * (s->N + 1) * 0.5 * r[0]);
        if (std::isnan(r[0]) || std::isnan(r[1]) || std::fabs(r[1]) > 1e8)
            throw unstable_error("A numerical instability has occurred!");
    });
}

int bdg_sw2d_run_adaptive(bdg_sw2d* s, double cfl, double final_time, int max_steps, int filter, double* t_inout,
                          double* dt_inout, int* steps_done) {
    int done = 0;
    const int rc = guard([&] {
        requireSolver(s, "bdg_sw2d_run_adaptive");
        if (!t_inout || !dt_inout) throw arg_error("bdg_sw2d_run_adaptive: NULL argument");
        s->use();
        double t = *t_inout, dt = *dt_inout;
        while (t < final_time && (max_steps <= 0 || done < max_steps)) {
            s->launchRk2Step(dt, filter != 0);
            double r[2];
            s->reduceDt(r);
            if (std::isnan(r[0]) || std::isnan(r[1]) || std::fabs(r[1]) > 1e8)
                throw unstable_error("A numerical instability has occurred!");
            dt = cfl / ((s->N + 1) * (s->N + 1) * 0.5 * r[0]);
            t += dt;
            ++done;
            *t_inout = t;
            *dt_inout = dt;
        }
    });
    if (steps_done) *steps_done = done;
    return rc;
}

int bdg_sw2d_synchronize(bdg_sw2d* s) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_synchronize");
        s->use();
        hipCheck(hipStreamSynchronize(s->stream), "hipStreamSynchronize");
        s->checkSyncError();
    });
}

int bdg_sw2d_time_lserk4_stages(bdg_sw2d* s, double dt, int num_stages, float* ms_per_launch) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_time_lserk4_stages");
        if (num_stages < 1 || !ms_per_launch) throw arg_error("bdg_sw2d_time_lserk4_stages: bad argument");
        s->use();
        s->dtStage = dt;
        hipCheck(hipEventRecord(s->ev0, s->stream), "hipEventRecord");
        for (int i = 0; i < num_stages; ++i) s->launchLserkStage();
        hipCheck(hipEventRecord(s->ev1, s->stream), "hipEventRecord");
        hipCheck(hipEventSynchronize(s->ev1), "hipEventSynchronize");
        float ms = 0.f;
        hipCheck(hipEventElapsedTime(&ms, s->ev0, s->ev1), "hipEventElapsedTime");
        *ms_per_launch = ms / num_stages;
    });
}

int bdg_sw2d_set_partition(bdg_sw2d* s, int num_interior, int num_owned, const int* send_elements, int num_send) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_set_partition");
        if (!s->permHost.empty())
            throw arg_error("bdg_sw2d_set_partition: the solver renumbered its elements; create it with BDG_SW2D_KEEP_ORDER");
        if (num_interior < 0 || num_interior > num_owned || num_owned > s->K || num_send < 0 ||
            (num_send > 0 && !send_elements))
            throw arg_error("bdg_sw2d_set_partition: need 0 <= num_interior <= num_owned <= K");
        for (int i = 0; i < num_send; ++i)
            if (send_elements[i] < 0 || send_elements[i] >= num_owned)
                throw arg_error("bdg_sw2d_set_partition: send element is not an owned element");
        s->use();
        s->sendSlots.release();
        if (num_send > 0) {
            s->sendSlots.alloc(num_send, s->bytes);
            hipCheck(hipMemcpy(s->sendSlots.p, send_elements, sizeof(int) * num_send, hipMemcpyHostToDevice),
                     "send list upload");
        }
        s->numInterior = num_interior;
        s->numOwned = num_owned;
        s->numSend = num_send;
        // the first interior element with a partition-boundary neighbour: interior tiles from there on read what the boundary
        // launch of the previous stage writes (and overwrite what it reads); tiles before it depend on interior elements only
        {
            const int NFN = s->NFN;
            std::vector<int> vm(static_cast<size_t>(NFN) * s->ld);
            hipCheck(hipMemcpy(vm.data(), s->vmapP.p, vm.size() * sizeof(int), hipMemcpyDeviceToHost), "gather table download");
            int ring = num_interior;
            for (int jf = 0; jf < NFN && ring > 0; ++jf) {
                const int* row = vm.data() + static_cast<size_t>(jf) * s->ld;
                for (int k = 0; k < ring; ++k) {
                    const int id = row[k];
                    const long long slot = static_cast<long long>(id < 0 ? -(id + 1) : id) % s->ld;
                    if (slot >= num_interior && slot < num_owned) { ring = k; break; }
                }
            }
            s->ringBegin = ring;
        }
        // for the boundary kernel with the halo staging folded in: the (up to three) send records of each
        // partition-boundary element
        const int nB = num_owned - num_interior;
        std::vector<int> of(static_cast<size_t>(3) * std::max(nB, 1), -1);
        bool ok = true;
        for (int r = 0; r < num_send && ok; ++r) {
            const int b = send_elements[r] - num_interior;
            if (b < 0) { ok = false; break; }
            int t = 0;
            while (t < 3 && of[3 * b + t] >= 0) ++t;
            if (t == 3) { ok = false; break; }
            of[3 * b + t] = r;
        }
        s->haloSendOf.release();
        s->haloFusable = ok && nB > 0;
        if (s->haloFusable) {
            s->haloSendOf.alloc(of.size(), s->bytes);
            hipCheck(hipMemcpy(s->haloSendOf.p, of.data(), of.size() * sizeof(int), hipMemcpyHostToDevice), "send table upload");
        }
    });
}

int bdg_sw2d_halo_doubles_per_element(const bdg_sw2d* s) { return s ? s->nf * s->Np : -1; }

int bdg_sw2d_halo_pack(bdg_sw2d* s, void* send_buffer) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_halo_pack");
        if (s->numSend == 0) return;
        if (!send_buffer) throw arg_error("bdg_sw2d_halo_pack: buffer is NULL");
        s->use();
        const int rows = s->nf * s->Np;
        const long long n = static_cast<long long>(s->numSend) * rows;
        hipLaunchKernelGGL(bdg_dev::halo_pack_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
                           s->stream, s->qcur, static_cast<double*>(send_buffer), s->sendSlots.p, s->numSend, rows,
                           s->ld);
        hipCheck(hipGetLastError(), "halo_pack_kernel");
    });
}

int bdg_sw2d_halo_unpack(bdg_sw2d* s, const void* recv_buffer) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_halo_unpack");
        const int ghosts = s->K - s->numOwned;
        if (ghosts == 0) return;
        if (!recv_buffer) throw arg_error("bdg_sw2d_halo_unpack: buffer is NULL");
        s->use();
        const int rows = s->nf * s->Np;
        const long long n = static_cast<long long>(ghosts) * rows;
        hipLaunchKernelGGL(bdg_dev::halo_unpack_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
                           s->stream, s->qcur, static_cast<const double*>(recv_buffer), s->numOwned, ghosts, rows,
                           s->ld);
        hipCheck(hipGetLastError(), "halo_unpack_kernel");
    });
}

int bdg_sw2d_lserk4_stage_part(bdg_sw2d* s, double dt, int part) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_lserk4_stage_part");
        if (part < 0 || part > 2) throw arg_error("bdg_sw2d_lserk4_stage_part: part must be 0, 1 or 2");
        s->use();
        s->dtStage = dt;
        s->launchLserkStage(part);
    });
}

int bdg_sw2d_rhs_resident(bdg_sw2d* s, double* r1, double* r2, double* r3) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_rhs_resident");
        if (s->nf != 3) throw arg_error("bdg_sw2d_rhs_resident: three-field solvers only");
        if (!r1 || !r2 || !r3) throw arg_error("bdg_sw2d_rhs_resident: NULL output");
        s->use();
        const size_t pl = s->planeSize();
        s->launchRhs(s->qcur, s->aux.p, false);
        s->downloadRows(s->aux.p, r1, s->Np);
        s->downloadRows(s->aux.p + pl, r2, s->Np);
        s->downloadRows(s->aux.p + 2 * pl, r3, s->Np);
    });
}

int bdg_comm_unique_id(void* id_out, int capacity) {
    return guard([&] {
        if (!id_out || capacity < static_cast<int>(sizeof(ncclUniqueId)))
            throw arg_error("bdg_comm_unique_id: need a buffer of at least 128 bytes");
        ncclUniqueId id;
        ncclCheck(rccl().GetUniqueId(&id), "ncclGetUniqueId");
        std::memcpy(id_out, &id, sizeof(id));
    });
}

int bdg_sw2d_comm_init(bdg_sw2d* s, int rank, int world, const void* unique_id, const int* peer_ranks,
                       const int* send_start, const int* send_count, const int* recv_start, const int* recv_count,
                       int num_peers) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_comm_init");
        if (!unique_id || world < 1 || rank < 0 || rank >= world || num_peers < 0 ||
            (num_peers > 0 && (!peer_ranks || !send_start || !send_count || !recv_start || !recv_count)))
            throw arg_error("bdg_sw2d_comm_init: bad argument");
        if (s->comm) throw arg_error("bdg_sw2d_comm_init: communicator already initialised");
        const int ghosts = s->K - s->numOwned;
        std::vector<bdg_sw2d::Peer> peers;
        for (int i = 0; i < num_peers; ++i) {
            const bdg_sw2d::Peer p{peer_ranks[i], send_start[i], send_count[i], recv_start[i], recv_count[i]};
            if (p.rank < 0 || p.rank >= world || p.sendStart < 0 || p.sendCount < 0 ||
                p.sendStart + p.sendCount > s->numSend || p.recvStart < 0 || p.recvCount < 0 ||
                p.recvStart + p.recvCount > ghosts)
                throw arg_error("bdg_sw2d_comm_init: peer ranges do not fit the partition set with bdg_sw2d_set_partition");
            peers.push_back(p);
        }
        s->use();
        ncclUniqueId id;
        std::memcpy(&id, unique_id, sizeof(id));
        ncclCheck(rccl().CommInitRank(&s->comm, world, id, rank), "ncclCommInitRank");
        s->commRank = rank;
        s->commWorld = world;
        s->peers = peers;
        // (default priority: at the highest priority every stage of every order took about twice as long -- 8-way N=4 0.0958 against 0.0483 ms,
        // N=8 0.113 against 0.052, 2-way 0.293 against 0.205: profiles/r04_rehearsal_experiments.txt, call 28)
        hipCheck(hipStreamCreateWithFlags(&s->commStream, hipStreamNonBlocking), "hipStreamCreate");
        // These four events only order kernels of the two streams of THIS device against each other (a kernel's own
        // end-of-kernel release is device-wide, and data from another GPU is made visible by the RCCL kernel that received
        // it, on the stream that then runs the boundary kernel): the system-scope fence of a default event record is not
        // needed and costs 2-3 us per stage (8-way rehearsal, N=4: 0.056 ms per stage without, 0.058-0.059 with).
        // tests/test_dist_gpu.py::test_loopback_rccl_result_is_independent_of_event_flags_and_halo_staging compares both
        // settings bit for bit through real RCCL; BDG_SW2D_EVENT_FENCE=1 (or BDG_SW2D_EVENT_NOFENCE=0) restores the fence.
        const char* fenceOn = std::getenv("BDG_SW2D_EVENT_FENCE");
        const char* fenceOff = std::getenv("BDG_SW2D_EVENT_NOFENCE");
        const bool fence = (fenceOn && fenceOn[0] == '1') || (fenceOff && fenceOff[0] == '0');
        const unsigned evFlags = hipEventDisableTiming | (fence ? 0u : hipEventDisableSystemFence);
        for (hipEvent_t* e : {&s->evA[0], &s->evA[1], &s->evB[0], &s->evB[1]})
            hipCheck(hipEventCreateWithFlags(e, evFlags), "hipEventCreate");
        const size_t rows = static_cast<size_t>(s->nf) * s->Np;
        s->sendBuf.alloc(std::max<size_t>(1, static_cast<size_t>(s->numSend) * rows), s->bytes);
        s->recvBuf.alloc(std::max<size_t>(1, static_cast<size_t>(ghosts) * rows), s->bytes);
        hipCheck(hipMemset(s->sendBuf.p, 0, s->sendBuf.n * sizeof(double)), "hipMemset");
        hipCheck(hipMemset(s->recvBuf.p, 0, s->recvBuf.n * sizeof(double)), "hipMemset");
        s->scalarBuf.alloc(2, s->bytes);
        s->syncBuf.alloc(8, s->bytes);
        hipCheck(hipMemset(s->syncBuf.p, 0, 8 * sizeof(unsigned long long)), "hipMemset");
        s->expectRing = s->expectStrip = 0;
        // The first send / receive between two ranks sets their connection up (host side, inside ncclGroupEnd, and only as fast as the
        // slower of the two gets there). Do that HERE, where every rank is anyway and nothing is in flight: one double each way with
        // every neighbour, in the buffers and with the pairing of the stage exchange. Otherwise it would happen in the first exchanged
        // stage, behind interior launches whose ring tiles wait -- with a bound -- for the launches queued behind that exchange.
        if (!s->peers.empty() && !std::getenv("BDG_SW2D_NO_COMM_WARMUP")) {
            RcclApi& nc = rccl();
            ncclCheck(nc.GroupStart(), "ncclGroupStart");
            for (const bdg_sw2d::Peer& pr : s->peers) {
                if (pr.recvCount > 0)
                    ncclCheck(nc.Recv(s->recvBuf.p + static_cast<size_t>(pr.recvStart) * rows, 1, ncclDouble, pr.rank, s->comm, s->commStream), "ncclRecv");
                if (pr.sendCount > 0)
                    ncclCheck(nc.Send(s->sendBuf.p + static_cast<size_t>(pr.sendStart) * rows, 1, ncclDouble, pr.rank, s->comm, s->commStream), "ncclSend");
            }
            ncclCheck(nc.GroupEnd(), "ncclGroupEnd");
            hipCheck(hipStreamSynchronize(s->commStream), "hipStreamSynchronize");
        }
    });
}

// ---- in-process group: every part of the split lives in this process (one per GPU of the node, or
// several on one GPU); the exchange is a device-to-device copy from the neighbour's send buffer. The
// schedule is the two-chain pipeline of launchLserkStagesExchanged with the RCCL group replaced by
//   B_r: wait packed(p) -> copy p.send[range towards r] -> r.recv[range from p]   for every neighbour p
// and the send buffer of a part protected by copied(r) of every neighbour before its next pack.
int bdg_sw2d_local_peers(bdg_sw2d* s, int rank, const int* peer_ranks, const int* send_start, const int* send_count,
                         const int* recv_start, const int* recv_count, int num_peers) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_local_peers");
        if (rank < 0 || num_peers < 0 ||
            (num_peers > 0 && (!peer_ranks || !send_start || !send_count || !recv_start || !recv_count)))
            throw arg_error("bdg_sw2d_local_peers: bad argument");
        if (s->comm || s->localGroup) throw arg_error("bdg_sw2d_local_peers: a transport is already initialised");
        const int ghosts = s->K - s->numOwned;
        std::vector<bdg_sw2d::Peer> peers;
        for (int i = 0; i < num_peers; ++i) {
            const bdg_sw2d::Peer p{peer_ranks[i], send_start[i], send_count[i], recv_start[i], recv_count[i]};
            if (p.rank < 0 || p.rank == rank || p.sendStart < 0 || p.sendCount < 0 || p.sendStart + p.sendCount > s->numSend ||
                p.recvStart < 0 || p.recvCount < 0 || p.recvStart + p.recvCount > ghosts)
                throw arg_error("bdg_sw2d_local_peers: peer ranges do not fit the partition set with bdg_sw2d_set_partition");
            peers.push_back(p);
        }
        s->use();
        s->commRank = rank;
        s->peers = peers;
        hipCheck(hipStreamCreateWithFlags(&s->commStream, hipStreamNonBlocking), "hipStreamCreate");
        for (hipEvent_t* e : {&s->evA[0], &s->evA[1], &s->evB[0], &s->evB[1], &s->evPacked[0], &s->evPacked[1],
                              &s->evCopied[0], &s->evCopied[1]})
            hipCheck(hipEventCreateWithFlags(e, hipEventDisableTiming), "hipEventCreate");
        const size_t rows = static_cast<size_t>(s->nf) * s->Np;
        s->sendBuf.alloc(std::max<size_t>(1, static_cast<size_t>(s->numSend) * rows), s->bytes);
        s->recvBuf.alloc(std::max<size_t>(1, static_cast<size_t>(ghosts) * rows), s->bytes);
        s->localGroup = true;
    });
}

int bdg_sw2d_group_lserk4_stages(bdg_sw2d** parts, int num_parts, double dt, int num_stages) {
    return guard([&] {
        if (!parts || num_parts < 1 || num_stages < 0) throw arg_error("bdg_sw2d_group_lserk4_stages: bad argument");
        for (int r = 0; r < num_parts; ++r) {
            if (!parts[r] || !parts[r]->localGroup || parts[r]->commRank != r)
                throw arg_error("bdg_sw2d_group_lserk4_stages: parts[r] must be the part given rank r in bdg_sw2d_local_peers");
            for (const bdg_sw2d::Peer& pr : parts[r]->peers) {
                if (pr.rank >= num_parts) throw arg_error("bdg_sw2d_group_lserk4_stages: peer rank outside the group");
                bool matched = false;
                for (const bdg_sw2d::Peer& back : parts[pr.rank]->peers)
                    matched = matched || (back.rank == r && back.sendCount == pr.recvCount && back.recvCount == pr.sendCount);
                if (!matched) throw arg_error("bdg_sw2d_group_lserk4_stages: neighbour tables of two parts do not match");
            }
            if (parts[r]->nf != parts[0]->nf || parts[r]->Np != parts[0]->Np)
                throw arg_error("bdg_sw2d_group_lserk4_stages: parts differ in order / field count");
        }
        if (num_stages == 0) return;
        for (int r = 0; r < num_parts; ++r)          // direct access between the GPUs that exchange ghosts
            for (const bdg_sw2d::Peer& pr : parts[r]->peers)
                if (parts[pr.rank]->device != parts[r]->device) {
                    parts[r]->use();
                    const hipError_t e = hipDeviceEnablePeerAccess(parts[pr.rank]->device, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) hipCheck(e, "hipDeviceEnablePeerAccess");
                    (void)hipGetLastError();
                }
        const size_t rows = static_cast<size_t>(parts[0]->nf) * parts[0]->Np;
        for (int r = 0; r < num_parts; ++r) {
            bdg_sw2d* s = parts[r];
            s->use();
            s->dtStage = dt;
            hipCheck(hipEventRecord(s->evA[1], s->stream), "hipEventRecord");
            hipCheck(hipStreamWaitEvent(s->commStream, s->evA[1], 0), "hipStreamWaitEvent");
        }
        // pack / unpack folded into the boundary kernel where every part can do it (see launchLserkStagesExchanged)
        bool fold = true;
        for (int r = 0; r < num_parts; ++r) fold = fold && parts[r]->halosFold();
        for (int i = 0; i < num_stages; ++i) {
            const int cur = i & 1, prev = cur ^ 1;
            const bool first = i == 0;
            for (int r = 0; r < num_parts; ++r) {           // chain A + pack
                bdg_sw2d* s = parts[r];
                s->use();
                if (!first) hipCheck(hipStreamWaitEvent(s->stream, s->evB[prev], 0), "hipStreamWaitEvent");
                s->launchLserkStage(0);
                hipCheck(hipEventRecord(s->evA[cur], s->stream), "hipEventRecord");
                if (!fold || first) {
                    if (!first)                               // neighbours are done reading our send buffer
                        for (const bdg_sw2d::Peer& pr : s->peers)
                            hipCheck(hipStreamWaitEvent(s->commStream, parts[pr.rank]->evCopied[prev], 0), "hipStreamWaitEvent");
                    s->launchPack(s->sendBuf.p, s->commStream);
                }                                             // (folded: boundary(i-1) wrote the records, same stream)
                hipCheck(hipEventRecord(s->evPacked[cur], s->commStream), "hipEventRecord");
            }
            for (int r = 0; r < num_parts; ++r) {           // pull the ghosts
                bdg_sw2d* s = parts[r];
                s->use();
                for (const bdg_sw2d::Peer& pr : s->peers) {
                    if (pr.recvCount == 0) continue;
                    bdg_sw2d* src = parts[pr.rank];
                    const bdg_sw2d::Peer* back = nullptr;
                    for (const bdg_sw2d::Peer& b : src->peers)
                        if (b.rank == r) back = &b;
                    hipCheck(hipStreamWaitEvent(s->commStream, src->evPacked[cur], 0), "hipStreamWaitEvent");
                    double* dst = s->recvBuf.p + static_cast<size_t>(pr.recvStart) * rows;
                    const double* from = src->sendBuf.p + static_cast<size_t>(back->sendStart) * rows;
                    const size_t bytes = static_cast<size_t>(pr.recvCount) * rows * sizeof(double);
                    if (src->device == s->device)
                        hipCheck(hipMemcpyAsync(dst, from, bytes, hipMemcpyDeviceToDevice, s->commStream), "ghost copy");
                    else  // another GPU of the node: the copy engine pulls over xGMI
                        hipCheck(hipMemcpyPeerAsync(dst, s->device, from, src->device, bytes, s->commStream), "ghost peer copy");
                }
                hipCheck(hipEventRecord(s->evCopied[cur], s->commStream), "hipEventRecord");
            }
            for (int r = 0; r < num_parts; ++r) {           // chain B: unpack, boundary elements
                bdg_sw2d* s = parts[r];
                s->use();
                if (!fold) s->launchUnpack(s->recvBuf.p, s->commStream);
                if (!first) hipCheck(hipStreamWaitEvent(s->commStream, s->evA[prev], 0), "hipStreamWaitEvent");
                if (fold) {
                    // the kernel overwrites the send records: every neighbour must have pulled this stage's first
                    for (const bdg_sw2d::Peer& pr : s->peers)
                        hipCheck(hipStreamWaitEvent(s->commStream, parts[pr.rank]->evCopied[cur], 0), "hipStreamWaitEvent");
                    s->launchBoundaryStageFolded(s->commStream, s->recvBuf.p, s->sendBuf.p);
                } else {
                    s->launchLserkStage(1, s->commStream);
                }
                hipCheck(hipEventRecord(s->evB[cur], s->commStream), "hipEventRecord");
            }
        }
        for (int r = 0; r < num_parts; ++r) {
            bdg_sw2d* s = parts[r];
            s->use();
            hipCheck(hipStreamWaitEvent(s->stream, s->evB[(num_stages - 1) & 1], 0), "hipStreamWaitEvent");
            // the neighbours' last copies read this part's send buffer: order them before anything later on A
            for (const bdg_sw2d::Peer& pr : s->peers)
                hipCheck(hipStreamWaitEvent(s->stream, parts[pr.rank]->evCopied[(num_stages - 1) & 1], 0), "hipStreamWaitEvent");
        }
    });
}

int bdg_sw2d_lserk4_stages_exchanged(bdg_sw2d* s, double dt, int num_stages) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_lserk4_stages_exchanged");
        if (num_stages < 0) throw arg_error("bdg_sw2d_lserk4_stages_exchanged: num_stages < 0");
        s->use();
        s->launchLserkStagesExchanged(dt, num_stages);
    });
}

int bdg_sw2d_compute_dt_global(bdg_sw2d* s, double cfl, double* dt, double* eta_max) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_compute_dt_global");
        s->use();
        double r[2];
        s->reduceDt(r);
        // NaN does not survive a max-reduction reliably: send it as +inf
        const double big = std::numeric_limits<double>::infinity();
        const double f = s->allReduceScalar(std::isnan(r[0]) ? big : r[0], true);
        const double e = s->allReduceScalar(std::isnan(r[1]) ? big : r[1], true);
        if (eta_max) *eta_max = e;
        if (dt) *dt = cfl / ((s->N + 1) * (s->N + 1) * 0.5 * f);
        if (std::isinf(f) || std::isinf(e) || std::fabs(e) > 1e8)
            throw unstable_error("A numerical instability has occurred!");
    });
}

int bdg_sw2d_allreduce_max(bdg_sw2d* s, double value, double* out) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_allreduce_max");
        if (!out) throw arg_error("bdg_sw2d_allreduce_max: out is NULL");
        s->use();
        *out = s->allReduceScalar(value, true);
    });
}

int bdg_sw2d_allreduce_sum(bdg_sw2d* s, double value, double* out) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_allreduce_sum");
        if (!out) throw arg_error("bdg_sw2d_allreduce_sum: out is NULL");
        s->use();
        *out = s->allReduceScalar(value, false, true);
    });
}

int bdg_sw2d_barrier(bdg_sw2d* s) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_barrier");
        s->use();
        hipCheck(hipStreamSynchronize(s->stream), "hipStreamSynchronize");
        if (s->commStream) hipCheck(hipStreamSynchronize(s->commStream), "hipStreamSynchronize");
        // every rank arrives before anyone leaves -- and a wait that gave up on ANY rank is reported by ALL of them here (a rank that
        // threw before the reduction would leave the others inside it), so that the callers can react together
        const bool mine = s->takeSyncMark();
        const double any = s->allReduceScalar(mine ? 1.0 : 0.0, true);
        hipCheck(hipDeviceSynchronize(), "hipDeviceSynchronize");
        if (mine) throw std::runtime_error(bdg_sw2d::syncErrorText());
        if (any != 0.0) throw std::runtime_error(std::string("on another rank: ") + bdg_sw2d::syncErrorText());
    });
}

int bdg_probe_stream_triad(int device, size_t bytes_per_array, int repeats, double* gbps) {
    return guard([&] {
        if (!gbps || repeats < 1 || bytes_per_array < (1u << 20)) throw arg_error("bdg_probe_stream_triad: bad argument");
        hipCheck(hipSetDevice(device), "hipSetDevice");
        const size_t n2 = bytes_per_array / sizeof(double2);
        size_t total = 0;
        DevBuf<double> a, b, c;
        a.alloc(2 * n2, total); b.alloc(2 * n2, total); c.alloc(2 * n2, total);
        hipCheck(hipMemset(a.p, 0, 2 * n2 * sizeof(double)), "hipMemset");
        hipCheck(hipMemset(b.p, 0, 2 * n2 * sizeof(double)), "hipMemset");
        hipCheck(hipMemset(c.p, 0, 2 * n2 * sizeof(double)), "hipMemset");
        hipEvent_t e0, e1;
        hipCheck(hipEventCreate(&e0), "hipEventCreate");
        hipCheck(hipEventCreate(&e1), "hipEventCreate");
        auto launch = [&] {
            hipLaunchKernelGGL(bdg_dev::triad_kernel, dim3(256 * 8), dim3(256), 0, nullptr, reinterpret_cast<double2*>(a.p),
                               reinterpret_cast<const double2*>(b.p), reinterpret_cast<const double2*>(c.p), 0.5, n2);
        };
        launch();
        hipCheck(hipEventRecord(e0, nullptr), "hipEventRecord");
        for (int i = 0; i < repeats; ++i) launch();
        hipCheck(hipEventRecord(e1, nullptr), "hipEventRecord");
        hipCheck(hipEventSynchronize(e1), "hipEventSynchronize");
        float ms = 0.f;
        hipCheck(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *gbps = 3.0 * static_cast<double>(n2) * sizeof(double2) * repeats / (ms * 1e-3) / 1e9;
    });
}

int bdg_sw2d_probe_stage_traffic(bdg_sw2d* s, int repeats, float* ms_per_launch) {
    return guard([&] {
        requireSolver(s, "bdg_sw2d_probe_stage_traffic");
        if (!ms_per_launch || repeats < 1) throw arg_error("bdg_sw2d_probe_stage_traffic: bad argument");
        if (!s->affine) throw arg_error("bdg_sw2d_probe_stage_traffic: affine geometry only");
        s->use();
        const unsigned grid = static_cast<unsigned>((s->K + 255) / 256);
        auto launch = [&] {
            // aux / qalt are scratch here: the resident state is not modified
            hipLaunchKernelGGL(bdg_dev::stage_traffic_probe_kernel, dim3(grid), dim3(256), 0, s->stream, s->qcur, s->qalt,
                               s->aux.p, s->ageo.p, s->vmapP.p, 3 * s->Np, s->NFN, s->ld, s->K);
        };
        launch();
        hipCheck(hipEventRecord(s->ev0, s->stream), "hipEventRecord");
        for (int i = 0; i < repeats; ++i) launch();
        hipCheck(hipEventRecord(s->ev1, s->stream), "hipEventRecord");
        hipCheck(hipEventSynchronize(s->ev1), "hipEventSynchronize");
        float ms = 0.f;
        hipCheck(hipEventElapsedTime(&ms, s->ev0, s->ev1), "hipEventElapsedTime");
        *ms_per_launch = ms / repeats;
    });
}

int bdg_sw2d_uses_affine_geometry(const bdg_sw2d* s) { return s ? (s->affine ? 1 : 0) : -1; }
int bdg_sw2d_is_renumbered(const bdg_sw2d* s) { return s ? (s->permHost.empty() ? 0 : 1) : -1; }

size_t bdg_sw2d_device_bytes(const bdg_sw2d* s) { return s ? s->bytes : 0; }
void* bdg_sw2d_stream(bdg_sw2d* s) { return s ? static_cast<void*>(s->stream) : nullptr; }

} // extern "C"
