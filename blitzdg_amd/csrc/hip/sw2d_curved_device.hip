// C ABI of the curved / over-integrated sw2d solver (include/blitzdg_hip.h, bdg_sw2d_curved_*): table
// validation, device layout, operator image in MFMA tile order, launches. Kernels: sw2d_curved_kernel.hpp.
// Reference: swhelpers/rhs.py:6-176 (the RHS), sw2d_curved.py:246-277 (the time loop).
#include "../host/capi_internal.hpp"
#include "../host/parallel_for.hpp"
#include "blitzdg/LSERK4.hpp"
#include "rccl_api.hpp"
#include "sw2d_curved_kernel.hpp"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace bdg_dev {
const CurvedKernelTable* curved_kernel_table_order1();
const CurvedKernelTable* curved_kernel_table_order2();
const CurvedKernelTable* curved_kernel_table_order3();
const CurvedKernelTable* curved_kernel_table_order4();
const CurvedKernelTable* curved_kernel_table_order5();
const CurvedKernelTable* curved_kernel_table_order6();
const CurvedKernelTable* curved_kernel_table_order7();
const CurvedKernelTable* curved_kernel_table_order8();

const CurvedKernelTable* curved_kernel_table(int order) {
    switch (order) {
    case 1: return curved_kernel_table_order1();
    case 2: return curved_kernel_table_order2();
    case 3: return curved_kernel_table_order3();
    case 4: return curved_kernel_table_order4();
    case 5: return curved_kernel_table_order5();
    case 6: return curved_kernel_table_order6();
    case 7: return curved_kernel_table_order7();
    case 8: return curved_kernel_table_order8();
    default: return nullptr;
    }
}
} // namespace bdg_dev

using bdg_detail::arg_error;
using bdg_detail::guard;
using bdg_detail::hip_error;

namespace {

void hipOk(hipError_t e, const char* what) {
    if (e != hipSuccess) throw hip_error(std::string(what) + ": " + hipGetErrorString(e));
}

template <typename T>
struct Buf {
    T* p = nullptr;
    size_t n = 0;
    void alloc(size_t count, size_t& total, hipStream_t stream) {
        if (p) total -= std::min(total, n * sizeof(T)); // a replaced buffer no longer counts (repeated set_partition)
        release();
        if (count == 0) return;
        hipOk(hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)), "hipMalloc");
        n = count;
        total += count * sizeof(T);
        hipOk(hipMemsetAsync(p, 0, count * sizeof(T), stream), "hipMemset");
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~Buf() { release(); }
};

// Ghost exchange of a partitioned run: one record of 4 Np doubles ([field][node]) per element. Pack: the elements a
// neighbour needs, in the order of the send list; unpack: the received records into the ghost columns (num_owned ...).
__global__ __launch_bounds__(256) void sw2d_curved_pack_kernel(const double* __restrict__ q, long long ld, int rows,
                                                               const int* __restrict__ sendEls, int numSend, double* __restrict__ out) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= static_cast<long long>(numSend) * rows) return;
    const int rec = static_cast<int>(i / rows), row = static_cast<int>(i - static_cast<long long>(rec) * rows);
    out[i] = q[static_cast<long long>(row) * ld + sendEls[rec]];
}
__global__ __launch_bounds__(256) void sw2d_curved_unpack_kernel(double* __restrict__ q, long long ld, int rows, int firstGhost,
                                                                 int numGhost, const double* __restrict__ in) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= static_cast<long long>(numGhost) * rows) return;
    const int row = static_cast<int>(i / numGhost), gcol = static_cast<int>(i - static_cast<long long>(row) * numGhost); // coalesced writes
    q[static_cast<long long>(row) * ld + firstGhost + gcol] = in[static_cast<long long>(gcol) * rows + row];
}

} // namespace

struct bdg_sw2d_curved {
    const bdg_dev::CurvedKernelTable* kt = nullptr;
    int N = 0, Np = 0, K = 0, device = 0, ncub = 0, ncb = 0, ng = 0, fb = 0, numCurved = 0;
    long long ld = 0, sideLd = 0;
    bool hasFilter = false, identityM = true;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    size_t bytes = 0;
    Buf<double> qA, qB, res, rhs, gq, cubG, gaussG, coef, mmSide, minvSide, ops, filt;
    // nodal coefficient planes inside coef (one allocation: the stage kernel reaches all of them through one buffer
    // descriptor): 1 / J first, then whichever of zx, zy, f, CD the caller gave (nullptr: absent)
    double *rJ = nullptr, *zx = nullptr, *zy = nullptr, *fcor = nullptr, *cd = nullptr;
    Buf<int> gmapP, gmapM, curvedSlot, curvedEls, affineEl;
    Buf<double> cubAffine, cubWref;
    int numAffine = 0;
    // nodal-trace form (sw2d_curved_nt_kernel.hpp): used when the context has the structure it needs (useNT)
    bool useNT = false;
    Buf<double> opsNT, elAffine, gaussWref;
    Buf<int> nodeP, faceFlags, faceNodesDev, tileOrder;
    int numAffineNT = 0;
    int prioMode = 2;   // see sw2d_curved_nt_kernel (BDG_SW2D_CURVED_PRIO at creation: A/B switch)
    double g = 9.81, fconst = 0.0, cdconst = 0.0;
    long long stageCount = 0;
    double bytesPerElement = 0.0;
    // partitioned runs (bdg_sw2d_curved_set_partition / _comm_init): elements [numOwned, K) are ghosts, refreshed from their
    // owners before every evaluation by grouped ncclSend / ncclRecv on the solver's stream
    struct Peer { int rank, sendStart, sendCount, recvStart, recvCount; };
    int numOwned = 0, numInterior = 0, numSend = 0, commRank = 0, commWorld = 1;
    Buf<int> sendEls;
    // overlapped schedule (nodal-trace form): elements [0, numInterior) have no ghost neighbour and are evaluated on the solver's
    // stream while the exchange and then the partition-boundary elements [numInterior, numOwned) run on commStream; the curved
    // elements of the two ranges (columns of the side buffer) are listed for the fix-up launches of either chain
    std::vector<int> curvedHost;          // element of each side-buffer column
    std::vector<int> maxNeighbourHost;    // largest element index a face of element k is paired with (from gmapP, kept for set_partition)
    Buf<int> slotsInterior, slotsBoundary;
    int numSlotsInterior = 0, numSlotsBoundary = 0;
    hipStream_t commStream = nullptr;
    hipEvent_t evA[2] = {nullptr, nullptr}, evB[2] = {nullptr, nullptr}, evEntry = nullptr;
    Buf<double> sendBuf, recvBuf, scalarBuf;
    std::vector<Peer> peers;
    ncclComm_t comm = nullptr;

    ~bdg_sw2d_curved() {
        if (comm) (void)bdg_rccl::rccl().CommDestroy(comm);
        for (hipEvent_t e : {evA[0], evA[1], evB[0], evB[1], evEntry})
            if (e) (void)hipEventDestroy(e);
        if (commStream) (void)hipStreamDestroy(commStream);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (stream) (void)hipStreamDestroy(stream);
    }
    void use() const { hipOk(hipSetDevice(device), "hipSetDevice"); }
    size_t plane() const { return static_cast<size_t>(Np) * static_cast<size_t>(ld); }

    // host (rows, K) row-major -> device rows [row0, row0 + rows) of a (., ld) plane
    template <typename T>
    void uploadRows(const T* host, T* dev, int rows, int row0 = 0) {
        hipOk(hipMemcpy2DAsync(dev + static_cast<size_t>(row0) * ld, static_cast<size_t>(ld) * sizeof(T), host,
                               static_cast<size_t>(K) * sizeof(T), static_cast<size_t>(K) * sizeof(T), rows,
                               hipMemcpyHostToDevice, stream), "H2D copy");
        hipOk(hipStreamSynchronize(stream), "upload sync");
    }
    void downloadRows(const double* dev, double* host, int rows) {
        hipOk(hipMemcpy2DAsync(host, static_cast<size_t>(K) * sizeof(double), dev, static_cast<size_t>(ld) * sizeof(double),
                               static_cast<size_t>(K) * sizeof(double), rows, hipMemcpyDeviceToHost, stream), "D2H copy");
        hipOk(hipStreamSynchronize(stream), "download sync");
    }

    bdg_dev::CurvedParams params() const {
        bdg_dev::CurvedParams p{};
        p.gq = gq.p; p.cubG = cubG.p; p.gaussG = gaussG.p; p.gmapP = gmapP.p; p.gmapM = identityM ? nullptr : gmapM.p;
        p.rJ = rJ; p.zx = zx; p.zy = zy; p.fcor = fcor; p.cd = cd; p.fconst = fconst; p.cdconst = cdconst;
        p.curvedSlot = numCurved ? curvedSlot.p : nullptr;
        p.mmSide = mmSide.p; p.minvSide = minvSide.p; p.curvedEls = curvedEls.p; p.numCurved = numCurved; p.sideLd = sideLd;
        p.affineEl = numAffine ? affineEl.p : nullptr; p.cubAffine = numAffine ? cubAffine.p : nullptr; p.cubWref = cubWref.p;
        p.ops = ops.p; p.filt = filt.p; p.ld = ld; p.K = K; p.ncb = ncb; p.ncub = ncub; p.ng = ng; p.fb = fb; p.g = g;
        if (useNT) {
            p.opsNT = opsNT.p; p.nodeP = nodeP.p; p.faceFlags = faceFlags.p; p.faceNodes = faceNodesDev.p; p.gaussWref = gaussWref.p;
            p.affineEl = nullptr; p.elAffine = elAffine.p; // (the straight-element flag rides in faceFlags)
            static const int interleave = [] { const char* e = std::getenv("BDG_SW2D_TILE_INTERLEAVE"); return e ? std::atoi(e) : 1; }();
            p.tileInterleave = interleave;
            p.tileOrder = tileOrder.p;
            p.prioMode = prioMode;
        }
        return p;
    }

    // One RHS evaluation fused with its update: Gauss traces of qin, stage kernel, curved-element kernel.
    // the same for the elements [kbegin, kend) and the listed columns of the side buffer only, on a given stream (nodal-trace form)
    void evaluateRange(int mode, bool filter, const double* qin, const double* qbase, double* qout, double ca, double cb, double cc,
                       int kbegin, int kend, const int* slotList, int numSlots, hipStream_t on, int gridReserve = 0) {
        if (filter && !hasFilter) throw arg_error("bdg_sw2d_curved: filter requested but the solver was created without a Filter matrix");
        if (!useNT || qin == qout) throw arg_error("bdg_sw2d_curved: range evaluations need the nodal-trace form and two state buffers");
        bdg_dev::CurvedParams p = params();
        p.qin = qin; p.qbase = qbase; p.qout = qout; p.res = res.p; p.rhs = rhs.p; p.ca = ca; p.cb = cb; p.cc = cc;
        p.kbegin = kbegin; p.K = kend; p.gridReserve = gridReserve;
        p.tileOrder = nullptr;  // (the list is built for the tiles of [0, K))
        hipOk(kt->stageNT(mode, filter, p, on), "sw2d_curved_nt_kernel");
        p.slotList = slotList; p.numCurved = numSlots;
        hipOk(kt->fixup(mode, filter, p, on), "sw2d_curved_fixup_kernel");
    }

    void evaluate(int mode, bool filter, const double* qin, const double* qbase, double* qout, double ca, double cb,
                  double cc) {
        if (filter && !hasFilter) throw arg_error("bdg_sw2d_curved: filter requested but the solver was created without a Filter matrix");
        bdg_dev::CurvedParams p = params();
        p.qin = qin; p.qbase = qbase; p.qout = qout; p.res = res.p; p.rhs = rhs.p; p.ca = ca; p.cb = cb; p.cc = cc;
        if (useNT) { // one launch: the neighbours' traces are products of their nodal values (no Gauss-trace planes)
            if (qin == qout) throw arg_error("bdg_sw2d_curved: the nodal-trace kernel cannot update the state it gathers from in place");
            hipOk(kt->stageNT(mode, filter, p, stream), "sw2d_curved_nt_kernel");
        } else {
            hipOk(kt->gauss(p, stream), "sw2d_curved_gauss_kernel");
            hipOk(kt->stage(mode, filter, p, stream), "sw2d_curved_stage_kernel");
        }
        hipOk(kt->fixup(mode, filter, p, stream), "sw2d_curved_fixup_kernel");
    }

    // one LSERK4 stage: res = a res + dt RHS(q); q += b res. The first form updates q in place (it reads neighbours
    // through the trace planes); the nodal-trace form reads neighbours' nodes from q, so it writes the other buffer
    // and the two swap roles.
    void lserkStage(double a, double b, double dt) {
        if (useNT) {
            evaluate(bdg_dev::CMODE_LSERK, false, qA.p, nullptr, qB.p, a, b, dt);
            std::swap(qA.p, qB.p);
        } else {
            evaluate(bdg_dev::CMODE_LSERK, false, qA.p, nullptr, qA.p, a, b, dt);
        }
    }

    // ghost columns of `state` from their owners (pack -> grouped send / receive with every neighbour -> unpack), in stream order
    void exchange(double* state) { exchangeOn(state, stream); }
    void exchangeOn(double* state, hipStream_t stream) {
        if (!comm) throw arg_error("bdg_sw2d_curved: no communicator (call bdg_sw2d_curved_comm_init first)");
        const int rows = 4 * Np, ghosts = K - numOwned;
        if (numSend > 0) {
            const long long n = static_cast<long long>(numSend) * rows;
            hipLaunchKernelGGL(sw2d_curved_pack_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, state, ld,
                               rows, sendEls.p, numSend, sendBuf.p);
            hipOk(hipGetLastError(), "sw2d_curved_pack_kernel");
        }
        if (!peers.empty()) {
            bdg_rccl::RcclApi& nc = bdg_rccl::rccl();
            bdg_rccl::ncclCheck(nc.GroupStart(), "ncclGroupStart");
            for (const Peer& pr : peers) {
                if (pr.recvCount > 0)
                    bdg_rccl::ncclCheck(nc.Recv(recvBuf.p + static_cast<size_t>(pr.recvStart) * rows, static_cast<size_t>(pr.recvCount) * rows,
                                                ncclDouble, pr.rank, comm, stream), "ncclRecv");
                if (pr.sendCount > 0)
                    bdg_rccl::ncclCheck(nc.Send(sendBuf.p + static_cast<size_t>(pr.sendStart) * rows, static_cast<size_t>(pr.sendCount) * rows,
                                                ncclDouble, pr.rank, comm, stream), "ncclSend");
            }
            bdg_rccl::ncclCheck(nc.GroupEnd(), "ncclGroupEnd");
        }
        if (ghosts > 0) {
            const long long n = static_cast<long long>(ghosts) * rows;
            hipLaunchKernelGGL(sw2d_curved_unpack_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, state, ld,
                               rows, numOwned, ghosts, recvBuf.p);
            hipOk(hipGetLastError(), "sw2d_curved_unpack_kernel");
        }
    }
    // the driver's RK2 step of a partitioned run: an exchange in front of EACH evaluation, of the state that evaluation reads
    //
    // Nodal-trace form with interior elements: two chains, as the straight-element solver's stage loop (sw2d_device.hip):
    //   solver stream A:  wait B(e-1) -> [elements without a ghost neighbour of evaluation e] -> signal A(e)
    //   comm stream   B:  wait A(e-1) -> pack, grouped send / receive, unpack of the state e reads -> [partition-boundary
    //                     elements of e] -> signal B(e)
    // interior(e) reads the boundary elements' columns boundary(e-1) wrote and overwrites columns boundary(e-1) read: it waits
    // for B(e-1); boundary(e) and its pack read / overwrite columns interior(e-1) wrote / read: B waits for A(e-1). Ghost
    // columns are written by the unpack and read by the boundary launch only, both on B. Ghost elements are not evaluated.
    // Otherwise (general form, or no interior elements): exchange, then every element, in stream order.
    struct Chains { bool haveA = false, haveB = false; int e = 0; };
    bool overlapped() const {
        return useNT && numInterior >= 1 && std::getenv("BDG_SW2D_CURVED_NO_OVERLAP") == nullptr; // (A/B switch, read per call)
    }
    void chainsBegin(Chains&) {
        hipOk(hipEventRecord(evEntry, stream), "hipEventRecord");          // whatever set the state, on A
        hipOk(hipStreamWaitEvent(commStream, evEntry, 0), "hipStreamWaitEvent");
    }
    // one evaluation on both chains: reads `in` (ghost columns refreshed first), writes the owned columns of `out`
    void chainsEval(Chains& c, int mode, bool filter, double* in, const double* base, double* out, double ca, double cb, double cc) {
        // the interior launch is one resident round of workgroups that loop over their tiles: it leaves the slots the
        // partition-boundary launch needs (a workgroup per four tiles, rounded up to the eight XCDs), or that launch would wait
        const int boundaryWgs = (((numOwned - numInterior + 15) / 16 + 3) / 4 + 7) / 8 * 8;
        const int cur = c.e & 1, prev = cur ^ 1;
        // ---- chain A
        if (c.haveB) hipOk(hipStreamWaitEvent(stream, evB[prev], 0), "hipStreamWaitEvent");
        evaluateRange(mode, filter, in, base, out, ca, cb, cc, 0, numInterior, slotsInterior.p, numSlotsInterior, stream, boundaryWgs);
        hipOk(hipEventRecord(evA[cur], stream), "hipEventRecord");
        // ---- chain B
        if (c.haveA) hipOk(hipStreamWaitEvent(commStream, evA[prev], 0), "hipStreamWaitEvent");
        exchangeOn(in, commStream);
        evaluateRange(mode, filter, in, base, out, ca, cb, cc, numInterior, numOwned, slotsBoundary.p, numSlotsBoundary, commStream);
        hipOk(hipEventRecord(evB[cur], commStream), "hipEventRecord");
        c.haveA = c.haveB = true;
        ++c.e;
    }
    void chainsEnd(Chains& c) { // join both ways: later work on A sees the last boundary update, later work on B the last interior launch
        if (c.e == 0) return;
        hipOk(hipStreamWaitEvent(stream, evB[(c.e - 1) & 1], 0), "hipStreamWaitEvent");
        hipOk(hipStreamWaitEvent(commStream, evA[(c.e - 1) & 1], 0), "hipStreamWaitEvent");
    }
    void stepRk2Exchanged(double dt, int steps, bool filter) {
        if (!overlapped()) {
            for (int i = 0; i < steps; ++i) {
                exchange(qA.p);
                evaluate(bdg_dev::CMODE_COMBINE, filter, qA.p, qA.p, qB.p, 1.0, 0.0, 0.5 * dt);
                exchange(qB.p);
                evaluate(bdg_dev::CMODE_COMBINE, filter, qB.p, qA.p, qA.p, 1.0, 0.0, dt);
            }
            return;
        }
        Chains c;
        chainsBegin(c);
        for (int i = 0; i < steps; ++i) {
            chainsEval(c, bdg_dev::CMODE_COMBINE, filter, qA.p, qA.p, qB.p, 1.0, 0.0, 0.5 * dt);
            chainsEval(c, bdg_dev::CMODE_COMBINE, filter, qB.p, qA.p, qA.p, 1.0, 0.0, dt);
        }
        chainsEnd(c);
    }
    // LSERK4 stages of a partitioned run (reference src/advec1d/main.cpp:92-102 per stage): res = a res + dt RHS(q); q += b res,
    // an exchange of q in front of every stage; the nodal-trace form writes the other state buffer and the two swap roles
    void lserkStagesExchanged(double dt, int numStages) {
        Chains c;
        const bool two = overlapped();
        if (two) chainsBegin(c);
        for (int i = 0; i < numStages; ++i) {
            const int st = static_cast<int>(stageCount % blitzdg::LSERK4::numStages);
            const double a = blitzdg::LSERK4::rk4a[st], b = blitzdg::LSERK4::rk4b[st];
            if (two) {
                chainsEval(c, bdg_dev::CMODE_LSERK, false, qA.p, nullptr, qB.p, a, b, dt);
                std::swap(qA.p, qB.p);
            } else {
                exchange(qA.p);
                lserkStage(a, b, dt);
            }
            ++stageCount;
        }
        if (two) chainsEnd(c);
    }
    void stepRk2(double dt, int steps, bool filter) {
        for (int i = 0; i < steps; ++i) {
            evaluate(bdg_dev::CMODE_COMBINE, filter, qA.p, qA.p, qB.p, 1.0, 0.0, 0.5 * dt); // predictor: q1 = q + dt/2 RHS(q)
            evaluate(bdg_dev::CMODE_COMBINE, filter, qB.p, qA.p, qA.p, 1.0, 0.0, dt);       // corrector: q += dt RHS(q1)
        }
    }
};

namespace {

void requireCurved(const bdg_sw2d_curved* s, const char* fn) {
    if (!s) throw arg_error(std::string(fn) + ": solver handle is NULL");
}

std::vector<double> matmul(const double* A, const double* B, int n, int m, int c) { // (n,m) (m,c)
    std::vector<double> C(static_cast<size_t>(n) * c, 0.0);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < m; ++k) {
            const double a = A[static_cast<size_t>(i) * m + k];
            for (int j = 0; j < c; ++j) C[static_cast<size_t>(i) * c + j] += a * B[static_cast<size_t>(k) * c + j];
        }
    return C;
}

// Tables of the nodal-trace form (sw2d_curved_nt_kernel.hpp), or useNT = false when the context lacks the structure:
//  (1) gmapM is the identity; (2) every row block of Interp that belongs to a face is zero outside N + 1 columns (its face
//  nodes); (3) gmapP pairs whole faces, in the same or in the opposite direction, and the paired faces' interpolation
//  columns agree at the paired Gauss points under one permutation of their face nodes; (4) a face's Gauss points are all
//  walls or none; (5) the image fits the kernel's LDS plan. BDG_SW2D_CURVED_GENERAL=1 keeps the first form (A/B, cross-check).
void buildNodalTraceTables(bdg_sw2d_curved& s, const bdg_sw2d_curved_desc& d, const std::vector<int>& slotOf) {
    s.useNT = false;
    if (std::getenv("BDG_SW2D_CURVED_GENERAL") || !s.identityM) return;
    const bdg_dev::CurvedKernelTable* kt = s.kt;
    const int N = s.N, Np = s.Np, K = s.K, Ncub = s.ncub, NG = s.ng, NG3 = 3 * NG, ncb = s.ncb, fb = s.fb, Nfp = N + 1;
    const int CR = 16 * ncb, KV = kt->KV, MT = kt->MT;
    const long long ld = s.ld;
    if (!kt->ntFits(ncb, fb, d.Filter != nullptr)) return;
    if (4ll * CR * ld * 8 > 4294967295LL) return; // the cubature planes behind ONE descriptor
    int off[5];
    kt->ntOffsets(ncb, fb, off);
    const int VCH = off[0], SCH = off[1], KE = off[2], offSurf = off[3], offMass = off[4];
    const double* I = d.gaussInterp; // (3 NG, Np)
    // (2) face nodes
    std::vector<std::vector<int>> faceNodes(3);
    for (int f = 0; f < 3; ++f) {
        for (int m = 0; m < Np; ++m) {
            double big = 0.0;
            for (int ig = 0; ig < NG; ++ig) big = std::max(big, std::fabs(I[static_cast<size_t>(f * NG + ig) * Np + m]));
            if (big > 1e-9) faceNodes[f].push_back(m);
            else if (big > 1e-12) return; // neither a face node nor nothing
        }
        if (static_cast<int>(faceNodes[f].size()) != Nfp) return;
    }
    auto IF = [&](int f, int ig, int i) { return I[static_cast<size_t>(f * NG + ig) * Np + faceNodes[f][i]]; };
    // (3) node pairing of (f, f2, direction): perm[i] = i2 with IF(f, ig, i) == IF(f2, ig2(ig), i2) for every ig
    std::vector<int> perm(static_cast<size_t>(18) * Nfp, -1);
    std::vector<int> permState(18, 0); // 0 not tried, 1 found, -1 none
    auto pairing = [&](int f, int f2, int rev) -> const int* {
        const int id = (f * 3 + f2) * 2 + rev;
        int* out = perm.data() + static_cast<size_t>(id) * Nfp;
        if (permState[id] == 0) {
            std::vector<char> used(Nfp, 0);
            bool ok = true;
            for (int i = 0; i < Nfp && ok; ++i) {
                int found = -1;
                for (int i2 = 0; i2 < Nfp && found < 0; ++i2) {
                    if (used[i2]) continue;
                    double worst = 0.0;
                    for (int ig = 0; ig < NG; ++ig) worst = std::max(worst, std::fabs(IF(f, ig, i) - IF(f2, rev ? NG - 1 - ig : ig, i2)));
                    if (worst < 1e-11) found = i2;
                }
                if (found < 0) ok = false;
                else { used[found] = 1; out[i] = found; }
            }
            permState[id] = ok ? 1 : -1;
        }
        return permState[id] == 1 ? out : nullptr;
    };
    // (3), (4) per face of every element
    const int rowsP = 3 * KE * 4;
    std::vector<int> nodeP(static_cast<size_t>(rowsP) * K), flags(static_cast<size_t>(ld), 0), wallCount(static_cast<size_t>(3) * K, 0);
    for (int i = 0; i < d.num_wall; ++i) {
        const int w = d.gmapW[i];
        ++wallCount[static_cast<size_t>(w / NG3) * 3 + (w % NG3) / NG];
    }
    bool structured = true;
    for (int k = 0; k < K && structured; ++k)
        for (int f = 0; f < 3 && structured; ++f) {
            const int* gp = d.gmapP + static_cast<size_t>(k) * NG3 + f * NG;
            const int k2 = gp[0] / NG3, f2 = (gp[0] % NG3) / NG, l0 = gp[0] % NG;
            int rev;
            if (l0 == 0 && (NG == 1 || gp[NG - 1] % NG == NG - 1)) rev = 0;
            else if (l0 == NG - 1) rev = 1;
            else { structured = false; break; }
            for (int ig = 0; ig < NG; ++ig)
                if (gp[ig] != k2 * NG3 + f2 * NG + (rev ? NG - 1 - ig : ig)) structured = false;
            const int* pm = structured ? pairing(f, f2, rev) : nullptr;
            if (!pm) { structured = false; break; }
            const int wc = wallCount[static_cast<size_t>(k) * 3 + f];
            if (wc != 0 && wc != NG) { structured = false; break; }
            if (wc) flags[k] |= 1 << f;
            for (int i = 0; i < KE * 4; ++i)
                nodeP[static_cast<size_t>(f * KE * 4 + i) * K + k] =
                    i < Nfp ? static_cast<int>(faceNodes[f2][pm[i]] * ld + k2) : k; // padding rows: an own node (the column of GE is zero)
        }
    if (!structured) return;

    hipStream_t st = s.stream;
    // ---- straight-sided elements: the cubature numbers of the first form's test, the Gauss geometry and the nodal Jacobian
    std::vector<int> aff(static_cast<size_t>(ld), 0);
    std::vector<double> ea(static_cast<size_t>(14) * K, 0.0), gwHalf(static_cast<size_t>(16) * fb, 0.0), wref(static_cast<size_t>(CR), 0.0);
    int count = 0;
    if (!std::getenv("BDG_SW2D_CURVED_NO_AFFINE")) {
        const double* geo[4] = {d.cubrx, d.cubry, d.cubsx, d.cubsy};
        auto straight = [&](int k, const double* wr, const double* gw, double* e) { // fills e[0..13]; false: not straight
            if (slotOf[k] >= 0) return false;                                       // elements of curvedEls keep the general path
            double scale = 0.0;
            for (int t = 0; t < 4; ++t) scale = std::max(scale, std::fabs(geo[t][k]));
            for (int t = 0; t < 4; ++t)
                for (int i = 1; i < Ncub; ++i)
                    if (std::fabs(geo[t][static_cast<size_t>(i) * K + k] - geo[t][k]) > 1e-10 * scale) return false;
            const double j0 = d.J[k];
            for (int m = 1; m < Np; ++m)
                if (std::fabs(d.J[static_cast<size_t>(m) * K + k] - j0) > 1e-10 * std::fabs(j0)) return false;
            for (int f = 0; f < 3; ++f) {
                const size_t r0 = static_cast<size_t>(f) * NG * K + k;
                for (int ig = 1; ig < NG; ++ig)
                    if (std::fabs(d.gaussnx[r0 + static_cast<size_t>(ig) * K] - d.gaussnx[r0]) > 1e-10 ||
                        std::fabs(d.gaussny[r0 + static_cast<size_t>(ig) * K] - d.gaussny[r0]) > 1e-10) return false;
            }
            if (!wr) return true; // (the search for the reference element stops here)
            const double ratio = d.cubW[k] / wr[0];
            if (!(ratio > 0.0)) return false;
            for (int i = 1; i < Ncub; ++i)
                if (std::fabs(d.cubW[static_cast<size_t>(i) * K + k] - ratio * wr[i]) > 1e-10 * std::fabs(ratio * wr[i])) return false;
            for (int t = 0; t < 4; ++t) e[t] = ratio * geo[t][k];
            for (int f = 0; f < 3; ++f) {
                const size_t r0 = static_cast<size_t>(f) * NG * K + k;
                const double sf = d.gaussW[r0] / gw[0];
                if (!(sf > 0.0)) return false;
                for (int ig = 1; ig < NG; ++ig)
                    if (std::fabs(d.gaussW[r0 + static_cast<size_t>(ig) * K] - sf * gw[ig]) > 1e-10 * std::fabs(sf * gw[ig])) return false;
                e[4 + 3 * f] = d.gaussnx[r0]; e[5 + 3 * f] = d.gaussny[r0]; e[6 + 3 * f] = sf;
            }
            e[13] = 1.0 / j0;
            return true;
        };
        int kref = -1;
        for (int k = 0; k < K && kref < 0; ++k)
            if (straight(k, nullptr, nullptr, nullptr) && d.cubW[k] > 0.0 && d.gaussW[k] > 0.0) kref = k;
        if (kref >= 0) {
            std::vector<double> gw(static_cast<size_t>(NG));
            for (int i = 0; i < Ncub; ++i) wref[i] = d.cubW[static_cast<size_t>(i) * K + kref];
            for (int ig = 0; ig < NG; ++ig) { gw[ig] = d.gaussW[static_cast<size_t>(ig) * K + kref]; gwHalf[ig] = 0.5 * gw[ig]; }
            std::atomic<int> counted{0};
            blitzdg::detail::parallelChunks(K, [&](int kBegin, int kEnd) {
                int mine = 0;
                double e[14];
                for (int k = kBegin; k < kEnd; ++k) {
                    if (!straight(k, wref.data(), gw.data(), e)) continue;
                    aff[k] = 1;
                    for (int i = 0; i < 14; ++i) ea[static_cast<size_t>(i) * K + k] = e[i];
                    ++mine;
                }
                counted += mine;
            });
            count = counted;
        }
    }
    s.numAffineNT = count;
    for (int k = 0; k < K; ++k) flags[k] |= aff[k] ? 8 : 0;          // bit 3: straight element (bits 0..2: wall faces)
    for (long long k = K; k < ld; ++k) flags[k] = flags[K - 1];       // padding lanes repeat the last element
    s.elAffine.alloc(static_cast<size_t>(14) * ld, s.bytes, st);
    s.uploadRows(ea.data(), s.elAffine.p, 14);
    if (!s.cubWref.p) { // (the first form's reference weights when it found straight elements: the same numbers)
        s.cubWref.alloc(wref.size(), s.bytes, st);
        hipOk(hipMemcpyAsync(s.cubWref.p, wref.data(), wref.size() * sizeof(double), hipMemcpyHostToDevice, st), "Wref upload");
    }
    s.gaussWref.alloc(gwHalf.size(), s.bytes, st);
    hipOk(hipMemcpyAsync(s.gaussWref.p, gwHalf.data(), gwHalf.size() * sizeof(double), hipMemcpyHostToDevice, st), "gauss Wref upload");
    s.faceFlags.alloc(static_cast<size_t>(ld), s.bytes, st);
    hipOk(hipMemcpyAsync(s.faceFlags.p, flags.data(), flags.size() * sizeof(int), hipMemcpyHostToDevice, st), "face flags upload");
    s.nodeP.alloc(static_cast<size_t>(rowsP) * ld, s.bytes, st);
    s.uploadRows(nodeP.data(), s.nodeP.p, rowsP);
    std::vector<int> fnodes(static_cast<size_t>(rowsP));
    for (int f = 0; f < 3; ++f)
        for (int i = 0; i < KE * 4; ++i) fnodes[static_cast<size_t>(f) * KE * 4 + i] = faceNodes[f][i < Nfp ? i : 0];
    s.faceNodesDev.alloc(fnodes.size(), s.bytes, st);
    hipOk(hipMemcpyAsync(s.faceNodesDev.p, fnodes.data(), fnodes.size() * sizeof(int), hipMemcpyHostToDevice, st), "face nodes upload");

    // ---- order of the tiles: positions [n x / 8, n (x + 1) / 8) of the list are XCD x's (sw2d_curved_nt_kernel); each eighth gets
    //      an eighth of the general tiles (in mesh order, first), then straight-sided ones (in mesh order)
    if (const char* e = std::getenv("BDG_SW2D_CURVED_PRIO")) s.prioMode = std::atoi(e);
    if (!std::getenv("BDG_SW2D_CURVED_NO_TILE_ORDER")) {
        const int ntiles = (K + 15) / 16;
        std::vector<int> general, straight, order;
        for (int t = 0; t < ntiles; ++t) {
            bool allStraight = true;
            for (int k = 16 * t; k < std::min(K, 16 * t + 16); ++k) allStraight = allStraight && aff[k];
            (allStraight ? straight : general).push_back(t);
        }
        const long long G = static_cast<long long>(general.size()), S = static_cast<long long>(straight.size());
        if (G > 0 && S > 0 && ntiles >= 64) {
            long long gi = 0, si = 0;
            for (long long x = 0; x < 8; ++x) {
                const long long cx = ntiles * (x + 1) / 8 - ntiles * x / 8;
                long long gq = std::min({G * (x + 1) / 8 - G * x / 8, cx, G - gi}), sq = cx - gq;
                if (sq > S - si) { sq = S - si; gq = cx - sq; }
                for (long long i = 0; i < gq; ++i) order.push_back(general[static_cast<size_t>(gi + i)]);
                for (long long i = 0; i < sq; ++i) order.push_back(straight[static_cast<size_t>(si + i)]);
                gi += gq; si += sq;
            }
            if (gi == G && si == S && static_cast<int>(order.size()) == ntiles) {
                s.tileOrder.alloc(order.size(), s.bytes, st);
                hipOk(hipMemcpyAsync(s.tileOrder.p, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice, st), "tile order upload");
                hipOk(hipStreamSynchronize(st), "tile order sync"); // (order is a local)
            }
        }
    }

    // ---- operator image (CurvedOpsNT)
    std::vector<double> img(static_cast<size_t>(kt->ntTiles(ncb, fb)) * 64, 0.0);
    auto at = [&](int tile, int l) -> double& { return img[static_cast<size_t>(tile) * 64 + l]; };
    std::vector<double> Vt(static_cast<size_t>(Np) * Np);
    for (int i = 0; i < Np; ++i)
        for (int jj = 0; jj < Np; ++jj) Vt[static_cast<size_t>(i) * Np + jj] = d.V[static_cast<size_t>(jj) * Np + i];
    const std::vector<double> M = matmul(d.V, Vt.data(), Np, Np, Np);
    std::vector<double> MF;
    if (d.Filter) MF = matmul(d.Filter, M.data(), Np, Np, Np);
    for (int l = 0; l < 64; ++l) {
        const int i = l & 15, sc = l >> 4;
        for (int rb = 0; rb < ncb; ++rb) {
            const int base = rb * VCH;
            for (int t = 0; t < KV; ++t) {
                const int row = 16 * rb + i, m = 4 * t + sc;
                if (row < Ncub && m < Np) at(base + t, l) = d.cubV[static_cast<size_t>(row) * Np + m];
            }
            for (int r = 0; r < MT; ++r)
                for (int reg = 0; reg < 4; ++reg) {
                    const int node = 16 * r + i, cp = 16 * rb + 4 * reg + sc;
                    if (node < Np && cp < Ncub) {
                        at(base + KV + r * 4 + reg, l) = d.cubDr[static_cast<size_t>(cp) * Np + node];
                        at(base + KV + 4 * MT + r * 4 + reg, l) = d.cubDs[static_cast<size_t>(cp) * Np + node];
                    }
                }
        }
        for (int gb = 0; gb < 3 * fb; ++gb) {
            const int base = offSurf + gb * SCH, f = gb / fb, b = gb % fb;
            for (int t2 = 0; t2 < KE; ++t2) {
                const int local = 16 * b + i, fn = 4 * t2 + sc;
                if (local < NG && fn < Nfp) at(base + t2, l) = IF(f, local, fn);
            }
            for (int r = 0; r < MT; ++r)
                for (int reg = 0; reg < 4; ++reg) {
                    const int node = 16 * r + i, local = 16 * b + 4 * reg + sc;
                    if (node < Np && local < NG) at(base + KE + r * 4 + reg, l) = -I[static_cast<size_t>(f * NG + local) * Np + node];
                }
        }
        for (int r = 0; r < MT; ++r)
            for (int t = 0; t < KV; ++t) {
                const int node = 16 * r + i, m = 4 * t + sc;
                if (node < Np && m < Np) {
                    at(offMass + r * KV + t, l) = M[static_cast<size_t>(node) * Np + m];
                    if (d.Filter) {
                        at(offMass + MT * KV + r * KV + t, l) = MF[static_cast<size_t>(node) * Np + m];
                        at(offMass + 2 * MT * KV + r * KV + t, l) = d.Filter[static_cast<size_t>(node) * Np + m];
                    }
                }
            }
    }
    s.opsNT.alloc(img.size(), s.bytes, st);
    hipOk(hipMemcpyAsync(s.opsNT.p, img.data(), img.size() * sizeof(double), hipMemcpyHostToDevice, st), "ops upload");
    hipOk(hipStreamSynchronize(st), "nodal-trace tables sync");
    s.useNT = true;
}

bdg_sw2d_curved* createCurved(const bdg_sw2d_curved_desc& d) {
    const bdg_dev::CurvedKernelTable* kt = bdg_dev::curved_kernel_table(d.order);
    if (!kt) throw arg_error("bdg_sw2d_curved_create: order must be 1..8");
    if (d.num_elements < 1 || d.num_cub < 1 || d.num_gauss < 1) throw arg_error("bdg_sw2d_curved_create: bad sizes");
    if (d.num_gauss > 32) throw arg_error("bdg_sw2d_curved_create: at most 32 Gauss points per face");
    if (!d.V || !d.J || !d.cubV || !d.cubDr || !d.cubDs || !d.cubW || !d.cubrx || !d.cubry || !d.cubsx || !d.cubsy ||
        !d.gaussInterp || !d.gaussW || !d.gaussnx || !d.gaussny || !d.gmapM || !d.gmapP)
        throw arg_error("bdg_sw2d_curved_create: a required table pointer is NULL");
    if (d.num_wall < 0 || (d.num_wall > 0 && !d.gmapW)) throw arg_error("bdg_sw2d_curved_create: bad wall-node list");
    if (d.num_curved < 0 || (d.num_curved > 0 && (!d.curvedEls || !d.MMChol)))
        throw arg_error("bdg_sw2d_curved_create: curvedEls given without MMChol (or a negative count)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        throw hip_error("bdg_sw2d_curved_create: no HIP device available (the sw2d path has no CPU fallback)");
    if (d.device < 0 || d.device >= ndev) throw arg_error("bdg_sw2d_curved_create: device ordinal out of range");

    auto s = std::unique_ptr<bdg_sw2d_curved>(new bdg_sw2d_curved());
    s->kt = kt;
    s->N = d.order; s->Np = kt->Np; s->K = d.num_elements; s->device = d.device; s->g = d.g;
    s->ncub = d.num_cub; s->ncb = (d.num_cub + 15) / 16; s->ng = d.num_gauss; s->fb = (d.num_gauss + 15) / 16;
    s->ld = (static_cast<long long>(s->K) + 63) / 64 * 64;
    const int Np = s->Np, K = s->K, Ncub = s->ncub, NG = s->ng, NG3 = 3 * NG, ncb = s->ncb, fb = s->fb;
    const int CR = 16 * ncb, GR = 48 * fb, KV = kt->KV, MT = kt->MT;
    const long long ld = s->ld;
    // lane addresses are a row pointer plus an unsigned 32-bit BYTE offset within one plane
    // (the state, trace and coefficient arrays are reached through ONE descriptor each, the plane picked by a 32-bit scalar
    // offset: 4 state / trace planes, up to 5 coefficient planes; the cubature planes have a descriptor each)
    if (static_cast<long long>(std::max({5 * Np, CR, 4 * GR})) * ld * 8 > 4294967295LL)
        throw arg_error("bdg_sw2d_curved_create: a table exceeds 4 GiB (32-bit byte offsets): partition the mesh");

    // ---- index tables, validated on the host before anything touches the GPU
    const long long nG = static_cast<long long>(NG3) * K;
    std::vector<int> offP(static_cast<size_t>(GR) * K, 0), offM;
    bool identityM = true;
    for (long long i = 0; i < nG; ++i) {
        if (d.gmapP[i] < 0 || d.gmapP[i] >= nG || d.gmapM[i] < 0 || d.gmapM[i] >= nG)
            throw arg_error("bdg_sw2d_curved_create: gmapM / gmapP entry out of range");
        identityM = identityM && d.gmapM[i] == i;
    }
    auto devOffset = [&](int v) { // flat id g + 3NG k -> padded row * ld + k
        const int k2 = v / NG3, g2 = v % NG3, f2 = g2 / NG, l2 = g2 % NG;
        return static_cast<int>((16ll * fb * f2 + l2) * ld + k2);
    };
    if (!identityM) offM.assign(static_cast<size_t>(GR) * K, 0);
    blitzdg::detail::parallelFor(K, [&](int k) {
        for (int gI = 0; gI < NG3; ++gI) {
            const int f = gI / NG, l = gI % NG;
            const size_t at = static_cast<size_t>(16 * fb * f + l) * K + k;
            offP[at] = devOffset(d.gmapP[static_cast<size_t>(k) * NG3 + gI]);
            if (!identityM) offM[at] = devOffset(d.gmapM[static_cast<size_t>(k) * NG3 + gI]);
        }
    });
    std::vector<int> maxNeighbour(static_cast<size_t>(K), 0);
    blitzdg::detail::parallelFor(K, [&](int k) {
        int m = k;
        for (int gI = 0; gI < NG3; ++gI) m = std::max(m, d.gmapP[static_cast<size_t>(k) * NG3 + gI] / NG3);
        maxNeighbour[static_cast<size_t>(k)] = m;
    });
    s->maxNeighbourHost = std::move(maxNeighbour);
    for (int i = 0; i < d.num_wall; ++i) {
        const int w = d.gmapW[i];
        if (w < 0 || w >= nG) throw arg_error("bdg_sw2d_curved_create: wall Gauss-node index out of range");
        const int k = w / NG3, gI = w % NG3, f = gI / NG, l = gI % NG;
        int& e = offP[static_cast<size_t>(16 * fb * f + l) * K + k];
        if (e >= 0) e = -(e + 1);
    }
    std::vector<int> curved;
    std::vector<int> slotOf(static_cast<size_t>(K), -1);
    for (int i = 0; i < d.num_curved; ++i) {
        const int k = d.curvedEls[i];
        if (k < 0 || k >= K) throw arg_error("bdg_sw2d_curved_create: curvedEls entry out of range");
        if (slotOf[k] < 0) { // the reference takes set(curvedEls)
            slotOf[k] = static_cast<int>(curved.size());
            curved.push_back(k);
        }
    }
    s->numCurved = static_cast<int>(curved.size());
    s->sideLd = (static_cast<long long>(s->numCurved) + 63) / 64 * 64;
    s->identityM = identityM;

    s->use();
    hipOk(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking), "hipStreamCreate");
    hipOk(hipEventCreate(&s->ev0), "hipEventCreate");
    hipOk(hipEventCreate(&s->ev1), "hipEventCreate");
    hipStream_t st = s->stream;

    // ---- state and work planes
    const size_t plane4 = 4 * s->plane();
    s->qA.alloc(plane4, s->bytes, st);
    s->qB.alloc(plane4, s->bytes, st);
    s->res.alloc(plane4, s->bytes, st);
    s->rhs.alloc(plane4, s->bytes, st);
    s->gq.alloc(static_cast<size_t>(4) * GR * ld, s->bytes, st);

    // ---- cubature geometry: W rx, W ry, W sx, W sy (the reference forms W (rx F + ry G); the products are
    //      folded here once, a rounding-level difference), rows padded to 16 ncb with zeros
    s->cubG.alloc(static_cast<size_t>(4) * CR * ld, s->bytes, st);
    {
        std::vector<double> tmp(static_cast<size_t>(Ncub) * K);
        const double* geo[4] = {d.cubrx, d.cubry, d.cubsx, d.cubsy};
        for (int t = 0; t < 4; ++t) {
            blitzdg::detail::parallelFor(static_cast<long long>(Ncub) * K, [&](long long i) { tmp[i] = d.cubW[i] * geo[t][i]; }, 1 << 16);
            s->uploadRows(tmp.data(), s->cubG.p + static_cast<size_t>(t) * CR * ld, Ncub);
        }
    }
    // ---- straight-sided elements: rx, ry, sx, sy do not vary over the cubature points and W[i] = w_i J, so the four
    //      Ncub-row planes are 4 numbers per element times one vector of weights. Found here from the tables themselves
    //      (to 1e-10 relative -- the tables of a fine mesh carry that much round-off from Dr x; the reference element is the first one with constant metric terms):
    //      (W rx)[i, k] = Wref[i] * (W[0, k] / Wref[0]) * rx[0, k]. BDG_SW2D_CURVED_NO_AFFINE=1 keeps every element general.
    if (!std::getenv("BDG_SW2D_CURVED_NO_AFFINE")) {
        const double* geo[4] = {d.cubrx, d.cubry, d.cubsx, d.cubsy};
        std::vector<int> flag(static_cast<size_t>(ld), 0);
        auto constantMetric = [&](int k) {
            for (int t = 0; t < 4; ++t) {
                const double v0 = geo[t][k];
                double scale = 0.0;
                for (int t2 = 0; t2 < 4; ++t2) scale = std::max(scale, std::fabs(geo[t2][k]));
                for (int i = 1; i < Ncub; ++i)
                    if (std::fabs(geo[t][static_cast<size_t>(i) * K + k] - v0) > 1e-10 * scale) return false;
            }
            return true;
        };
        int kref = -1;
        for (int k = 0; k < K && kref < 0; ++k)
            if (constantMetric(k) && d.cubW[k] > 0.0) kref = k;
        if (kref >= 0) {
            std::vector<double> wref(static_cast<size_t>(CR), 0.0), ca(static_cast<size_t>(4) * K, 0.0);
            for (int i = 0; i < Ncub; ++i) wref[i] = d.cubW[static_cast<size_t>(i) * K + kref];
            std::atomic<int> counted{0};
            blitzdg::detail::parallelChunks(K, [&](int kBegin, int kEnd) {
                int mine = 0;
                for (int k = kBegin; k < kEnd; ++k) {
                    if (slotOf[k] >= 0 || !constantMetric(k)) continue; // elements of curvedEls keep the general path
                    const double ratio = d.cubW[k] / wref[0];
                    bool ok = ratio > 0.0;
                    for (int i = 1; i < Ncub && ok; ++i)
                        ok = std::fabs(d.cubW[static_cast<size_t>(i) * K + k] - ratio * wref[i]) <= 1e-10 * std::fabs(ratio * wref[i]);
                    if (!ok) continue;
                    flag[k] = 1;
                    for (int t = 0; t < 4; ++t) ca[static_cast<size_t>(t) * K + k] = ratio * geo[t][k];
                    ++mine;
                }
                counted += mine;
            });
            const int count = counted;
            s->numAffine = count;
            if (count) {
                for (long long k = K; k < ld; ++k) flag[k] = flag[K - 1]; // (padding lanes repeat the last element)
                s->affineEl.alloc(static_cast<size_t>(ld), s->bytes, st);
                hipOk(hipMemcpyAsync(s->affineEl.p, flag.data(), flag.size() * sizeof(int), hipMemcpyHostToDevice, st), "affine flags upload");
                s->cubAffine.alloc(static_cast<size_t>(4) * ld, s->bytes, st);
                s->uploadRows(ca.data(), s->cubAffine.p, 4);
                s->cubWref.alloc(wref.size(), s->bytes, st);
                hipOk(hipMemcpyAsync(s->cubWref.p, wref.data(), wref.size() * sizeof(double), hipMemcpyHostToDevice, st), "Wref upload");
                hipOk(hipStreamSynchronize(st), "affine upload sync");
            }
        }
    }
    // ---- Gauss geometry nx, ny, W and the maps, each face padded to 16 fb rows
    s->gaussG.alloc(static_cast<size_t>(3) * GR * ld, s->bytes, st);
    {
        const double* geo[3] = {d.gaussnx, d.gaussny, d.gaussW};
        for (int t = 0; t < 3; ++t)
            for (int f = 0; f < 3; ++f)
                s->uploadRows(geo[t] + static_cast<size_t>(f) * NG * K, s->gaussG.p + static_cast<size_t>(t) * GR * ld, NG, 16 * fb * f);
    }
    s->gmapP.alloc(static_cast<size_t>(GR) * ld, s->bytes, st);
    s->uploadRows(offP.data(), s->gmapP.p, GR);
    if (!identityM) {
        s->gmapM.alloc(static_cast<size_t>(GR) * ld, s->bytes, st);
        s->uploadRows(offM.data(), s->gmapM.p, GR);
    }
    // ---- nodal tables
    {
        const double* given[4] = {d.zx, d.zy, d.coriolis, d.drag};
        int planes = 1;
        for (const double* g4 : given) planes += g4 ? 1 : 0;
        s->coef.alloc(static_cast<size_t>(planes) * s->plane(), s->bytes, st);
        s->rJ = s->coef.p;
        std::vector<double> tmp(static_cast<size_t>(Np) * K);
        for (size_t i = 0; i < tmp.size(); ++i) {
            if (!(d.J[i] > 0.0)) throw arg_error("bdg_sw2d_curved_create: nodal Jacobian J must be positive");
            tmp[i] = 1.0 / d.J[i];
        }
        s->uploadRows(tmp.data(), s->rJ, Np);
        double** slot[4] = {&s->zx, &s->zy, &s->fcor, &s->cd};
        int next = 1;
        for (int i = 0; i < 4; ++i) {
            if (!given[i]) continue;
            *slot[i] = s->coef.p + static_cast<size_t>(next++) * s->plane();
            s->uploadRows(given[i], *slot[i], Np);
        }
    }
    s->fconst = d.coriolis_const;
    s->cdconst = d.drag_const;

    // ---- elements of curvedEls: slot table, element list, Cholesky factors with the slot index contiguous
    if (s->numCurved) {
        s->curvedSlot.alloc(ld, s->bytes, st);
        std::vector<int> slots(static_cast<size_t>(ld), -1);
        std::copy(slotOf.begin(), slotOf.end(), slots.begin());
        hipOk(hipMemcpyAsync(s->curvedSlot.p, slots.data(), slots.size() * sizeof(int), hipMemcpyHostToDevice, st), "slot upload");
        s->curvedHost = curved;
        s->curvedEls.alloc(curved.size(), s->bytes, st);
        hipOk(hipMemcpyAsync(s->curvedEls.p, curved.data(), curved.size() * sizeof(int), hipMemcpyHostToDevice, st), "curvedEls upload");
        // inverse mass matrix of every listed element from its upper Cholesky factor U (M = U^T U): W = U^-1 by back
        // substitution, Minv = W W^T (the reference solves U^T y = b, U x = y per evaluation, rhs.py:157-162)
        std::vector<double> minv(static_cast<size_t>(Np) * Np * curved.size(), 0.0);
        std::atomic<bool> badDiagonal{false};
        blitzdg::detail::parallelChunks(static_cast<int>(curved.size()), [&](int cBegin, int cEnd) {
            std::vector<double> U(static_cast<size_t>(Np) * Np), W(static_cast<size_t>(Np) * Np);
            for (int c = cBegin; c < cEnd; ++c) {
                for (int i = 0; i < Np; ++i)
                    for (int jj = 0; jj < Np; ++jj) U[static_cast<size_t>(i) * Np + jj] = d.MMChol[(static_cast<size_t>(i) * Np + jj) * K + curved[c]];
                std::fill(W.begin(), W.end(), 0.0);
                for (int jj = 0; jj < Np; ++jj) { // column jj of W: U w = e_jj, from row jj upwards
                    if (!(U[static_cast<size_t>(jj) * Np + jj] > 0.0)) { badDiagonal = true; break; }
                    W[static_cast<size_t>(jj) * Np + jj] = 1.0 / U[static_cast<size_t>(jj) * Np + jj];
                    for (int i = jj - 1; i >= 0; --i) {
                        double sum = 0.0;
                        for (int m = i + 1; m <= jj; ++m) sum += U[static_cast<size_t>(i) * Np + m] * W[static_cast<size_t>(m) * Np + jj];
                        W[static_cast<size_t>(i) * Np + jj] = -sum / U[static_cast<size_t>(i) * Np + i];
                    }
                }
                double* out = minv.data() + static_cast<size_t>(c) * Np * Np;
                for (int i = 0; i < Np; ++i)
                    for (int jj = i; jj < Np; ++jj) {
                        double sum = 0.0;
                        for (int m = jj; m < Np; ++m) sum += W[static_cast<size_t>(i) * Np + m] * W[static_cast<size_t>(jj) * Np + m];
                        out[static_cast<size_t>(i) * Np + jj] = out[static_cast<size_t>(jj) * Np + i] = sum;
                    }
            }
        }, 8);
        if (badDiagonal) throw arg_error("bdg_sw2d_curved_create: MMChol has a non-positive diagonal entry on an element of curvedEls");
        s->minvSide.alloc(minv.size(), s->bytes, st);
        hipOk(hipMemcpyAsync(s->minvSide.p, minv.data(), minv.size() * sizeof(double), hipMemcpyHostToDevice, st), "Minv upload");
        s->mmSide.alloc(static_cast<size_t>(4) * Np * curved.size(), s->bytes, st);
        hipOk(hipStreamSynchronize(st), "side upload sync");
    }

    // ---- operator image in MFMA A-tile order (layout: CurvedOps in sw2d_curved_kernel.hpp)
    s->hasFilter = d.Filter != nullptr;
    {
        int off[8];
        kt->opsOffsets(ncb, fb, off);
        std::vector<double> img(static_cast<size_t>(kt->opsTiles(ncb, fb)) * 64, 0.0);
        auto at = [&](int tile, int l) -> double& { return img[static_cast<size_t>(tile) * 64 + l]; };
        std::vector<double> Vt(static_cast<size_t>(Np) * Np);
        for (int i = 0; i < Np; ++i)
            for (int jj = 0; jj < Np; ++jj) Vt[static_cast<size_t>(i) * Np + jj] = d.V[static_cast<size_t>(jj) * Np + i];
        const std::vector<double> M = matmul(d.V, Vt.data(), Np, Np, Np);
        std::vector<double> MF, ident(static_cast<size_t>(Np) * Np, 0.0);
        if (d.Filter) MF = matmul(d.Filter, M.data(), Np, Np, Np);
        for (int l = 0; l < 64; ++l) {
            const int i = l & 15, sc = l >> 4;
            for (int rb = 0; rb < ncb; ++rb)
                for (int t = 0; t < KV; ++t) {
                    const int row = 16 * rb + i, m = 4 * t + sc;
                    if (row < Ncub && m < Np) at(off[0] + rb * KV + t, l) = d.cubV[static_cast<size_t>(row) * Np + m];
                }
            for (int r = 0; r < MT; ++r)
                for (int rb = 0; rb < ncb; ++rb)
                    for (int reg = 0; reg < 4; ++reg) {
                        const int node = 16 * r + i, cp = 16 * rb + 4 * reg + sc;
                        if (node < Np && cp < Ncub) {
                            at(off[1] + (r * ncb + rb) * 4 + reg, l) = d.cubDr[static_cast<size_t>(cp) * Np + node];
                            at(off[2] + (r * ncb + rb) * 4 + reg, l) = d.cubDs[static_cast<size_t>(cp) * Np + node];
                        }
                    }
            for (int r = 0; r < MT; ++r)
                for (int gb = 0; gb < 3 * fb; ++gb)
                    for (int reg = 0; reg < 4; ++reg) {
                        const int node = 16 * r + i, f = gb / fb, local = 16 * (gb % fb) + 4 * reg + sc;
                        if (node < Np && local < NG)
                            at(off[3] + (r * 3 * fb + gb) * 4 + reg, l) = -d.gaussInterp[static_cast<size_t>(f * NG + local) * Np + node];
                    }
            for (int r = 0; r < MT; ++r)
                for (int t = 0; t < KV; ++t) {
                    const int node = 16 * r + i, m = 4 * t + sc;
                    if (node < Np && m < Np) {
                        at(off[4] + r * KV + t, l) = M[static_cast<size_t>(node) * Np + m];
                        if (d.Filter) {
                            at(off[5] + r * KV + t, l) = MF[static_cast<size_t>(node) * Np + m];
                            at(off[6] + r * KV + t, l) = d.Filter[static_cast<size_t>(node) * Np + m];
                        }
                    }
                }
            for (int gb = 0; gb < 3 * fb; ++gb)
                for (int t = 0; t < KV; ++t) {
                    const int f = gb / fb, local = 16 * (gb % fb) + i, m = 4 * t + sc;
                    if (local < NG && m < Np) at(off[7] + gb * KV + t, l) = d.gaussInterp[static_cast<size_t>(f * NG + local) * Np + m];
                }
        }
        s->ops.alloc(img.size(), s->bytes, st);
        hipOk(hipMemcpyAsync(s->ops.p, img.data(), img.size() * sizeof(double), hipMemcpyHostToDevice, st), "ops upload");
        if (d.Filter) {
            s->filt.alloc(static_cast<size_t>(Np) * Np, s->bytes, st);
            hipOk(hipMemcpyAsync(s->filt.p, d.Filter, static_cast<size_t>(Np) * Np * sizeof(double), hipMemcpyHostToDevice, st), "filter upload");
        }
        hipOk(hipStreamSynchronize(st), "ops sync");
    }
    buildNodalTraceTables(*s, d, slotOf);
    // compulsory bytes of one RHS evaluation per element: state in, RHS out, geometry, maps, traces out and in (twice: both sides)
    // state in, RHS out, cubature geometry (4 Ncub per general element, 4 per straight one), Gauss geometry, 1/J, sources,
    // state again in the trace kernel, traces out and in (the exterior side; the interior side too when gmapM is not the identity)
    const double fracAffine = static_cast<double>(s->numAffine) / K;
    s->bytesPerElement = 8.0 * (4 * Np + 4 * Np + (1.0 - fracAffine) * 4 * Ncub + fracAffine * 4 + 3 * NG3 + Np + (d.zx ? Np : 0) +
                                (d.zy ? Np : 0) + (d.coriolis ? Np : 0) + (d.drag ? Np : 0) + 4 * Np +
                                (identityM ? 2 : 3) * 4 * NG3) +
                         4.0 * (NG3 * (identityM ? 1 : 2) + 2);
    if (s->useNT) { // the Gauss-trace planes and maps belong to the general form only
        for (Buf<double>* b : {&s->gq}) { s->bytes -= b->n * sizeof(double); b->release(); }
        for (Buf<int>* b : {&s->gmapP, &s->gmapM}) { s->bytes -= b->n * sizeof(int); b->release(); }
    }
    if (s->useNT) { // state in, update in / out, the neighbours' face nodes, node map + flags, geometry (14 numbers on straight elements), sources
        const double fa = static_cast<double>(s->numAffineNT) / K;
        const int Nfp = s->N + 1, KE = (Nfp + 3) / 4;
        s->bytesPerElement = 8.0 * (4 * Np + 4 * Np + 4 * 3 * Nfp + (1.0 - fa) * (4 * Ncub + 3 * NG3 + Np) + fa * 14 + (d.zx ? Np : 0) +
                                    (d.zy ? Np : 0) + (d.coriolis ? Np : 0) + (d.drag ? Np : 0)) +
                             4.0 * (3 * KE * 4 + 3);
    }
    return s.release();
}

} // namespace

extern "C" {

int bdg_sw2d_curved_create(const bdg_sw2d_curved_desc* desc, bdg_sw2d_curved** out) {
    return guard([&] {
        if (!desc || !out) throw arg_error("bdg_sw2d_curved_create: NULL argument");
        *out = createCurved(*desc);
    });
}

void bdg_sw2d_curved_destroy(bdg_sw2d_curved* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->commStream) (void)hipStreamSynchronize(s->commStream);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    delete s;
}

int bdg_sw2d_curved_rhs(bdg_sw2d_curved* s, const double* h, const double* hu, const double* hv, const double* hN,
                        double* r1, double* r2, double* r3, double* r4, int filter) {
    return guard([&] {
        requireCurved(s, "bdg_sw2d_curved_rhs");
        if (!h || !hu || !hv || !hN || !r1 || !r2 || !r3 || !r4) throw arg_error("bdg_sw2d_curved_rhs: NULL field pointer");
        s->use();
        const size_t pl = s->plane();
        const double* in[4] = {h, hu, hv, hN};
        double* outp[4] = {r1, r2, r3, r4};
        for (int c = 0; c < 4; ++c) s->uploadRows(in[c], s->qB.p + c * pl, s->Np); // the scratch state: qA stays untouched
        s->evaluate(bdg_dev::CMODE_RHS, filter != 0, s->qB.p, nullptr, nullptr, 0.0, 0.0, 0.0);
        for (int c = 0; c < 4; ++c) s->downloadRows(s->rhs.p + c * pl, outp[c], s->Np);
    });
}

int bdg_sw2d_curved_set_state(bdg_sw2d_curved* s, const double* h, const double* hu, const double* hv, const double* hN) {
    return guard([&] {
        requireCurved(s, "bdg_sw2d_curved_set_state");
        if (!h || !hu || !hv || !hN) throw arg_error("bdg_sw2d_curved_set_state: NULL field pointer");
        s->use();
        const double* in[4] = {h, hu, hv, hN};
        for (int c = 0; c < 4; ++c) s->uploadRows(in[c], s->qA.p + c * s->plane(), s->Np);
        hipOk(hipMemsetAsync(s->res.p, 0, s->res.n * sizeof(double), s->stream), "hipMemset");
        s->stageCount = 0;
    });
}

int bdg_sw2d_curved_get_state(bdg_sw2d_curved* s, double* h, double* hu, double* hv, double* hN) {
    return guard([&] {
        requireCurved(s, "bdg_sw2d_curved_get_state");
        if (!h || !hu || !hv || !hN) throw arg_error("bdg_sw2d_curved_get_state: NULL field pointer");
        s->use();
        double* outp[4] = {h, hu, hv, hN};
        for (int c = 0; c < 4; ++c) s->downloadRows(s->qA.p + c * s->plane(), outp[c], s->Np);
    });
}

int bdg_sw2d_curved_step_rk2(bdg_sw2d_curved* s, double dt, int num_steps, int filter) {
    return guard([&] {
        requireCurved(s, "bdg_sw2d_curved_step_rk2");
        if (num_steps < 0) throw arg_error("bdg_sw2d_curved_step_rk2: num_steps < 0");
        s->use();
        s->stepRk2(dt, num_steps, filter != 0);
    });
}

int bdg_sw2d_curved_lserk4_stages(bdg_sw2d_curved* s, double dt, int num_stages) {
    return guard([&] {
        requireCurved(s, "bdg_sw2d_curved_lserk4_stages");
        if (num_stages < 0) throw arg_error("bdg_sw2d_curved_lserk4_stages: num_stages < 0");
        s->use();
        for (int i = 0; i < num_stages; ++i) {
            const int st = static_cast<int>(s->stageCount % blitzdg::LSERK4::numStages);
            // res = a res + dt RHS(q); q += b res   (reference src/advec1d/main.cpp:92-102), in place: only own
            // elements are read from q by the stage kernel, the neighbours' traces come from gq
            s->lserkStage(blitzdg::LSERK4::rk4a[st], blitzdg::LSERK4::rk4b[st], dt);
            ++s->stageCount;
        }
    });
}

int bdg_sw2d_curved_time_rk2(bdg_sw2d_curved* s, double dt, int num_steps, int filter, float* ms_per_rhs) {
    return guard([&] {
        requireCurved(s, "bdg_sw2d_curved_time_rk2");
        if (num_steps < 1 || !ms_per_rhs) throw arg_error("bdg_sw2d_curved_time_rk2: bad argument");
        s->use();
        hipOk(hipEventRecord(s->ev0, s->stream), "hipEventRecord");
        s->stepRk2(dt, num_steps, filter != 0);
        hipOk(hipEventRecord(s->ev1, s->stream), "hipEventRecord");
        hipOk(hipEventSynchronize(s->ev1), "hipEventSynchronize");
        float ms = 0.f;
        hipOk(hipEventElapsedTime(&ms, s->ev0, s->ev1), "hipEventElapsedTime");
        *ms_per_rhs = ms / (2.0f * static_cast<float>(num_steps));
    });
}

namespace {
double* curvedBuffer(bdg_sw2d_curved* s, int which, int first, int count, const char* fn) {
    requireCurved(s, fn);
    if (which != 0 && which != 1) throw arg_error(std::string(fn) + ": which must be 0 (state) or 1 (intermediate)");
    if (first < 0 || count < 0 || static_cast<long long>(first) + count > s->K) throw arg_error(std::string(fn) + ": element range out of bounds");
    return (which == 0 ? s->qA.p : s->qB.p) + first;
}
} // namespace

int bdg_sw2d_curved_get_elements(bdg_sw2d_curved* s, int which, int first, int count, double* out) {
    return guard([&] {
        const double* dev = curvedBuffer(s, which, first, count, "bdg_sw2d_curved_get_elements");
        if (count == 0) return;
        if (!out) throw arg_error("bdg_sw2d_curved_get_elements: NULL buffer");
        s->use();
        hipOk(hipMemcpy2DAsync(out, static_cast<size_t>(count) * sizeof(double), dev, static_cast<size_t>(s->ld) * sizeof(double),
                               static_cast<size_t>(count) * sizeof(double), static_cast<size_t>(4) * s->Np, hipMemcpyDeviceToHost, s->stream), "D2H copy");
        hipOk(hipStreamSynchronize(s->stream), "download sync");
    });
}

int bdg_sw2d_curved_set_elements(bdg_sw2d_curved* s, int which, int first, int count, const double* in) {
    return guard([&] {
        double* dev = curvedBuffer(s, which, first, count, "bdg_sw2d_curved_set_elements");
        if (count == 0) return;
        if (!in) throw arg_error("bdg_sw2d_curved_set_elements: NULL buffer");
        s->use();
        hipOk(hipMemcpy2DAsync(dev, static_cast<size_t>(s->ld) * sizeof(double), in, static_cast<size_t>(count) * sizeof(double),
                               static_cast<size_t>(count) * sizeof(double), static_cast<size_t>(4) * s->Np, hipMemcpyHostToDevice, s->stream), "H2D copy");
        hipOk(hipStreamSynchronize(s->stream), "upload sync");
    });
}

int bdg_sw2d_curved_rk2_phase(bdg_sw2d_curved* s, double dt, int phase, int filter) {
    return guard([&] {
        requireCurved(s, "bdg_sw2d_curved_rk2_phase");
        if (phase != 0 && phase != 1) throw arg_error("bdg_sw2d_curved_rk2_phase: phase must be 0 (predictor) or 1 (corrector)");
        s->use();
        if (phase == 0) s->evaluate(bdg_dev::CMODE_COMBINE, filter != 0, s->qA.p, s->qA.p, s->qB.p, 1.0, 0.0, 0.5 * dt);
        else s->evaluate(bdg_dev::CMODE_COMBINE, filter != 0, s->qB.p, s->qA.p, s->qA.p, 1.0, 0.0, dt);
    });
}

int bdg_sw2d_curved_set_partition(bdg_sw2d_curved* s, int num_interior, int num_owned, const int* send_elements, int num_send) {
    return guard([&] {
        requireCurved(s, "bdg_sw2d_curved_set_partition");
        if (num_owned < 1 || num_owned > s->K || num_interior < 0 || num_interior > num_owned || num_send < 0 ||
            (num_send > 0 && !send_elements))
            throw arg_error("bdg_sw2d_curved_set_partition: bad argument");
        for (int i = 0; i < num_send; ++i)
            if (send_elements[i] < 0 || send_elements[i] >= num_owned)
                throw arg_error("bdg_sw2d_curved_set_partition: a send element is not an owned element");
        // The two-chain schedule evaluates [0, num_interior) beside the exchange: it is race-free only if no such element
        // reads a ghost column and none of them is packed for a neighbour. A plan that breaks either is refused here
        // (it would otherwise give stale ghost reads, not an error).
        for (int k = 0; k < num_interior; ++k)
            if (s->maxNeighbourHost[static_cast<size_t>(k)] >= num_owned)
                throw arg_error("bdg_sw2d_curved_set_partition: element " + std::to_string(k) +
                                " is listed as interior but has a ghost neighbour (elements >= num_owned)");
        for (int i = 0; i < num_send; ++i)
            if (send_elements[i] < num_interior)
                throw arg_error("bdg_sw2d_curved_set_partition: send element " + std::to_string(send_elements[i]) +
                                " lies in the interior range [0, num_interior)");
        if (s->comm) throw arg_error("bdg_sw2d_curved_set_partition: the communicator is already initialised");
        s->use();
        s->numOwned = num_owned;
        s->numInterior = num_interior;
        s->numSend = num_send;
        std::vector<int> inner, outer; // columns of the side buffer whose element is an interior / a partition-boundary one
        for (size_t c = 0; c < s->curvedHost.size(); ++c) {
            const int k = s->curvedHost[c];
            if (k < num_interior) inner.push_back(static_cast<int>(c));
            else if (k < num_owned) outer.push_back(static_cast<int>(c));
        }
        s->numSlotsInterior = static_cast<int>(inner.size());
        s->numSlotsBoundary = static_cast<int>(outer.size());
        s->slotsInterior.alloc(std::max<size_t>(1, inner.size()), s->bytes, s->stream);
        s->slotsBoundary.alloc(std::max<size_t>(1, outer.size()), s->bytes, s->stream);
        if (!inner.empty())
            hipOk(hipMemcpyAsync(s->slotsInterior.p, inner.data(), inner.size() * sizeof(int), hipMemcpyHostToDevice, s->stream), "slot list upload");
        if (!outer.empty())
            hipOk(hipMemcpyAsync(s->slotsBoundary.p, outer.data(), outer.size() * sizeof(int), hipMemcpyHostToDevice, s->stream), "slot list upload");
        hipOk(hipStreamSynchronize(s->stream), "slot list sync"); // (inner, outer are locals)
        s->sendEls.alloc(static_cast<size_t>(std::max(1, num_send)), s->bytes, s->stream);
        if (num_send > 0)
            hipOk(hipMemcpyAsync(s->sendEls.p, send_elements, static_cast<size_t>(num_send) * sizeof(int), hipMemcpyHostToDevice, s->stream),
                  "send list upload");
        hipOk(hipStreamSynchronize(s->stream), "send list sync");
    });
}

int bdg_sw2d_curved_comm_init(bdg_sw2d_curved* s, int rank, int world, const void* unique_id, const int* peer_ranks,
                              const int* send_start, const int* send_count, const int* recv_start, const int* recv_count,
                              int num_peers) {
    return guard([&] {
        requireCurved(s, "bdg_sw2d_curved_comm_init");
        if (!unique_id || world < 1 || rank < 0 || rank >= world || num_peers < 0 ||
            (num_peers > 0 && (!peer_ranks || !send_start || !send_count || !recv_start || !recv_count)))
            throw arg_error("bdg_sw2d_curved_comm_init: bad argument");
        if (s->comm) throw arg_error("bdg_sw2d_curved_comm_init: communicator already initialised");
        if (s->numOwned < 1) throw arg_error("bdg_sw2d_curved_comm_init: call bdg_sw2d_curved_set_partition first");
        const int ghosts = s->K - s->numOwned;
        std::vector<bdg_sw2d_curved::Peer> peers;
        for (int i = 0; i < num_peers; ++i) {
            const bdg_sw2d_curved::Peer p{peer_ranks[i], send_start[i], send_count[i], recv_start[i], recv_count[i]};
            if (p.rank < 0 || p.rank >= world || p.sendStart < 0 || p.sendCount < 0 || p.sendStart + p.sendCount > s->numSend ||
                p.recvStart < 0 || p.recvCount < 0 || p.recvStart + p.recvCount > ghosts)
                throw arg_error("bdg_sw2d_curved_comm_init: peer ranges do not fit the partition set with bdg_sw2d_curved_set_partition");
            peers.push_back(p);
        }
        s->use();
        ncclUniqueId id;
        std::memcpy(&id, unique_id, sizeof(id));
        bdg_rccl::ncclCheck(bdg_rccl::rccl().CommInitRank(&s->comm, world, id, rank), "ncclCommInitRank");
        s->commRank = rank;
        s->commWorld = world;
        s->peers = peers;
        hipOk(hipStreamCreateWithFlags(&s->commStream, hipStreamNonBlocking), "hipStreamCreate");
        // (events that only order kernels of this device's two streams: no system-scope fence, as in bdg_sw2d_comm_init)
        for (hipEvent_t* e : {&s->evA[0], &s->evA[1], &s->evB[0], &s->evB[1], &s->evEntry})
            hipOk(hipEventCreateWithFlags(e, hipEventDisableTiming | hipEventDisableSystemFence), "hipEventCreate");
        const size_t rows = static_cast<size_t>(4) * s->Np;
        s->sendBuf.alloc(std::max<size_t>(1, static_cast<size_t>(s->numSend) * rows), s->bytes, s->stream);
        s->recvBuf.alloc(std::max<size_t>(1, static_cast<size_t>(ghosts) * rows), s->bytes, s->stream);
        s->scalarBuf.alloc(2, s->bytes, s->stream);
        hipOk(hipStreamSynchronize(s->stream), "exchange buffers");
    });
}

int bdg_sw2d_curved_step_rk2_exchanged(bdg_sw2d_curved* s, double dt, int num_steps, int filter) {
    return guard([&] {
        requireCurved(s, "bdg_sw2d_curved_step_rk2_exchanged");
        if (num_steps < 0) throw arg_error("bdg_sw2d_curved_step_rk2_exchanged: num_steps < 0");
        if (!s->comm) throw arg_error("bdg_sw2d_curved_step_rk2_exchanged: no communicator (call bdg_sw2d_curved_comm_init first)");
        s->use();
        s->stepRk2Exchanged(dt, num_steps, filter != 0);
    });
}

int bdg_sw2d_curved_lserk4_stages_exchanged(bdg_sw2d_curved* s, double dt, int num_stages) {
    return guard([&] {
        requireCurved(s, "bdg_sw2d_curved_lserk4_stages_exchanged");
        if (num_stages < 0) throw arg_error("bdg_sw2d_curved_lserk4_stages_exchanged: num_stages < 0");
        if (!s->comm) throw arg_error("bdg_sw2d_curved_lserk4_stages_exchanged: no communicator (call bdg_sw2d_curved_comm_init first)");
        s->use();
        s->lserkStagesExchanged(dt, num_stages);
    });
}

int bdg_sw2d_curved_exchange(bdg_sw2d_curved* s, int intermediate) {
    return guard([&] {
        requireCurved(s, "bdg_sw2d_curved_exchange");
        s->use();
        s->exchange(intermediate ? s->qB.p : s->qA.p);
    });
}

int bdg_sw2d_curved_barrier(bdg_sw2d_curved* s) {
    return guard([&] {
        requireCurved(s, "bdg_sw2d_curved_barrier");
        if (!s->comm) throw arg_error("bdg_sw2d_curved_barrier: no communicator");
        s->use();
        hipOk(hipStreamSynchronize(s->commStream), "hipStreamSynchronize");
        hipOk(hipStreamSynchronize(s->stream), "hipStreamSynchronize");
        bdg_rccl::ncclCheck(bdg_rccl::rccl().AllReduce(s->scalarBuf.p, s->scalarBuf.p, 1, ncclDouble, ncclMax, s->comm, s->stream), "ncclAllReduce");
        hipOk(hipStreamSynchronize(s->stream), "hipStreamSynchronize");
    });
}

int bdg_sw2d_curved_synchronize(bdg_sw2d_curved* s) {
    return guard([&] {
        requireCurved(s, "bdg_sw2d_curved_synchronize");
        s->use();
        hipOk(hipStreamSynchronize(s->stream), "hipStreamSynchronize");
    });
}

size_t bdg_sw2d_curved_device_bytes(const bdg_sw2d_curved* s) { return s ? s->bytes : 0; }
double bdg_sw2d_curved_bytes_per_element(const bdg_sw2d_curved* s) { return s ? s->bytesPerElement : 0.0; }
int bdg_sw2d_curved_form(const bdg_sw2d_curved* s) { return s ? (s->useNT ? 1 : 0) : -1; }

} // extern "C"
