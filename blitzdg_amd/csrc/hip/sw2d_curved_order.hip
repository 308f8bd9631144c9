// Instantiates the curved / over-integrated sw2d kernels for one polynomial order (-DBDG_ORDER=N).
#include "sw2d_curved_kernel.hpp"
#include "sw2d_curved_nt_kernel.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#ifndef BDG_ORDER
#error "compile with -DBDG_ORDER=<polynomial order>"
#endif

namespace bdg_dev {
namespace {

constexpr int kN = BDG_ORDER;
using O = CurvedOps<kN>;
// Dynamic LDS a workgroup of the stage kernel may claim for the operator image; beyond it the tiles are read
// from global memory (they stay in L1 / L2: every wave reads the same image).
constexpr int kLdsBudgetBytes = 150 * 1024;
// measured per evaluation (profiles/r02_curved_timings.json), 2 waves per SIMD against 1: N=3 0.46 / 0.61 ms, N=4 0.54 /
// 0.70, N=5 0.84 / 0.74, N=6 0.88 / 0.73, N=8 1.86 / 1.76
constexpr int kDefaultWaves = BDG_ORDER <= 4 ? 2 : 1;

int opsTiles(int ncb, int fb) { return O::tiles(ncb, fb); }
int stageTiles(int ncb, int fb) { return O::tiles(ncb, fb); }
void opsOffsets(int ncb, int fb, int* off) {
    off[0] = O::offVc(ncb, fb); off[1] = O::offDrT(ncb, fb); off[2] = O::offDsT(ncb, fb); off[3] = O::offIT(ncb, fb);
    off[4] = O::offM(ncb, fb); off[5] = O::offMF(ncb, fb); off[6] = O::offF(ncb, fb); off[7] = O::offGI(ncb, fb);
}

unsigned gridFor(int K, int wgPerCu) {
    const unsigned ntiles = (static_cast<unsigned>(K) + 15u) / 16u, wgs = (ntiles + 3u) / 4u;
    return std::max(1u, std::min(wgs, 256u * static_cast<unsigned>(wgPerCu) * 2u));
}

hipError_t gauss(const CurvedParams& p, hipStream_t stream) {
    if (p.K < 1) return hipSuccess;
    const size_t lds = static_cast<size_t>(3) * p.fb * O::KV * 64 * sizeof(double);
    hipLaunchKernelGGL((sw2d_curved_gauss_kernel<kN>), dim3(gridFor(p.K, 4)), dim3(256), lds, stream, p);
    return hipGetLastError();
}

template <int MODE, bool FILTER, int FB, int WAVES, bool MAPM = false>
hipError_t launchStageFb(const CurvedParams& p, hipStream_t stream);

// Register budget: BDG_SW2D_CURVED_WAVES = 1 | 2 waves per SIMD (A/B switch; default below).
int curvedWaves() {
    static const int w = [] {
        const char* e = std::getenv("BDG_SW2D_CURVED_WAVES");
        return (e && e[0] == '2') ? 2 : ((e && e[0] == '1') ? 1 : kDefaultWaves);
    }();
    return w;
}

template <int MODE, bool FILTER>
hipError_t launchStage(const CurvedParams& p, hipStream_t stream) {
    const bool two = curvedWaves() == 2;
    if (p.gmapM) { // rewired interior map (rare): one register budget only
        if (p.fb == 1) return launchStageFb<MODE, FILTER, 1, 1, true>(p, stream);
        if (p.fb == 2) return launchStageFb<MODE, FILTER, 2, 1, true>(p, stream);
        return hipErrorInvalidValue;
    }
    if (p.fb == 1) return two ? launchStageFb<MODE, FILTER, 1, 2>(p, stream) : launchStageFb<MODE, FILTER, 1, 1>(p, stream);
    if (p.fb == 2) return two ? launchStageFb<MODE, FILTER, 2, 2>(p, stream) : launchStageFb<MODE, FILTER, 2, 1>(p, stream);
    return hipErrorInvalidValue; // more than 32 Gauss points per face: refused at creation
}

template <int MODE, bool FILTER, int FB, int WAVES, bool MAPM>
hipError_t launchStageFb(const CurvedParams& p, hipStream_t stream) {
    if (p.K < 1) return hipSuccess;
    const size_t wref = static_cast<size_t>(16 * p.ncb) * sizeof(double); // reference weights of straight elements, behind the image
    const size_t lds = static_cast<size_t>(O::tiles(p.ncb, p.fb)) * 64 * sizeof(double) + wref;
    if (lds <= static_cast<size_t>(kLdsBudgetBytes)) {
        auto kern = sw2d_curved_stage_kernel<kN, MODE, FILTER, true, FB, WAVES, MAPM>;
        if (lds > 64 * 1024) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
            if (e != hipSuccess) return e;
        }
        const int wgPerCu = std::max(1, static_cast<int>(160 * 1024 / std::max<size_t>(lds, 1)));
        hipLaunchKernelGGL(kern, dim3(gridFor(p.K, std::min(wgPerCu, 2))), dim3(256), lds, stream, p);
    } else {
        hipLaunchKernelGGL((sw2d_curved_stage_kernel<kN, MODE, FILTER, false, FB, WAVES, MAPM>), dim3(gridFor(p.K, 2)), dim3(256), wref, stream, p);
    }
    return hipGetLastError();
}

hipError_t stage(int mode, bool filter, const CurvedParams& p, hipStream_t stream) {
    switch (mode) {
    case CMODE_RHS: return filter ? launchStage<CMODE_RHS, true>(p, stream) : launchStage<CMODE_RHS, false>(p, stream);
    case CMODE_LSERK: return filter ? launchStage<CMODE_LSERK, true>(p, stream) : launchStage<CMODE_LSERK, false>(p, stream);
    case CMODE_COMBINE: return filter ? launchStage<CMODE_COMBINE, true>(p, stream) : launchStage<CMODE_COMBINE, false>(p, stream);
    default: return hipErrorInvalidValue;
    }
}

template <int MODE, bool FILTER>
hipError_t launchFixup(const CurvedParams& p, hipStream_t stream) {
    if (p.numCurved < 1) return hipSuccess;
    constexpr int P = O::Np <= 16 ? 16 : (O::Np <= 32 ? 32 : 64), EPW = 64 / P; // elements per wave (sw2d_curved_fixup_kernel)
    hipLaunchKernelGGL((sw2d_curved_fixup_kernel<kN, MODE, FILTER>), dim3((p.numCurved + EPW - 1) / EPW), dim3(64), 0, stream, p);
    return hipGetLastError();
}

hipError_t fixup(int mode, bool filter, const CurvedParams& p, hipStream_t stream) {
    switch (mode) {
    case CMODE_RHS: return filter ? launchFixup<CMODE_RHS, true>(p, stream) : launchFixup<CMODE_RHS, false>(p, stream);
    case CMODE_LSERK: return filter ? launchFixup<CMODE_LSERK, true>(p, stream) : launchFixup<CMODE_LSERK, false>(p, stream);
    case CMODE_COMBINE: return filter ? launchFixup<CMODE_COMBINE, true>(p, stream) : launchFixup<CMODE_COMBINE, false>(p, stream);
    default: return hipErrorInvalidValue;
    }
}

// ---- nodal-trace form (sw2d_curved_nt_kernel.hpp)
using ONT = CurvedOpsNT<kN>;
constexpr size_t kLdsLimitBytes = 160 * 1024;

int ntTiles(int ncb, int fb) { return ONT::tiles(ncb, fb); }
void ntOffsets(int ncb, int fb, int* off) {
    off[0] = ONT::VCH; off[1] = ONT::SCH; off[2] = ONT::KE; off[3] = ONT::offSurf(ncb, 0); off[4] = ONT::offMass(ncb, fb);
}
size_t ntLdsResident(int ncb, int fb) { return (static_cast<size_t>(ONT::tiles(ncb, fb)) * 64 + 16 * ncb + 16 * fb + 6 * ONT::KE) * sizeof(double); }
size_t ntLdsStreamed(int ncb, int fb, bool filter) {
    return (static_cast<size_t>(2 * ONT::VCH + 3 * fb * ONT::SCH + (filter ? 2 : 1) * ONT::MT * ONT::KV) * 64 + 16 * ncb + 16 * fb + 6 * ONT::KE) * sizeof(double);
}
// Which forms are compiled for this order (compile time: every form is 24 kernels): the resident-image form up to order 6
// (with the builders' default rules it fits there), the streamed form from order 5 on (BDG_SW2D_CURVED_STREAM=1 selects it
// where both exist: A/B switch). An image that does not fit the form(s) of its order keeps the first kernels.
constexpr bool kNtResident = BDG_ORDER <= 6, kNtStream = BDG_ORDER >= 5;
bool ntStreamed(int ncb, int fb) {
    static const bool force = [] { const char* e = std::getenv("BDG_SW2D_CURVED_STREAM"); return e && e[0] == '1'; }();
    if (!kNtStream) return false;
    return !kNtResident || force || ntLdsResident(ncb, fb) > static_cast<size_t>(kLdsBudgetBytes);
}
bool ntFits(int ncb, int fb, bool filter) {
    if (fb < 1 || fb > 2) return false;
    return ntStreamed(ncb, fb) ? ntLdsStreamed(ncb, fb, filter) <= kLdsLimitBytes
                               : ntLdsResident(ncb, fb) <= static_cast<size_t>(kLdsBudgetBytes);
}

template <int MODE, bool FILTER, int STREAM, int FB, int RL>
hipError_t launchNT(const CurvedParams& p, hipStream_t stream, size_t lds) {
    constexpr int WAVES = kDefaultWaves;
    auto kern = sw2d_curved_nt_kernel<kN, MODE, FILTER, STREAM, FB, WAVES, RL>;
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(lds));
        if (e != hipSuccess) return e;
    }
    // resident workgroups: WAVES per SIMD = WAVES workgroups of four waves per CU, as far as LDS allows
    const int wgPerCu = std::max(1, std::min<int>(WAVES, static_cast<int>(kLdsLimitBytes / std::max<size_t>(lds, 1))));
    const unsigned ntiles = (static_cast<unsigned>(p.K - p.kbegin) + 15u) / 16u, wgs = (ntiles + 3u) / 4u;
    static const unsigned rounds = [] { const char* e = std::getenv("BDG_SW2D_CURVED_ROUNDS"); return e ? static_cast<unsigned>(std::atoi(e)) : 0u; }();
    // one resident round of workgroups (each wave loops over its tiles): a second round re-stages the operator image and balances
    // no better (BDG_SW2D_CURVED_ROUNDS=2..4 measured 1-4 % slower at N = 3, 6, 8)
    const unsigned slots = 256u * static_cast<unsigned>(wgPerCu) * (rounds ? rounds : 1u);
    const unsigned reserve = std::min(static_cast<unsigned>(std::max(p.gridReserve, 0)), slots / 2u);
    const unsigned grid = std::max(1u, std::min(wgs, slots - reserve));
#ifdef BDG_PHASE_CLOCK
    // profiling build: the per-wave phase cycles of the last launch (workgroups 0..1023) go to $BDG_PHASE_CLOCK_FILE when the
    // process exits (pinned host memory the kernel writes directly, so nothing of HIP is needed at that point)
    static unsigned long long* clockBuf = nullptr;
    if (!clockBuf) {
        if (hipHostMalloc(&clockBuf, 4096 * 12 * sizeof(unsigned long long), hipHostMallocMapped) != hipSuccess) return hipErrorOutOfMemory;
        std::memset(clockBuf, 0, 4096 * 12 * sizeof(unsigned long long));
        static unsigned long long* dump = clockBuf;
        std::atexit([] {
            const char* name = std::getenv("BDG_PHASE_CLOCK_FILE");
            if (FILE* f = name ? std::fopen(name, "w") : nullptr) {
                for (unsigned w = 0; w < 4096; ++w) {
                    std::fprintf(f, "%u", w);
                    for (int i = 0; i < 12; ++i) std::fprintf(f, " %llu", dump[w * 12 + i]);
                    std::fprintf(f, "\n");
                }
                std::fclose(f);
            }
        });
    }
    CurvedParams pc = p;
    pc.phaseClock = clockBuf;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, pc);
    return hipGetLastError();
#else
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
    return hipGetLastError();
#endif
}

// Compiled shapes of the face blocks: the builders' default rule NGauss = 2 (N + 1) exactly -- FB blocks per face, RL live 4-row
// steps in the last one -- and the two general shapes (every step of one / two blocks) for any other rule.
constexpr int kNgDef = 2 * (BDG_ORDER + 1), kFbDef = (kNgDef + 15) / 16, kRlDef = (kNgDef - 16 * (kFbDef - 1) + 3) / 4;

template <int MODE, bool FILTER, int STREAM>
hipError_t launchShape(const CurvedParams& p, hipStream_t stream, size_t lds) {
    const int rl = (p.ng - 16 * (p.fb - 1) + 3) / 4;
    if (p.fb == kFbDef && rl == kRlDef) return launchNT<MODE, FILTER, STREAM, kFbDef, kRlDef>(p, stream, lds);
    if (p.fb == 1) return launchNT<MODE, FILTER, STREAM, 1, 4>(p, stream, lds);
    return launchNT<MODE, FILTER, STREAM, 2, 4>(p, stream, lds);
}

template <int MODE, bool FILTER>
hipError_t launchStageNT(const CurvedParams& p, hipStream_t stream) {
    if (p.K <= p.kbegin) return hipSuccess;
    const bool streamed = ntStreamed(p.ncb, p.fb);
    const size_t lds = streamed ? ntLdsStreamed(p.ncb, p.fb, FILTER) : ntLdsResident(p.ncb, p.fb);
    if (lds > kLdsLimitBytes || p.fb < 1 || p.fb > 2) return hipErrorInvalidValue; // (ntFits was asked at creation)
    if (streamed) {
        if constexpr (kNtStream) return launchShape<MODE, FILTER, 1>(p, stream, lds);
        return hipErrorInvalidValue;
    }
    if constexpr (kNtResident) return launchShape<MODE, FILTER, 0>(p, stream, lds);
    return hipErrorInvalidValue;
}

hipError_t stageNT(int mode, bool filter, const CurvedParams& p, hipStream_t stream) {
    switch (mode) {
    case CMODE_RHS: return filter ? launchStageNT<CMODE_RHS, true>(p, stream) : launchStageNT<CMODE_RHS, false>(p, stream);
    case CMODE_LSERK: return filter ? launchStageNT<CMODE_LSERK, true>(p, stream) : launchStageNT<CMODE_LSERK, false>(p, stream);
    case CMODE_COMBINE: return filter ? launchStageNT<CMODE_COMBINE, true>(p, stream) : launchStageNT<CMODE_COMBINE, false>(p, stream);
    default: return hipErrorInvalidValue;
    }
}

const CurvedKernelTable kTable = {kN, O::Np, O::KV, O::MT, opsTiles, stageTiles, opsOffsets, gauss, stage, fixup,
                                  ntTiles, ntOffsets, ntFits, stageNT};

} // namespace

#define BDG_CAT2(a, b) a##b
#define BDG_CAT(a, b) BDG_CAT2(a, b)
const CurvedKernelTable* BDG_CAT(curved_kernel_table_order, BDG_ORDER)() { return &kTable; }

} // namespace bdg_dev
