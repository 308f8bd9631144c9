// Instantiates the sw2d kernels for one polynomial order (-DBDG_ORDER=N).
#include "sw2d_launch.hpp"
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#ifndef BDG_ORDER
#error "compile with -DBDG_ORDER=<polynomial order>"
#endif

// A launch that records p.stopEvent (when set) through the dispatch's own completion signal: one packet instead of two on the
// queue, which is what the dependent launch on the other stream of a partitioned stage waits behind (sw2d_device.hip).
#define BDG_LAUNCH_EV(kern, grid, block, lds, stream, p, ...)                                                          \
    do {                                                                                                                \
        if ((p).stopEvent) {                                                                                            \
            hipExtLaunchKernelGGL(kern, grid, block, lds, stream, nullptr, (p).stopEvent, 0, p, ##__VA_ARGS__);         \
            if ((p).stopEventUsed) *(p).stopEventUsed = true;                                                           \
        } else {                                                                                                        \
            hipLaunchKernelGGL(kern, grid, block, lds, stream, p, ##__VA_ARGS__);                                       \
        }                                                                                                               \
    } while (0)

namespace bdg_dev {
namespace {

constexpr int kN = BDG_ORDER;
constexpr int kBlock = 256;
// The unrolled kernels run one wave per SIMD (their registers fill a SIMD's file). With one-wave workgroups
// every SIMD takes its next wave the moment its own finishes; with four-wave workgroups the CU waits for the
// slowest of the four (C3, N=4: 0.345 ms against 0.384 ms; 250 k elements: 0.102 against 0.111 ms).
constexpr int kUnrolledBlock = 64;
// Orders above this use the field-split kernels only (3*Np accumulators exceed the VGPR file).
constexpr bool kHighOrder = BDG_ORDER > 6;
// The unrolled source-term / tracer / variant-B kernels are the default only up to N = 4 (createSolver:
// kUnrolledSourcesMaxOrder); above it they are not instantiated (each costs minutes of compile time).
constexpr bool kNoUnrolledSources = BDG_ORDER > 4;
// The streamed unrolled A/B variants (BDG_SW2D_AFFINE_VARIANT = 2, 3) exist up to N = 5.
constexpr bool kNoStream = BDG_ORDER > 5;
// The matrix-core source-term / tracer / variant-B kernels exist from this order up.
constexpr bool kMfmaSources = BDG_ORDER >= 5;
// state-once kernel with sources (sw2d_mfma3src_kernel.hpp): with the tracer too at N = 5, 6, 7 (N = 7 with tracer: 501-512
// registers, 3 spilled in the combine form and still 1.9 times the two-wave kernels); three fields at N = 8, without the
// next-tile prefetch (registers) -- four fields' state tiles + operators + F' tiles would exceed 160 KB of LDS there
constexpr int kMfma3SrcFields = (BDG_ORDER >= 5 && BDG_ORDER <= 7) ? 4 : (BDG_ORDER == 8 ? 3 : 0);
// ... and at N = 8 the tracer equation as a second phase of every tile of that three-field kernel (TPHASE, sw2d_mfma3src_kernel.hpp)
constexpr bool kTracerPhase = BDG_ORDER == 8;

// Rolled kernels. FIELDS = 1 (three waves per 64 elements, one field each) exists for every
// order; FIELDS = 3 (all fields per lane) only where 3*Np accumulators fit (N <= 6).
template <int MODE, int FIELDS>
hipError_t launchRolled(const StageParams& p, hipStream_t stream) {
    if (p.kend <= p.kbegin) return hipSuccess;
    const int per = FIELDS == 3 ? 256 : 64;
    const unsigned grid = static_cast<unsigned>((p.kend - p.kbegin + per - 1) / per);
    hipLaunchKernelGGL((sw2d_stage_affine_rolled_kernel<kN, MODE, FIELDS>), dim3(grid), dim3(FIELDS == 3 ? 256 : 192), 0,
                       stream, p);
    return hipGetLastError();
}

template <int FIELDS>
hipError_t stageRolled(int mode, const StageParams& p, hipStream_t stream) {
    switch (mode) {
    case MODE_RHS: return launchRolled<MODE_RHS, FIELDS>(p, stream);
    case MODE_LSERK: return launchRolled<MODE_LSERK, FIELDS>(p, stream);
    case MODE_COMBINE: return launchRolled<MODE_COMBINE, FIELDS>(p, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t stageFieldSplit(int mode, const StageParams& p, hipStream_t stream) { return stageRolled<1>(mode, p, stream); }

template <int MODE, bool FILTER>
hipError_t launchStage(const StageParams& p, hipStream_t stream) {
    if constexpr (kHighOrder) return hipErrorNotSupported;
    else {
    if (p.kend <= p.kbegin) return hipSuccess;
    const unsigned grid = static_cast<unsigned>((p.kend - p.kbegin + kBlock - 1) / kBlock);
    hipLaunchKernelGGL((sw2d_stage_kernel<kN, MODE, FILTER>), dim3(grid), dim3(kBlock), 0, stream, p);
    return hipGetLastError();
    }
}

hipError_t stage(int mode, bool filter, const StageParams& p, hipStream_t stream) {
    switch (mode) {
    case MODE_RHS: return filter ? launchStage<MODE_RHS, true>(p, stream) : launchStage<MODE_RHS, false>(p, stream);
    case MODE_LSERK: return filter ? hipErrorInvalidValue : launchStage<MODE_LSERK, false>(p, stream);
    case MODE_COMBINE:
        return filter ? launchStage<MODE_COMBINE, true>(p, stream) : launchStage<MODE_COMBINE, false>(p, stream);
    default: return hipErrorInvalidValue;
    }
}

template <int MODE>
hipError_t launchAffine(const StageParams& p, hipStream_t stream) {
    if constexpr (kHighOrder) return stageFieldSplit(MODE, p, stream);
    else {
    if (p.kend <= p.kbegin) return hipSuccess;
    const unsigned grid = static_cast<unsigned>((p.kend - p.kbegin + kUnrolledBlock - 1) / kUnrolledBlock);
    if constexpr (MODE == MODE_COMBINE) {
        if (p.sponge != 0.0) {
            hipLaunchKernelGGL((sw2d_stage_affine_kernel<kN, MODE, 0, false, true>), dim3(grid), dim3(kUnrolledBlock), 0, stream, p, PhysParams{});
            return hipGetLastError();
        }
    }
    if (p.syncSignal) return hipErrorNotSupported; // (no SYNC instance of the unrolled kernel: see flagSyncUsable in sw2d_device.hip)
    BDG_LAUNCH_EV((sw2d_stage_affine_kernel<kN, MODE>), dim3(grid), dim3(kUnrolledBlock), 0, stream, p, PhysParams{});
    return hipGetLastError();
    }
}

// unrolled kernel with the momentum source terms (orders where it exists: N <= 6)
template <int MODE>
hipError_t launchAffineSrc(const StageParams& p, const PhysParams& ph, int tracer, hipStream_t stream) {
    if constexpr (kNoUnrolledSources) return hipErrorNotSupported;
    else {
    if (p.kend <= p.kbegin) return hipSuccess;
    const unsigned grid = static_cast<unsigned>((p.kend - p.kbegin + kUnrolledBlock - 1) / kUnrolledBlock);
    const dim3 blk(kUnrolledBlock);
    if constexpr (MODE == MODE_COMBINE) {
        if (p.sponge != 0.0) { // the instances with the momentum relaxation
            if (tracer) {
                if (ph.fmat) hipLaunchKernelGGL((sw2d_stage_affine_kernel<kN, MODE, 2, true, true>), dim3(grid), blk, 0, stream, p, ph);
                else hipLaunchKernelGGL((sw2d_stage_affine_kernel<kN, MODE, 1, true, true>), dim3(grid), blk, 0, stream, p, ph);
            } else {
                if (ph.fmat) hipLaunchKernelGGL((sw2d_stage_affine_kernel<kN, MODE, 2, false, true>), dim3(grid), blk, 0, stream, p, ph);
                else hipLaunchKernelGGL((sw2d_stage_affine_kernel<kN, MODE, 1, false, true>), dim3(grid), blk, 0, stream, p, ph);
            }
            return hipGetLastError();
        }
    }
    if (tracer) {
        if (ph.fmat) hipLaunchKernelGGL((sw2d_stage_affine_kernel<kN, MODE, 2, true>), dim3(grid), blk, 0, stream, p, ph);
        else hipLaunchKernelGGL((sw2d_stage_affine_kernel<kN, MODE, 1, true>), dim3(grid), blk, 0, stream, p, ph);
    } else {
        if (ph.fmat) hipLaunchKernelGGL((sw2d_stage_affine_kernel<kN, MODE, 2>), dim3(grid), blk, 0, stream, p, ph);
        else hipLaunchKernelGGL((sw2d_stage_affine_kernel<kN, MODE, 1>), dim3(grid), blk, 0, stream, p, ph);
    }
    return hipGetLastError();
    }
}

template <int MODE>
hipError_t launchTracer(const StageParams& p, hipStream_t stream) {
    if constexpr (kNoUnrolledSources) return hipErrorNotSupported;
    else {
    if (p.kend <= p.kbegin) return hipSuccess;
    const unsigned grid = static_cast<unsigned>((p.kend - p.kbegin + kUnrolledBlock - 1) / kUnrolledBlock);
    hipLaunchKernelGGL((sw2d_stage_tracer_kernel<kN, MODE>), dim3(grid), dim3(kUnrolledBlock), 0, stream, p);
    return hipGetLastError();
    }
}

hipError_t stageTracer(int mode, const StageParams& p, hipStream_t stream) {
    switch (mode) {
    case MODE_RHS: return launchTracer<MODE_RHS>(p, stream);
    case MODE_LSERK: return launchTracer<MODE_LSERK>(p, stream);
    case MODE_COMBINE: return launchTracer<MODE_COMBINE>(p, stream);
    default: return hipErrorInvalidValue;
    }
}

// tracer != 0: four-field state, the tracer equation in the same pass
hipError_t stageAffineSrc(int mode, const StageParams& p, const PhysParams& ph, int tracer, hipStream_t stream) {
    switch (mode) {
    case MODE_RHS: return launchAffineSrc<MODE_RHS>(p, ph, tracer, stream);
    case MODE_LSERK: return launchAffineSrc<MODE_LSERK>(p, ph, tracer, stream);
    case MODE_COMBINE: return launchAffineSrc<MODE_COMBINE>(p, ph, tracer, stream);
    default: return hipErrorInvalidValue;
    }
}

template <int MODE, int WAVES>
hipError_t launchStream(const StageParams& p, hipStream_t stream) {
    if constexpr (kNoStream) return stageFieldSplit(MODE, p, stream); // A/B variant, N <= 5 only
    else {
    if (p.kend <= p.kbegin) return hipSuccess;
    const unsigned grid = static_cast<unsigned>((p.kend - p.kbegin + kBlock - 1) / kBlock);
    hipLaunchKernelGGL((sw2d_stage_affine_stream_kernel<kN, MODE, WAVES>), dim3(grid), dim3(kBlock), 0, stream, p);
    return hipGetLastError();
    }
}

// variant: 0 = register-resident state (1 wave/SIMD), 2 / 3 = streamed state at 2 / 3 waves per SIMD
hipError_t stageAffine(int mode, int variant, const StageParams& p, hipStream_t stream) {
    if (variant == 1) return stageFieldSplit(mode, p, stream);
    if (variant == 9) { // A/B: in-wave neighbour traces through LDS (sw2d_affine_xchg_kernel.hpp), LSERK stages; other modes: variant 0
        if constexpr (!kNoStream) {
            if (mode == MODE_LSERK && 3ll * Elem<kN>::Np * p.ld * 8 <= 4294967295ll) {
                if (p.kend <= p.kbegin) return hipSuccess;
                const unsigned grid = static_cast<unsigned>((p.kend - p.kbegin + kUnrolledBlock - 1) / kUnrolledBlock);
                const unsigned magic = static_cast<unsigned>((0x100000000ull + static_cast<unsigned long long>(p.ld) - 1ull) / static_cast<unsigned long long>(p.ld));
                hipLaunchKernelGGL((sw2d_stage_affine_xchg_kernel<kN>), dim3(grid), dim3(kUnrolledBlock), 0, stream, p, magic);
                return hipGetLastError();
            }
        }
        variant = 0;
    }
    if (variant == 8) { // A/B: state-resident kernel at two waves per SIMD (sw2d_affine_lean_kernel.hpp), LSERK stages; other modes: variant 0
        if constexpr (!kNoStream) {
            if (mode == MODE_LSERK) {
                if (p.kend <= p.kbegin) return hipSuccess;
                const unsigned grid = static_cast<unsigned>((p.kend - p.kbegin + kUnrolledBlock - 1) / kUnrolledBlock);
                hipLaunchKernelGGL((sw2d_stage_affine_lean_kernel<kN>), dim3(grid), dim3(kUnrolledBlock), 0, stream, p);
                return hipGetLastError();
            }
        }
        variant = 0;
    }
    if (variant == 4) {
        if constexpr (kHighOrder) return stageFieldSplit(mode, p, stream);
        else return stageRolled<3>(mode, p, stream);
    }
    if (variant == 2) {
        switch (mode) {
        case MODE_RHS: return launchStream<MODE_RHS, 2>(p, stream);
        case MODE_LSERK: return launchStream<MODE_LSERK, 2>(p, stream);
        case MODE_COMBINE: return launchStream<MODE_COMBINE, 2>(p, stream);
        default: return hipErrorInvalidValue;
        }
    }
    if (variant == 3) {
        switch (mode) {
        case MODE_RHS: return launchStream<MODE_RHS, 3>(p, stream);
        case MODE_LSERK: return launchStream<MODE_LSERK, 3>(p, stream);
        case MODE_COMBINE: return launchStream<MODE_COMBINE, 3>(p, stream);
        default: return hipErrorInvalidValue;
        }
    }
    switch (mode) {
    case MODE_RHS: return launchAffine<MODE_RHS>(p, stream);
    case MODE_LSERK: return launchAffine<MODE_LSERK>(p, stream);
    case MODE_COMBINE: return launchAffine<MODE_COMBINE>(p, stream);
    default: return hipErrorInvalidValue;
    }
}

template <int MODE>
hipError_t launchMfma(const StageParams& p, hipStream_t stream) {
    if (p.kend <= p.kbegin) return hipSuccess;
    const size_t ldsBytes = sizeof(double) * MfmaOps<kN>::DOUBLES;
    const unsigned ntiles = static_cast<unsigned>((p.kend - p.kbegin + 15) / 16);
    const unsigned perCu = static_cast<unsigned>(std::min<size_t>(8, std::max<size_t>(1, (160u * 1024u) / ldsBytes)));
    const unsigned grid = std::min((ntiles + 3u) / 4u, 256u * perCu);
    if (p.syncSignal) { // interior launch of a partitioned stage with in-kernel dependencies: one signal per ring tile
        if constexpr (MODE == MODE_LSERK) {
            if (p.syncSignalsOut) *p.syncSignalsOut = ntiles - std::min(ntiles, static_cast<unsigned>(p.syncFirstTile));
            hipLaunchKernelGGL((sw2d_stage_mfma_kernel<kN, MODE_LSERK, false, true>), dim3(grid), dim3(256), ldsBytes, stream, p);
            return hipGetLastError();
        } else return hipErrorNotSupported;
    }
    hipLaunchKernelGGL((sw2d_stage_mfma_kernel<kN, MODE>), dim3(grid), dim3(256), ldsBytes, stream, p);
    return hipGetLastError();
}

// one LSERK4 stage of the partition-boundary elements with the halo staging folded in
hipError_t stageMfmaHalo(const StageParams& p, hipStream_t stream) {
    if (p.kend <= p.kbegin) return hipSuccess;
    const size_t ldsBytes = sizeof(double) * MfmaOps<kN>::DOUBLES;
    const unsigned ntiles = static_cast<unsigned>((p.kend - p.kbegin + 15) / 16);
    const unsigned perCu = static_cast<unsigned>(std::min<size_t>(8, std::max<size_t>(1, (160u * 1024u) / ldsBytes)));
    const unsigned grid = std::min((ntiles + 3u) / 4u, 256u * perCu);
    if (p.syncSignal) { // waits for the previous interior launch's ring tiles in the kernel, one signal per workgroup
        if (p.syncSignalsOut) *p.syncSignalsOut = grid;
        hipLaunchKernelGGL((sw2d_stage_mfma_kernel<kN, MODE_LSERK, true, true>), dim3(grid), dim3(256), ldsBytes, stream, p);
        return hipGetLastError();
    }
    // event form (2- and 4-way splits of N <= 4: the interior share runs on the unrolled kernel): one-wave workgroups, one tile each, so that
    // the launch finds room beside the interior launch (sw2d_stage_mfma_kernel, THREADS); BDG_SW2D_STRIP_WORKGROUP=256 restores the four-wave form
    static const bool fourWaves = [] { const char* e = std::getenv("BDG_SW2D_STRIP_WORKGROUP"); return e && std::atoi(e) == 256; }();
    if (!fourWaves && kN <= 4) {
        BDG_LAUNCH_EV((sw2d_stage_mfma_kernel<kN, MODE_LSERK, true, false, 64>), dim3(std::min(ntiles, 2048u * perCu)), dim3(64), ldsBytes, stream, p);
        return hipGetLastError();
    }
    BDG_LAUNCH_EV((sw2d_stage_mfma_kernel<kN, MODE_LSERK, true>), dim3(grid), dim3(256), ldsBytes, stream, p);
    return hipGetLastError();
}

// the same for the face-by-face matrix-core kernel (the boundary strip's kernel at N >= 5)
hipError_t stageMfma2Halo(const StageParams& p, hipStream_t stream) {
    if (p.kend <= p.kbegin) return hipSuccess;
    const size_t ldsBytes = sizeof(double) * MfmaOps2<kN>::DOUBLES;
    const unsigned ntiles = static_cast<unsigned>((p.kend - p.kbegin + 15) / 16);
    const unsigned perCu = static_cast<unsigned>(std::min<size_t>(BDG_MFMA2_WAVES, std::max<size_t>(1, (160u * 1024u) / ldsBytes)));
    const unsigned grid = std::min((ntiles + 3u) / 4u, 256u * perCu);
    hipLaunchKernelGGL((sw2d_stage_mfma2_kernel<kN, MODE_LSERK, 0, false, true>), dim3(grid), dim3(256), ldsBytes, stream, p, PhysParams{});
    return hipGetLastError();
}

hipError_t stageMfma(int mode, const StageParams& p, hipStream_t stream) {
    switch (mode) {
    case MODE_RHS: return launchMfma<MODE_RHS>(p, stream);
    case MODE_LSERK: return launchMfma<MODE_LSERK>(p, stream);
    case MODE_COMBINE: return launchMfma<MODE_COMBINE>(p, stream);
    default: return hipErrorInvalidValue;
    }
}

template <int MODE>
hipError_t launchMfma2(const StageParams& p, hipStream_t stream) {
    if (p.kend <= p.kbegin) return hipSuccess;
    const size_t ldsBytes = sizeof(double) * MfmaOps2<kN>::DOUBLES;
    const unsigned ntiles = static_cast<unsigned>((p.kend - p.kbegin + 15) / 16);
    const unsigned perCu = static_cast<unsigned>(std::min<size_t>(BDG_MFMA2_WAVES, std::max<size_t>(1, (160u * 1024u) / ldsBytes)));
    const unsigned grid = std::min((ntiles + 3u) / 4u, 256u * perCu);
    hipLaunchKernelGGL((sw2d_stage_mfma2_kernel<kN, MODE>), dim3(grid), dim3(256), ldsBytes, stream, p, PhysParams{});
    return hipGetLastError();
}

template <int MODE, bool NODAL = false, bool NFILT = false>
hipError_t launchMfma3(const StageParams& p, hipStream_t stream) {
    if (p.kend <= p.kbegin) return hipSuccess;
    const size_t ldsBytes = sizeof(double) * (Mfma3Lds<kN>::DOUBLES + (NFILT ? MfmaOps2<kN>::MT * MfmaOps2<kN>::KV * 64 : 0));
    auto kern = sw2d_stage_mfma3_kernel<kN, MODE, false, NODAL, NFILT>;
    if (ldsBytes > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(ldsBytes));
        if (e != hipSuccess) return e;
    }
    const unsigned ntiles = static_cast<unsigned>((p.kend - p.kbegin + 15) / 16);
    const unsigned cap = p.gridCap > 0 ? static_cast<unsigned>(p.gridCap) : 256u;
    const unsigned grid = std::min((ntiles + 3u) / 4u, cap); // one four-wave workgroup per CU, one wave per SIMD
    static const int interleave = [] { const char* e = std::getenv("BDG_SW2D_TILE_INTERLEAVE"); return e ? std::atoi(e) : 1; }();
    static const int stagger = [] { const char* e = std::getenv("BDG_SW2D_STAGGER"); return e ? std::atoi(e) : 0; }();
    StageParams pi = p;
    pi.tileInterleave = interleave;
    pi.stagger = stagger;
#ifdef BDG_PHASE_CLOCK
    // profiling build: the per-wave phase cycles of the last launch go to $BDG_PHASE_CLOCK_FILE when the process exits
    // (the buffer is pinned host memory the kernel writes directly, so nothing of HIP is needed at that point)
    static unsigned long long* clockBuf = nullptr;
    if (!clockBuf) {
        if (hipHostMalloc(&clockBuf, 1024 * 16 * sizeof(unsigned long long), hipHostMallocMapped) != hipSuccess) return hipErrorOutOfMemory;
        std::memset(clockBuf, 0, 1024 * 16 * sizeof(unsigned long long));
        static unsigned long long* dump = clockBuf;
        std::atexit([] {
            const char* name = std::getenv("BDG_PHASE_CLOCK_FILE");
            if (FILE* f = name ? std::fopen(name, "w") : nullptr) {
                for (unsigned w = 0; w < 1024; ++w) {
                    std::fprintf(f, "%u", w);
                    for (int i = 0; i < 16; ++i) std::fprintf(f, " %llu", dump[w * 16 + i]);
                    std::fprintf(f, "\n");
                }
                std::fclose(f);
            }
        });
    }
    pi.phaseClock = clockBuf;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), ldsBytes, stream, pi);
    return hipGetLastError();
#else
    if (p.syncSignal) { // interior launch of a partitioned stage with in-kernel dependencies: one signal per ring tile
        if constexpr (MODE == MODE_LSERK && !NODAL && !NFILT) {
            auto kernSync = sw2d_stage_mfma3_kernel<kN, MODE_LSERK, false, false, false, true>;
            if (ldsBytes > 64 * 1024) {
                const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernSync), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                         static_cast<int>(ldsBytes));
                if (e != hipSuccess) return e;
            }
            if (p.syncSignalsOut) *p.syncSignalsOut = ntiles - std::min(ntiles, static_cast<unsigned>(p.syncFirstTile));
            hipLaunchKernelGGL(kernSync, dim3(grid), dim3(256), ldsBytes, stream, pi);
            return hipGetLastError();
        } else return hipErrorNotSupported;
    }
    BDG_LAUNCH_EV(kern, dim3(grid), dim3(256), ldsBytes, stream, pi);
    return hipGetLastError();
#endif
}

// per-node geometry (StageParams::geo / fgeo) on the same schedule
// filter: the image in p.opsAffine carries the Filter tiles behind the plain operators
hipError_t stageMfma3Nodal(int mode, bool filter, const StageParams& p, hipStream_t stream) {
    switch (mode) {
    case MODE_RHS: return filter ? launchMfma3<MODE_RHS, true, true>(p, stream) : launchMfma3<MODE_RHS, true>(p, stream);
    case MODE_LSERK: return filter ? hipErrorInvalidValue : launchMfma3<MODE_LSERK, true>(p, stream);
    case MODE_COMBINE: return filter ? launchMfma3<MODE_COMBINE, true, true>(p, stream) : launchMfma3<MODE_COMBINE, true>(p, stream);
    default: return hipErrorInvalidValue;
    }
}

// one LSERK4 stage of the partition-boundary elements with the halo staging folded in, state-once schedule
hipError_t stageMfma3Halo(const StageParams& p, hipStream_t stream) {
    if (p.kend <= p.kbegin) return hipSuccess;
    const size_t ldsBytes = sizeof(double) * Mfma3Lds<kN>::DOUBLES;
    auto kern = sw2d_stage_mfma3_kernel<kN, MODE_LSERK, true>;
    if (ldsBytes > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(ldsBytes));
        if (e != hipSuccess) return e;
    }
    const unsigned ntiles = static_cast<unsigned>((p.kend - p.kbegin + 15) / 16);
    // Strips (up to a few thousand elements): the latency form, a tile shared by three waves (one field each). Larger
    // boundary sets: the throughput form.
    if (ntiles <= 1024u && !std::getenv("BDG_SW2D_STRIP_THROUGHPUT")) {
        const size_t stripLds = sizeof(double) * MfmaOps2<kN>::DOUBLES;
        if (p.syncSignal) { // waits for the previous interior launch's ring tiles in the kernel, one signal per workgroup
            if (p.syncSignalsOut) *p.syncSignalsOut = ntiles;
            hipLaunchKernelGGL((sw2d_strip_mfma3_kernel<kN, true>), dim3(ntiles), dim3(192), stripLds, stream, p);
            return hipGetLastError();
        }
        BDG_LAUNCH_EV((sw2d_strip_mfma3_kernel<kN>), dim3(ntiles), dim3(192), stripLds, stream, p);
        return hipGetLastError();
    }
    if (p.syncSignal) return hipErrorNotSupported; // (the throughput form has no in-kernel dependencies: the caller keeps the events)
    BDG_LAUNCH_EV(kern, dim3(std::min((ntiles + 3u) / 4u, 256u)), dim3(256), ldsBytes, stream, p);
    return hipGetLastError();
}

hipError_t stageMfma3(int mode, const StageParams& p, hipStream_t stream) {
    switch (mode) {
    case MODE_RHS: return launchMfma3<MODE_RHS>(p, stream);
    case MODE_LSERK: return launchMfma3<MODE_LSERK>(p, stream);
    case MODE_COMBINE: return launchMfma3<MODE_COMBINE>(p, stream);
    default: return hipErrorInvalidValue;
    }
}

// matrix-core kernel with the momentum sources (operator image: MfmaOps2 + MT*KV tiles of F'); tracer = 1
// launches the tracer pass instead (plain MfmaOps2 image), tracer = 2 the variant-B form (same image as the
// sources), tracer = 3 sources and tracer fused (N <= 6). Orders above the unrolled kernels' range only.
// tracer | kSrcIdentity (state-once forms 4 ... 7): the caller's image holds identity F' tiles (no filter) -- the instance that adds
// the sources pointwise instead of multiplying by them (IDF, sw2d_mfma3src_kernel.hpp)
template <int MODE>
hipError_t launchMfma2Src(const StageParams& p, const PhysParams& ph, int tracerArg, hipStream_t stream) {
    if constexpr (!kMfmaSources) return hipErrorNotSupported;
    else {
    const int tracer = tracerArg & ~kSrcIdentity;
    const bool idf = (tracerArg & kSrcIdentity) != 0 && tracer >= 4 && tracer <= 7;
    if (p.kend <= p.kbegin) return hipSuccess;
    using O = MfmaOps2<kN>;
    const size_t ldsBytes = sizeof(double) * (O::DOUBLES + (tracer == 1 ? 0 : O::MT * O::KV * 64));
    const unsigned ntiles = static_cast<unsigned>((p.kend - p.kbegin + 15) / 16);
    const unsigned waves = tracer == 1 ? 3u : static_cast<unsigned>(BDG_MFMA2_WAVES);
    const unsigned perCu = static_cast<unsigned>(std::min<size_t>(waves, std::max<size_t>(1, (160u * 1024u) / ldsBytes)));
    const unsigned grid = std::min((ntiles + 3u) / 4u, 256u * perCu);
    if (tracer == 4 || tracer == 5 || tracer == 6 || tracer == 7) { // state-once schedule with sources (and tracer); 6: variant B; 7: tracer as a second phase
        if constexpr (kMfma3SrcFields == 0) return hipErrorNotSupported;
        else {
            if (tracer == 5 && kMfma3SrcFields < 4) return hipErrorNotSupported;
            if (tracer == 7 && !kTracerPhase) return hipErrorNotSupported;
            const long long arrayBytes = static_cast<long long>((tracer == 5 || tracer == 7) ? 4 : 3) * Elem<kN>::Np * p.ld * 8;
            if (arrayBytes > 4294967295LL) return hipErrorNotSupported; // one descriptor per array
            const unsigned grid3 = std::min((ntiles + 3u) / 4u, 256u); // one four-wave workgroup per CU
            auto launch = [&](auto kern, size_t lds) -> hipError_t {
                if (lds > 64 * 1024) {
                    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                             static_cast<int>(lds));
                    if (e != hipSuccess) return e;
                }
                hipLaunchKernelGGL(kern, dim3(grid3), dim3(256), lds, stream, p, ph);
                return hipGetLastError();
            };
            if (tracer == 5) {
                if constexpr (kMfma3SrcFields >= 4)
                    return idf ? launch(sw2d_stage_mfma3src_kernel<kN, MODE, true, 1, false, true>, sizeof(double) * Mfma3SrcLds<kN, true>::DOUBLES)
                               : launch(sw2d_stage_mfma3src_kernel<kN, MODE, true>, sizeof(double) * Mfma3SrcLds<kN, true>::DOUBLES);
                else return hipErrorNotSupported;
            }
            if (tracer == 7) {
                if constexpr (kTracerPhase)
                    return idf ? launch(sw2d_stage_mfma3src_kernel<kN, MODE, false, 1, true, true>, sizeof(double) * Mfma3SrcLds<kN, true>::DOUBLES)
                               : launch(sw2d_stage_mfma3src_kernel<kN, MODE, false, 1, true>, sizeof(double) * Mfma3SrcLds<kN, true>::DOUBLES);
                else return hipErrorNotSupported;
            }
            if (tracer == 6)
                return idf ? launch(sw2d_stage_mfma3src_kernel<kN, MODE, false, 2, false, true>, sizeof(double) * Mfma3SrcLds<kN, false>::DOUBLES)
                           : launch(sw2d_stage_mfma3src_kernel<kN, MODE, false, 2>, sizeof(double) * Mfma3SrcLds<kN, false>::DOUBLES);
            return idf ? launch(sw2d_stage_mfma3src_kernel<kN, MODE, false, 1, false, true>, sizeof(double) * Mfma3SrcLds<kN, false>::DOUBLES)
                       : launch(sw2d_stage_mfma3src_kernel<kN, MODE, false>, sizeof(double) * Mfma3SrcLds<kN, false>::DOUBLES);
        }
    }
    if (tracer == 3) { // sources + tracer in one pass (MT <= 2)
        if constexpr (MfmaOps2<kN>::MT <= 2)
            hipLaunchKernelGGL((sw2d_stage_mfma2_kernel<kN, MODE, 1, true>), dim3(grid), dim3(256), ldsBytes, stream, p, ph);
        else return hipErrorNotSupported;
    } else if (tracer == 1) hipLaunchKernelGGL((sw2d_stage_mfma2_tracer_kernel<kN, MODE>), dim3(grid), dim3(256), ldsBytes, stream, p);
    else if (tracer == 2) hipLaunchKernelGGL((sw2d_stage_mfma2_kernel<kN, MODE, 2>), dim3(grid), dim3(256), ldsBytes, stream, p, ph);
    else hipLaunchKernelGGL((sw2d_stage_mfma2_kernel<kN, MODE, 1>), dim3(grid), dim3(256), ldsBytes, stream, p, ph);
    return hipGetLastError();
    }
}

hipError_t stageMfma2Src(int mode, const StageParams& p, const PhysParams& ph, int tracer, hipStream_t stream) {
    switch (mode) {
    case MODE_RHS: return launchMfma2Src<MODE_RHS>(p, ph, tracer, stream);
    case MODE_LSERK: return launchMfma2Src<MODE_LSERK>(p, ph, tracer, stream);
    case MODE_COMBINE: return launchMfma2Src<MODE_COMBINE>(p, ph, tracer, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t stageMfma2(int mode, const StageParams& p, hipStream_t stream) {
    switch (mode) {
    case MODE_RHS: return launchMfma2<MODE_RHS>(p, stream);
    case MODE_LSERK: return launchMfma2<MODE_LSERK>(p, stream);
    case MODE_COMBINE: return launchMfma2<MODE_COMBINE>(p, stream);
    default: return hipErrorInvalidValue;
    }
}

template <int MODE>
hipError_t launchVd(const StageParams& p, const VdParams& vp, hipStream_t stream) {
    if (p.kend <= p.kbegin) return hipSuccess;
    const unsigned grid = static_cast<unsigned>((p.kend - p.kbegin + 63) / 64);
    hipLaunchKernelGGL((sw2d_stage_vd_kernel<kN, MODE>), dim3(grid), dim3(64 * (vp.nf - vp.cbase)), 0, stream, p, vp);
    return hipGetLastError();
}

hipError_t stageVd(int mode, const StageParams& p, const VdParams& vp, hipStream_t stream) {
    switch (mode) {
    case MODE_RHS: return launchVd<MODE_RHS>(p, vp, stream);
    case MODE_LSERK: return launchVd<MODE_LSERK>(p, vp, stream);
    case MODE_COMBINE: return launchVd<MODE_COMBINE>(p, vp, stream);
    default: return hipErrorInvalidValue;
    }
}

// filterT: nullptr = rolled kernel with the operator image in p.opsAffine (VdOps, plain or filtered);
// otherwise the unrolled kernel (N <= 5) with plain AffineOps in p.opsAffine, and filterT[0] != nullptr ...
template <int MODE>
hipError_t launchVb(const StageParams& p, const VbParams& vp, double* partials, double* lam, int unrolled,
                    const double* filterT, hipStream_t stream) {
    if (p.kend <= p.kbegin) return hipSuccess;
    // 2, 6: vp.lam already holds the speed of this state (2: accumulated by the launch that wrote it, unrolled kernel;
    // 6: reduced over all ranks of a partitioned run beforehand, rolled kernel)
    if (unrolled != 2 && unrolled != 6) {
        const unsigned nblocks = static_cast<unsigned>((p.kend - p.kbegin + 255) / 256);
        hipLaunchKernelGGL((sw2d_vb_speed_kernel<kN>), dim3(nblocks), dim3(256), 0, stream, p, vp, partials);
        hipLaunchKernelGGL((sw2d_vb_speed_reduce_kernel<kN>), dim3(1), dim3(256), 0, stream, partials, static_cast<int>(nblocks), lam);
    }
    if (unrolled == 4) return hipGetLastError(); // speed pass only: the stage pass is the matrix-core kernel
    if constexpr (!kNoUnrolledSources) {
        if (unrolled && unrolled != 6) {
            const unsigned grid = static_cast<unsigned>((p.kend - p.kbegin + kUnrolledBlock - 1) / kUnrolledBlock);
            if (filterT)
                hipLaunchKernelGGL((sw2d_stage_vb_unrolled_kernel<kN, MODE, true>), dim3(grid), dim3(kUnrolledBlock), 0, stream, p, vp, filterT);
            else
                hipLaunchKernelGGL((sw2d_stage_vb_unrolled_kernel<kN, MODE, false>), dim3(grid), dim3(kUnrolledBlock), 0, stream, p, vp, filterT);
            return hipGetLastError();
        }
    }
    const unsigned grid = static_cast<unsigned>((p.kend - p.kbegin + 63) / 64);
    hipLaunchKernelGGL((sw2d_stage_vb_kernel<kN, MODE>), dim3(grid), dim3(192), 0, stream, p, vp);
    return hipGetLastError();
}

hipError_t stageVb(int mode, const StageParams& p, const VbParams& vp, double* partials, double* lam, int unrolled,
                   const double* filterT, hipStream_t stream) {
    switch (mode) {
    case MODE_RHS: return launchVb<MODE_RHS>(p, vp, partials, lam, unrolled, filterT, stream);
    case MODE_LSERK: return launchVb<MODE_LSERK>(p, vp, partials, lam, unrolled, filterT, stream);
    case MODE_COMBINE: return launchVb<MODE_COMBINE>(p, vp, partials, lam, unrolled, filterT, stream);
    default: return hipErrorInvalidValue;
    }
}

// ---- variants B / C / D on per-node geometry (sw2d_vn_kernel.hpp)
template <int MODE, int PHYS>
hipError_t launchVn(const StageParams& p, const VdParams& vd, const VbParams& vb, const double* ops, const double* filt, double* raw,
                    double* partials, double* lam, bool speedPass, hipStream_t stream) {
    if (p.kend <= p.kbegin) return hipSuccess;
    if (PHYS == 2 && speedPass) {
        const unsigned nblocks = static_cast<unsigned>((p.kend - p.kbegin + 255) / 256);
        hipLaunchKernelGGL((sw2d_vn_speed_kernel<kN>), dim3(nblocks), dim3(256), 0, stream, p, vb, partials);
        hipLaunchKernelGGL((sw2d_vb_speed_reduce_kernel<kN>), dim3(1), dim3(256), 0, stream, partials, static_cast<int>(nblocks), lam);
    }
    const int nf = PHYS == 2 ? 3 : vd.nf;
    const unsigned grid = static_cast<unsigned>((p.kend - p.kbegin + 63) / 64);
    if (!filt) {
        hipLaunchKernelGGL((sw2d_stage_vn_kernel<kN, MODE, PHYS>), dim3(grid), dim3(64 * nf), 0, stream, p, vd, vb, ops);
    } else {
        StageParams pr = p;
        pr.rhs = raw;
        hipLaunchKernelGGL((sw2d_stage_vn_kernel<kN, MODE_RHS, PHYS>), dim3(grid), dim3(64 * nf), 0, stream, pr, vd, vb, ops);
        hipLaunchKernelGGL((sw2d_filter_rows_kernel<kN, MODE, PHYS>), dim3(grid), dim3(64 * nf), 0, stream, p, raw, filt, vb.sponge, 0);
    }
    return hipGetLastError();
}

template <int PHYS>
hipError_t stageVnPhys(int mode, const StageParams& p, const VdParams& vd, const VbParams& vb, const double* ops, const double* filt,
                       double* raw, double* partials, double* lam, bool speedPass, hipStream_t stream) {
    switch (mode) {
    case MODE_RHS: return launchVn<MODE_RHS, PHYS>(p, vd, vb, ops, filt, raw, partials, lam, speedPass, stream);
    case MODE_LSERK: return launchVn<MODE_LSERK, PHYS>(p, vd, vb, ops, filt, raw, partials, lam, speedPass, stream);
    case MODE_COMBINE: return launchVn<MODE_COMBINE, PHYS>(p, vd, vb, ops, filt, raw, partials, lam, speedPass, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t stageVn(int mode, int phys, const StageParams& p, const VdParams& vd, const VbParams& vb, const double* ops,
                   const double* filt, double* raw, double* partials, double* lam, bool speedPass, hipStream_t stream) {
    return phys == 2 ? stageVnPhys<2>(mode, p, vd, vb, ops, filt, raw, partials, lam, speedPass, stream)
                     : stageVnPhys<1>(mode, p, vd, vb, ops, filt, raw, partials, lam, speedPass, stream);
}

hipError_t dt(const double* q, const double* fscale, const double* H, long long ld, int K, double g, double* partials,
              hipStream_t stream) {
    const unsigned grid = static_cast<unsigned>((K + kBlock - 1) / kBlock);
    hipLaunchKernelGGL((sw2d_dt_kernel<kN>), dim3(grid), dim3(kBlock), 0, stream, q, fscale, H, ld, K, g, partials);
    return hipGetLastError();
}

hipError_t output(const double* q, const double* H, const double* M, double* out, long long ld, int K, int which,
                  hipStream_t stream) {
    const unsigned grid = static_cast<unsigned>((K + kBlock - 1) / kBlock);
    hipLaunchKernelGGL((sw2d_output_kernel<kN>), dim3(grid), dim3(kBlock), 0, stream, q, H, M, out, ld, K, which);
    return hipGetLastError();
}

int fmaskOf(int f, int n) { return Elem<kN>::fmask(f, n); }

} // namespace

#define BDG_CAT2(a, b) a##b
#define BDG_CAT(a, b) BDG_CAT2(a, b)
// A host function (not a namespace-scope constant, which hipcc would also emit for
// the device and then fail to resolve the host function pointers in).
const KernelTable* BDG_CAT(kernel_table_order, BDG_ORDER)() {
    static const KernelTable table = {kN, Elem<kN>::Np, Elem<kN>::Nfp, kHighOrder ? 0 : Elem<kN>::LDS_DOUBLES, &stage,
                                      AffineOps<kN>::DOUBLES, &stageAffine, MfmaOps<kN>::DOUBLES, MfmaOps<kN>::MT,
                                      MfmaOps<kN>::KV, MfmaOps<kN>::KS, &stageMfma, &stageMfmaHalo, &stageMfma2Halo, MfmaOps2<kN>::DOUBLES, MfmaOps2<kN>::KF,
                                      &stageMfma2, &stageMfma3, &stageMfma3Halo, &stageMfma3Nodal, &stageMfma2Src, kMfma3SrcFields, kTracerPhase ? 1 : 0, VdOps<kN>::DOUBLES, &stageVd, &stageAffineSrc, &stageTracer, &stageVb, VnOps<kN>::DOUBLES, &stageVn, &dt, &output,
                                      &fmaskOf};
    return &table;
}

} // namespace bdg_dev
