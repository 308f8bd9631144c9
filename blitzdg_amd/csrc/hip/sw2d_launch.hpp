// Per-order launch table: each polynomial order is compiled in its own
// translation unit (sw2d_order.hip with -DBDG_ORDER=N) so the fully unrolled
// kernels build in parallel; the solver picks the table at run time.
#pragma once
#include "sw2d_affine_kernel.hpp"
#include "sw2d_affine_lean_kernel.hpp"
#include "sw2d_vd_kernel.hpp"
#include "sw2d_tracer_kernel.hpp"
#include "sw2d_vb_kernel.hpp"
#include "sw2d_vn_kernel.hpp"
#include "sw2d_mfma_kernel.hpp"
#include "sw2d_mfma3_kernel.hpp"
#include "sw2d_mfma3src_kernel.hpp"
#include "sw2d_affine_xchg_kernel.hpp"
#include "sw2d_kernels.hpp"

namespace bdg_dev {

constexpr int kSrcIdentity = 16; // stageMfma2Src's tracer argument, state-once forms: "the image's F' tiles are the identity" (no filter)

struct KernelTable {
    int order, Np, Nfp, ldsDoubles;
    // mode: StageMode; launches one fused RHS(+stage update) pass over K elements.
    hipError_t (*stage)(int mode, bool filter, const StageParams& p, hipStream_t stream);
    // affine-geometry fast path (no FILTER template: the filter is folded into the operators)
    int affineOpsDoubles;
    hipError_t (*stageAffine)(int mode, int variant, const StageParams& p, hipStream_t stream);
    // matrix-core path (v_mfma_f64_16x16x4_f64), straight-sided elements; needs ldsBytes of dynamic LDS
    int mfmaOpsDoubles, mfmaMT, mfmaKV, mfmaKS;
    hipError_t (*stageMfma)(int mode, const StageParams& p, hipStream_t stream);
    // MODE_LSERK on partition-boundary elements with pack / unpack folded in (StageParams::halo*)
    hipError_t (*stageMfmaHalo)(const StageParams& p, hipStream_t stream);
    hipError_t (*stageMfma2Halo)(const StageParams& p, hipStream_t stream); // the same on the face-by-face kernel (N >= 5)
    int mfma2OpsDoubles, mfma2KF; // face-by-face schedule (lift tiles padded per face)
    hipError_t (*stageMfma2)(int mode, const StageParams& p, hipStream_t stream);
    // state-once schedule, one wave per SIMD with software-pipelined loads (same MfmaOps2 image; three fields, no
    // sources, no halo staging): sw2d_mfma3_kernel.hpp
    hipError_t (*stageMfma3)(int mode, const StageParams& p, hipStream_t stream);
    hipError_t (*stageMfma3Halo)(const StageParams& p, hipStream_t stream); // MODE_LSERK with the halo staging folded in
    // per-node geometry (geo / fgeo planes); filter: plain operators + MT*KV Filter tiles in the image
    hipError_t (*stageMfma3Nodal)(int mode, bool filter, const StageParams& p, hipStream_t stream);
    // N >= 5: the same kernel with momentum sources (image = MfmaOps2 + MT*KV tiles of F'); tracer = 1: the
    // tracer-equation pass (plain MfmaOps2 image); tracer = 2: variant B (image as for the sources); tracer = 3: sources +
    // tracer in one pass (MT <= 2); tracer = 4 / 5: sources without / with the tracer on the state-once schedule
    // (sw2d_mfma3src_kernel.hpp; hipErrorNotSupported where its LDS tiles or registers do not fit: see mfma3SrcFields)
    hipError_t (*stageMfma2Src)(int mode, const StageParams& p, const PhysParams& ph, int tracer, hipStream_t stream);
    int mfma3SrcFields; // 0: no state-once kernel with sources at this order; else the number of fields it takes (up to)
    int mfma3TracerPhase; // 1: tracer = 7 exists -- the three-field state-once kernel with the tracer equation as a second phase of every tile (N = 8)
    // variant D (tracer + sources), straight-sided elements, nf = 3 or 4 waves per 64 elements
    int vdOpsDoubles;
    hipError_t (*stageVd)(int mode, const StageParams& p, const VdParams& vp, hipStream_t stream);
    // unrolled affine kernel + momentum sources (3 fields; N <= 6), the fast path of variants C/D
    hipError_t (*stageAffineSrc)(int mode, const StageParams& p, const PhysParams& ph, int tracer, hipStream_t stream);
    // tracer equation alone (field 3 of a four-field state), unrolled; N <= 6
    hipError_t (*stageTracer)(int mode, const StageParams& p, hipStream_t stream);
    // variant B (depth, star states, open boundary, global Lax-Friedrichs speed, sources): speed pass over
    // [kbegin, kend) into partials (one double per 256 elements) and *lam, then the fused stage pass
    // unrolled != 0 (N <= 5): the unrolled stage kernel with plain AffineOps in p.opsAffine and the filter
    // (filterT, [m][i] = F[i][m], or nullptr) applied at the end; 2: the same without the speed pass (vp.lam is
    // current); 4: speed pass only; 0: speed pass + rolled kernel with a VdOps image
    hipError_t (*stageVb)(int mode, const StageParams& p, const VbParams& vp, double* partials, double* lam,
                          int unrolled, const double* filterT, hipStream_t stream);
    // variants B (phys = 2) and C / D (phys = 1) on per-node geometry (sw2d_vn_kernel.hpp): ops = VnOps image; filt != nullptr:
    // the filtered RHS in two passes (unfiltered rows to raw, nf planes; then Filter and the update); speedPass: variant B's
    // global speed of this state is reduced into *lam first (else vb.lam is current)
    int vnOpsDoubles;
    hipError_t (*stageVn)(int mode, int phys, const StageParams& p, const VdParams& vd, const VbParams& vb, const double* ops,
                          const double* filt, double* raw, double* partials, double* lam, bool speedPass, hipStream_t stream);
    // per-block partial maxima (2 doubles per block of 256 elements)
    hipError_t (*dt)(const double* q, const double* fscale, const double* H, long long ld, int K, double g,
                     double* partials, hipStream_t stream);
    // output step: which = 0 eta (h - H or h), 1 u, 2 v; M = (Np, Np) lattice interpolation or nullptr
    hipError_t (*output)(const double* q, const double* H, const double* M, double* out, long long ld, int K, int which,
                         hipStream_t stream);
    // volume node index of face node (f, n), for validating the caller's vmapM
    int (*fmask)(int f, int n);
};

const KernelTable* kernel_table(int order); // nullptr if the order is not compiled in

} // namespace bdg_dev
