// Does a wave's fp64 vector work run beside its fp64 matrix instructions on gfx950, or do the two share one pipe?
//   hipcc -O3 --offload-arch=gfx950 mfma_valu_overlap.hip -o mfma_valu_overlap && ./mfma_valu_overlap
// One wave per SIMD (256 workgroups of 256 threads), each running REPS rounds of
//   mode 0: 8 independent v_mfma_f64_16x16x4_f64
//   mode 1: 8 x VPER independent v_fma_f64
//   mode 2: the two interleaved (1 matrix instruction, VPER vector instructions, ...)
//   mode 3: as 2 with 32-bit integer vector work (v_mad_u32_u24) in place of v_fma_f64
//   mode 4: the integer work alone
// If time(2) ~ max(time(0), time(1)) the pipes overlap; if ~ time(0) + time(1) they do not.
// pair<...>: TWO waves per SIMD (512-thread workgroups), waves 0-3 run the matrix instructions, waves 4-7 the vector
// ones (what = 1: both, 2: matrix waves only, 3: vector waves only): do two waves of one SIMD overlap the two kinds?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int REPS = 2000;

template <int MODE, int VPER>
__global__ __launch_bounds__(256) void probe(double* out, double seed) {
    v4d acc[8];
    double a = seed + threadIdx.x, b = seed * 0.5, v[8];
    unsigned u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] = v4d{0, 0, 0, 0}; v[i] = seed + i; u[i] = threadIdx.x + i; }
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0 || MODE == 2 || MODE == 3)
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (MODE == 1 || MODE == 2) {
#pragma unroll
                for (int j = 0; j < VPER; ++j)
                    asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(v[(i + j) & 7]) : "v"(a), "v"(b));
            }
            if (MODE == 3 || MODE == 4) {
#pragma unroll
                for (int j = 0; j < VPER; ++j)
                    asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(u[(i + j) & 7]) : "v"(u[(i + j + 1) & 7]), "v"(u[(i + j + 2) & 7]));
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i] + u[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VPER, bool INTEGER>
__global__ __launch_bounds__(512) void pair(double* out, double seed, int what) {
    v4d acc[8];
    double a = seed + threadIdx.x, b = seed * 0.5, v[8];
    unsigned u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] = v4d{0, 0, 0, 0}; v[i] = seed + i; u[i] = threadIdx.x + i; }
    const bool matrixWave = threadIdx.x < 256;
    if (matrixWave ? (what == 3) : (what == 2)) return;
    if (matrixWave) {
        for (int r = 0; r < REPS; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    } else {
        for (int r = 0; r < REPS; ++r)
#pragma unroll
            for (int i = 0; i < 8 * VPER; ++i) {
                if (INTEGER) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(u[i & 7]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
                else asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(v[i & 7]) : "v"(a), "v"(b));
            }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i] + u[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int VPER, bool INTEGER>
float runPair(double* out, int what) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((pair<VPER, INTEGER>), dim3(256), dim3(512), 0, 0, out, 1e-9, what);
    hipEventRecord(e0);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((pair<VPER, INTEGER>), dim3(256), dim3(512), 0, 0, out, 1e-9, what);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

template <int VPER>
void sweepPair(double* out) {
    std::printf("{\"two_waves_per_simd\": true, \"vector_per_matrix\": %d, \"ms_matrix_waves\": %.4f, \"ms_fma64_waves\": %.4f, \"ms_both_fma64\": %.4f, "
                "\"ms_int_waves\": %.4f, \"ms_both_int\": %.4f}\n", VPER, runPair<VPER, false>(out, 2), runPair<VPER, false>(out, 3),
                runPair<VPER, false>(out, 1), runPair<VPER, true>(out, 3), runPair<VPER, true>(out, 1));
}

template <int MODE, int VPER>
float run(double* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<MODE, VPER>), dim3(256), dim3(256), 0, 0, out, 1e-9);
    hipEventRecord(e0);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((probe<MODE, VPER>), dim3(256), dim3(256), 0, 0, out, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

template <int VPER>
void sweep(double* out) {
    const float m = run<0, VPER>(out), v = run<1, VPER>(out), both = run<2, VPER>(out), bi = run<3, VPER>(out), i = run<4, VPER>(out);
    // cycles per round of 8 matrix instructions at the clock the matrix-only run implies (64 cycles each)
    std::printf("{\"vector_per_matrix\": %d, \"ms_mfma\": %.4f, \"ms_fma64\": %.4f, \"ms_mfma_fma64\": %.4f, \"ms_int\": %.4f, \"ms_mfma_int\": %.4f, "
                "\"implied_GHz\": %.3f}\n", VPER, m, v, both, i, bi, REPS * 8 * 64 / (m * 1e6));
}

int main() {
    double* out;
    if (hipMalloc(&out, 256 * 512 * sizeof(double)) != hipSuccess) { std::fprintf(stderr, "no device\n"); return 1; }
    for (int w = 0; w < 200; ++w) hipLaunchKernelGGL((probe<0, 4>), dim3(256), dim3(256), 0, 0, out, 1e-9); // clock ramp
    hipDeviceSynchronize();
    sweep<4>(out);
    sweep<8>(out);
    sweep<12>(out);
    sweepPair<8>(out);
    sweepPair<12>(out);
    hipFree(out);
    return 0;
}
