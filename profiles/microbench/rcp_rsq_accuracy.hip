// Accuracy of v_rcp_f64 / v_rsq_f64 on gfx950 and of the Newton refinements behind them (fast_rcp / fast_sqrt of
// sw2d_affine_kernel.hpp): largest relative error, in units of 2^-52, over 2^22 arguments spread over 1e-3 .. 1e6.
//   hipcc -O3 --offload-arch=gfx950 rcp_rsq_accuracy.hip -o rcp_rsq_accuracy && ./rcp_rsq_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void probe(const double* x, double* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    out[i] = r;
    r = fma(fma(-v, r, 1.0), r, r);
    out[n + i] = r;
    r = fma(fma(-v, r, 1.0), r, r);
    out[2 * n + i] = r;
    const double y = __builtin_amdgcn_rsq(v);
    double g = v * y, h = 0.5 * y;
    out[3 * n + i] = g;
    const double rr = fma(-h, g, 0.5);
    g = fma(g, rr, g);
    h = fma(h, rr, h);
    out[4 * n + i] = g;
    g = fma(fma(-g, g, v), h, g);
    out[5 * n + i] = g;
    g = fma(fma(-g, g, v), h, g);
    out[6 * n + i] = g;
}

int main() {
    const int n = 1 << 22;
    std::vector<double> x(n);
    for (int i = 0; i < n; ++i) x[i] = std::exp(std::log(1e-3) + (std::log(1e6) - std::log(1e-3)) * (i + 0.37) / n);
    double *dx, *dout;
    if (hipMalloc(&dx, n * 8) != hipSuccess || hipMalloc(&dout, 7ull * n * 8) != hipSuccess) return 1;
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    std::vector<double> o(7ull * n);
    hipMemcpy(o.data(), dout, 7ull * n * 8, hipMemcpyDeviceToHost);
    const char* name[7] = {"v_rcp_f64", "rcp + 1 Newton step", "rcp + 2 Newton steps (fast_rcp)", "x * v_rsq_f64", "sqrt: coupled step",
                           "sqrt: + 1 correction", "sqrt: + 2 corrections (fast_sqrt)"};
    for (int k = 0; k < 7; ++k) {
        long double worst = 0;
        for (int i = 0; i < n; ++i) {
            const long double exact = k < 3 ? 1.0L / x[i] : sqrtl(static_cast<long double>(x[i]));
            const long double e = fabsl((o[static_cast<size_t>(k) * n + i] - exact) / exact);
            if (e > worst) worst = e;
        }
        std::printf("%-36s max relative error %.3Le = %.2Lf x 2^-52\n", name[k], worst, worst / 2.220446049250313e-16L);
    }
    return 0;
}
