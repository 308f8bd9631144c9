// Which offsets does the range check of a raw buffer load (stride 0, offen) look at on gfx950?
//   hipcc -O3 --offload-arch=gfx950 buffer_range_check.hip -o buffer_range_check && ./buffer_range_check
// data[i] = i + 1 (doubles); descriptor range R bytes; a load that is out of range returns 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const double* data, unsigned records, const unsigned* voff, const unsigned* soff, double* out, int n) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(data), 0, records, 0x00020000);
    for (int i = 0; i < n; ++i) {
        const unsigned so = __builtin_amdgcn_readfirstlane(soff[i]);
        out[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff[i], so, 0));
    }
}

int main() {
    const int N = 4096; // 32 KiB of data
    std::vector<double> h(N);
    for (int i = 0; i < N; ++i) h[i] = i + 1;
    double *d, *out;
    unsigned *vo, *so;
    if (hipMalloc(&d, N * 8) != hipSuccess) return 1;
    hipMemcpy(d, h.data(), N * 8, hipMemcpyHostToDevice);
    const unsigned R = 8192; // descriptor range: the first 1024 doubles
    struct Case { unsigned v, s; const char* what; };
    const Case cases[] = {
        {80, 0, "voffset < R, soffset 0"},
        {80, 8192, "voffset < R, soffset = R (sum inside the allocation)"},
        {80, 16384, "voffset < R, soffset = 2 R"},
        {8000, 800, "voffset < R, soffset < R, sum >= R"},
        {8192, 0, "voffset = R, soffset 0"},
        {16384, 0, "voffset = 2 R, soffset 0"},
        {0xfffffff8u, 0, "voffset 0xfffffff8, soffset 0"},
        {0xfffffff8u, 808, "voffset 0xfffffff8, soffset 808 (32-bit wrapped sum = 800)"},
        {0xfffffff8u, 16392, "voffset 0xfffffff8, soffset 16392 (wrapped sum = 16384)"},
        {8184, 0, "voffset = R - 8 (last record)"},
        {8188, 0, "voffset = R - 4 (8-byte load straddles the end)"},
    };
    const int n = sizeof(cases) / sizeof(cases[0]);
    std::vector<unsigned> hv(n), hs(n);
    for (int i = 0; i < n; ++i) { hv[i] = cases[i].v; hs[i] = cases[i].s; }
    hipMalloc(&vo, n * 4); hipMalloc(&so, n * 4); hipMalloc(&out, n * 8);
    hipMemcpy(vo, hv.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(so, hs.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(1), 0, 0, d, R, vo, so, out, n);
    std::vector<double> ho(n);
    hipMemcpy(ho.data(), out, n * 8, hipMemcpyDeviceToHost);
    std::printf("descriptor range R = %u bytes; data[i] = i + 1; 0 = out of range\n", R);
    for (int i = 0; i < n; ++i)
        std::printf("  %-62s -> %8.1f   (data at voffset+soffset: %.1f)\n", cases[i].what, ho[i],
                    (static_cast<unsigned long long>(cases[i].v) + cases[i].s) / 8 < N ? h[(static_cast<unsigned long long>(cases[i].v) + cases[i].s) / 8] : -1.0);
    return 0;
}
