// Developer probe: accuracy of v_rcp_f64 and of div_counts() against IEEE division on gfx950.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o tools/rcp_probe.bin tools/rcp_probe.hip && tools/rcp_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include "../kpal_amd/csrc/vec_kernels.hpp"

__global__ void probe(uint64_t n, double *out)
{
    double worst_rcp = 0.0, worst_div = 0.0;
    uint64_t state = 0x9E3779B97F4A7C15ULL * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    for (uint64_t i = 0; i < n; ++i) {
        state = state * 6364136223846793005ULL + 1442695040888963407ULL;
        const uint32_t x = (uint32_t)(state >> 33), y = (uint32_t)(state >> 2) & 0x7FFFFFFFu;
        const double den = (i & 1) ? ((double)x + 1.0) * ((double)y + 1.0) : (double)x + (double)y + 1.0;
        const double num = fabs((double)x - (double)y);
        const double r = __builtin_amdgcn_rcp(den);
        worst_rcp = fmax(worst_rcp, fabs(r * den - 1.0));
        const double exact = num / den;
        if (exact != 0.0) worst_div = fmax(worst_div, fabs(kpal::div_counts(num, den) - exact) / exact);
    }
    out[2 * (blockIdx.x * blockDim.x + threadIdx.x)] = worst_rcp;
    out[2 * (blockIdx.x * blockDim.x + threadIdx.x) + 1] = worst_div;
}

int main()
{
    const int threads = 256 * 1024;
    double *d;
    hipMalloc(&d, threads * 2 * sizeof(double));
    probe<<<1024, 256>>>(4096, d);
    double *h = new double[threads * 2];
    hipMemcpy(h, d, threads * 2 * sizeof(double), hipMemcpyDeviceToHost);
    double wr = 0, wd = 0;
    for (int i = 0; i < threads; ++i) {
        wr = fmax(wr, h[2 * i]);
        wd = fmax(wd, h[2 * i + 1]);
    }
    printf("v_rcp_f64: max |r*den - 1| = %.3g (2^%.1f); div_counts vs IEEE: max relative difference %.3g (%.2f ulp)\n", wr, log2(wr), wd,
           wd / 2.220446049250313e-16);
    return 0;
}
