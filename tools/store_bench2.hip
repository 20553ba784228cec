// store_bench2.hip -- cost of writing a tile's bucket runs back-to-back (tile-major layout):
// each wave writes 64 runs of `lanes` 2-byte keys that are adjacent in memory, tile after tile,
// versus the same runs scattered to 64 far-apart cursors (bucket-major layout).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <bool ADJACENT>
__global__ __launch_bounds__(512) void k(unsigned char *buf, size_t region, int iters, int lanes, uint32_t cursor_stride)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    unsigned char *base = buf + wave * region;
    uint32_t s = (uint32_t)wave * 2654435761u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
        for (int c = 0; c < 64; ++c) {
            unsigned char *p = ADJACENT ? base + ((size_t)it * 64 + c) * (lanes * 2)
                                        : base + (size_t)c * cursor_stride + (size_t)it * (lanes * 2);
            if (lane < lanes) *(uint16_t *)(p + lane * 2) = (uint16_t)s;
            s = s * 1664525u + 1013904223u;
        }
    }
}

template <bool ADJACENT>
void run(unsigned char *buf, size_t span, int lanes)
{
    const int iters = 40, blocks = 512, waves = blocks * 8;
    const size_t region = (span / waves) & ~(size_t)127;
    const uint32_t cursor_stride = (uint32_t)((region / 64) & ~(size_t)127);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<ADJACENT>, dim3(blocks), dim3(512), 0, 0, buf, region, 2, lanes, cursor_stride);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k<ADJACENT>, dim3(blocks), dim3(512), 0, 0, buf, region, iters, lanes, cursor_stride);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double runs_per_cu = (double)iters * 64 * 8 * 2;
    printf("%s runs of %3d B (span %zu MiB): %6.1f clk per run per CU @2.2GHz, %.2f TB/s\n", ADJACENT ? "adjacent " : "scattered", lanes * 2, span >> 20,
           ms * 1e6 / runs_per_cu * 2.2, (double)iters * 64 * waves * lanes * 2 / ms / 1e9);
}

int main()
{
    const size_t span = (size_t)2 << 30;
    unsigned char *buf;
    CHECK(hipMalloc(&buf, span));
    for (int lanes : {44, 64}) {
        run<false>(buf, span, lanes);
        run<true>(buf, span, lanes);
    }
    return 0;
}
