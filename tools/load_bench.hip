// load_bench.hip -- cost of gathering short runs (2 B/lane) from scattered locations, gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int BYTES, int UNROLL>
__global__ __launch_bounds__(1024) void load_kernel(const unsigned char *buf, size_t span, int iters, int lanes, size_t tile_stride, uint32_t *out)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint32_t acc = 0;
    // run r of this wave lives in "tile" r at a pseudo-random offset (like bucket runs inside sorted tiles)
    uint32_t s = (uint32_t)wave * 2654435761u + 99u;
    for (int it = 0; it < iters; it += UNROLL) {
        uint32_t v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            s = s * 1664525u + 1013904223u;
            const uint32_t tile = ((uint32_t)(it + u) * 4099u + (uint32_t)wave * 7u) & 65535u;   // 64 Ki tiles x 48 KiB = 3 GiB
            const uint32_t off = (s >> 8) & 0x7FFEu;
            const unsigned char *p = buf + tile * tile_stride + off;
            v[u] = 0;
            if (lane < lanes) {
                if (BYTES == 2) v[u] = *(const uint16_t *)(p + lane * 2);
                else v[u] = *(const uint32_t *)(p + lane * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int BYTES, int UNROLL>
void run(const unsigned char *buf, size_t span, int lanes, uint32_t *out)
{
    const int iters = 4096;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((load_kernel<BYTES, UNROLL>), dim3(256), dim3(1024), 0, 0, buf, span, 64, lanes, (size_t)49152, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((load_kernel<BYTES, UNROLL>), dim3(256), dim3(1024), 0, 0, buf, span, iters, lanes, (size_t)49152, out);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double instr_per_cu = (double)iters * 16;
    printf("load %d B/lane x %2d lanes, %d in flight/wave: %6.1f ns per run per CU (%.1f clk @2.2GHz), %.2f TB/s useful\n", BYTES, lanes, UNROLL,
           ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.2, (double)iters * 16 * 256 * lanes * BYTES / ms / 1e9);
}

int main()
{
    const size_t span = (size_t)4 << 30;
    unsigned char *buf;
    uint32_t *out;
    CHECK(hipMalloc(&buf, span));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(buf, 1, span));
    run<2, 4>(buf, span, 44, out);
    run<2, 8>(buf, span, 44, out);
    run<2, 16>(buf, span, 44, out);
    run<2, 8>(buf, span, 64, out);
    run<4, 8>(buf, span, 44, out);
    run<4, 8>(buf, span, 64, out);
    return 0;
}
