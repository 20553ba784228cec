// store_bench.hip -- global store issue rate per CU by store width (gfx950); sizes the scatter
// kernel's copy-out.  Each wave writes runs of `lanes` elements at
// pseudo-random 2-byte-aligned positions of a large buffer (like bucket cursors).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int BYTES>
__global__ __launch_bounds__(768) void store_kernel(unsigned char *buf, size_t span, int iters, int lanes, int skew, int groups, int lanes_log2, int cur_shift, uint32_t adv_mask)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    // each wave owns 64 "cursors" spread over its private region; round-robin like bucket runs
    const size_t region = span / ((size_t)gridDim.x * (blockDim.x >> 6));
    unsigned char *base = buf + wave * region;
    uint32_t s = (uint32_t)wave * 2654435761u;
    uint32_t adv = 0;
    for (int it = 0; it < iters; ++it, adv += lanes * BYTES + skew) {
#pragma unroll 8
        for (int c = 0; c < 64; ++c) {
            // cursor c advances by lanes*BYTES each visit (+ a 2-byte skew so runs are unaligned)
            // `groups` runs per instruction: lane group q = lane / lanes writes cursor c*groups+q.
            // All index arithmetic is 32-bit shifts/masks (no divisions) so the loop is store-bound.
            const int q = lane >> lanes_log2, l = lane & (lanes - 1);
            unsigned char *p = base + (size_t)((uint32_t)(c * groups + q) << cur_shift) + (adv & adv_mask);
            if (q < groups) {
                const int lane = l;
                if (BYTES == 2) *(uint16_t *)(p + lane * 2) = (uint16_t)s;
                else if (BYTES == 4) *(uint32_t *)(p + lane * 4) = s;
                else if (BYTES == 8) *(uint2 *)(p + lane * 8) = make_uint2(s, s);
                else *(uint4 *)(p + lane * 16) = make_uint4(s, s, s, s);
            }
            s = s * 1664525u + 1013904223u;
        }
    }
}

template <int BYTES>
void run(unsigned char *buf, size_t span_full, int lanes, int skew, int groups = 1, size_t span = 0)
{
    const int iters = 40;
    if (span == 0) span = span_full;
    int lanes_log2 = 0;
    while ((1 << lanes_log2) < lanes) ++lanes_log2;
    const size_t region = span / (512 * 12);
    int cur_shift = 0;
    while (((size_t)2 << cur_shift) * 64 * groups <= region) ++cur_shift;   // cursor spacing: power of two
    const uint32_t adv_mask = ((1u << cur_shift) - 1u) >> 1 & ~1u;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(store_kernel<BYTES>, dim3(512), dim3(768), 0, 0, buf, span, 2, lanes, skew, groups, lanes_log2, cur_shift, adv_mask);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(store_kernel<BYTES>, dim3(512), dim3(768), 0, 0, buf, span, iters, lanes, skew, groups, lanes_log2, cur_shift, adv_mask);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double instr_per_cu = (double)iters * 64 * 12 * 2;  // 2 blocks of 12 waves per CU
    const double bytes = (double)iters * 64 * 12 * 512 * lanes * BYTES * groups;
    printf("span %5zu MiB: store %2d B/lane, %2d lanes x%d runs/instr, skew %d: %7.1f ns per store-instr per CU (%.1f clk @2.2GHz), %.2f TB/s\n", span >> 20, BYTES, lanes, groups, skew,
           ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.2, bytes / ms / 1e9);
}

int main()
{
    const size_t span = (size_t)8 << 30;
    unsigned char *buf;
    CHECK(hipMalloc(&buf, span));
    for (size_t sp : {(size_t)8 << 30, (size_t)2 << 30, (size_t)512 << 20, (size_t)96 << 20}) {
        run<2>(buf, span, 32, 2, 1, sp);
        run<2>(buf, span, 64, 0, 1, sp);
        run<16>(buf, span, 8, 0, 1, sp);
        run<16>(buf, span, 8, 0, 8, sp);
        run<16>(buf, span, 64, 0, 1, sp);
    }
    return 0;
}
