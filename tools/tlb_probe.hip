// tlb_probe.hip -- what limits scattered 128-byte line writes (the scatter kernels' store path)?
// Every wave owns NCUR cursors; cursor c of wave w lives at  base + (w * NCUR + c) * stride  and
// advances by 128 B per visit inside a window of `win` bytes (wrapping).  One store instruction
// writes `groups` whole aligned lines (8 lanes x 16 B each) at `groups` different cursors.
//   footprint = waves * NCUR * win        (vs the 256 MiB Infinity Cache)
//   pages     = distinct stride-sized regions per CU (vs the TLB reach)
// Usage: tlb_probe  (prints a table)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(512) void probe(unsigned char *buf, size_t stride, uint32_t win, int ncur, int iters, int groups_log2)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int groups = 1 << groups_log2;
    const int q = lane >> 3, l = lane & 7;
    uint32_t s = (uint32_t)wave * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
        const uint32_t adv = ((uint32_t)it * 128u) & (win - 1u);
        for (int c = 0; c < ncur; c += groups) {
            if (q < groups) {
                unsigned char *p = buf + (wave * (size_t)ncur + (size_t)(c + q)) * stride + adv;
                *(uint4 *)(p + l * 16) = make_uint4(s, s, s, s);
            }
            s = s * 1664525u + 1013904223u;
        }
    }
}

static void run(unsigned char *buf, size_t cap, size_t stride, uint32_t win, int ncur, int groups_log2, int blocks)
{
    const int waves = blocks * 8;
    const size_t need = (size_t)waves * ncur * stride;
    if (need > cap) { printf("skip (needs %zu MiB)\n", need >> 20); return; }
    const int iters = 64;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, buf, stride, win, ncur, 4, groups_log2);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, buf, stride, win, ncur, iters, groups_log2);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double lines = (double)iters * ncur * waves;
    const int cus = blocks >= 512 ? 256 : (blocks + 1) / 2;
    printf("blocks %4d stride %8zu win %6u ncur %4d lines/instr %d: span %6zu MiB footprint %6zu MiB | %6.1f clk per line per CU @2.2GHz, %.2f TB/s, %.1f G lines/s\n",
           blocks, stride, win, ncur, 1 << groups_log2, need >> 20, ((size_t)waves * ncur * win) >> 20,
           ms * 1e6 * 2.2 / (lines / cus), lines * 128 / ms / 1e9, lines / ms / 1e6);
}

int main()
{
    const size_t cap = (size_t)40 << 30;
    unsigned char *buf;
    CHECK(hipMalloc(&buf, cap));
    CHECK(hipMemset(buf, 0, cap));
    // A: footprint fixed (~2 GiB written region), vary the address spread (stride) -> TLB reach
    for (size_t stride : {(size_t)8192, (size_t)16384, (size_t)65536, (size_t)262144, (size_t)1 << 20})
        run(buf, cap, stride, 8192, 64, 0, 512);
    // B: stride fixed 8 KiB (the chunk size), vary cursors per wave (open set per CU) at win 8 KiB
    for (int ncur : {16, 64, 256, 1024}) run(buf, cap, 8192, 8192, ncur, 0, 512);
    // C: small windows: footprint inside the Infinity Cache vs outside, same spread
    for (uint32_t win : {128u, 512u, 2048u, 8192u}) run(buf, cap, 8192, win, 64, 0, 512);
    // D: lines per instruction
    for (int g : {0, 1, 2, 3}) run(buf, cap, 8192, 8192, 64, g, 512);
    // E: fewer CUs active (is the limit per CU or chip-wide?)
    for (int blocks : {64, 128, 256, 512}) run(buf, cap, 8192, 8192, 64, 0, blocks);
    // F: like the real kernel: 512 cursors per 8-wave block = 64 per wave, stride 8 KiB, but blocks' ranges 16 MiB apart is what
    //    stride 8 KiB x 64 x 8 = 4 MiB per block already gives; bigger chunk stride = more pages per CU
    for (size_t stride : {(size_t)1024, (size_t)2048, (size_t)4096}) run(buf, cap, stride, (uint32_t)stride, 64, 0, 512);
    return 0;
}
