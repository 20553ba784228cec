// store_probe2.hip -- cost of scattered ALIGNED pieces by piece size and store width, with and
// without a concurrent streaming read (sizes the flush of the record-based scatter).
// Every wave owns NCUR cursors 8 KiB apart; one store instruction writes G pieces of PIECE bytes at
// G different cursors, each piece by PIECE/W lanes storing W bytes.  Cursors advance by PIECE.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int W, bool READ>
__global__ __launch_bounds__(512) void probe(unsigned char *buf, const uint4 *__restrict__ src, uint4 *sink, int piece, int iters)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lanes_per_piece = piece / W;
    const int G = 64 / lanes_per_piece;          // pieces per instruction
    const int q = lane / lanes_per_piece, l = lane % lanes_per_piece;
    constexpr int NCUR = 64;
    uint32_t s = (uint32_t)wave * 2654435761u + 12345u;
    uint4 acc = make_uint4(0, 0, 0, 0);
    const uint4 *rp = src + wave * (size_t)iters * NCUR * 64 + lane;   // this wave's private input stream (READ)
    for (int it = 0; it < iters; ++it) {
        const uint32_t adv = ((uint32_t)it * (uint32_t)piece) & 8191u;
        for (int c = 0; c < NCUR; c += G) {
            if (READ) {
                // bytes read = bytes written: one 1 KiB load per 1 KiB of pieces
                if (((c / G) * G * piece) % 1024 == 0) {
                    const uint4 v = *rp;
                    rp += 64;
                    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
                }
            }
            unsigned char *p = buf + (wave * (size_t)NCUR + (size_t)(c + q)) * 8192 + adv + l * W;
            if (W == 16) *(uint4 *)p = make_uint4(s, s, s, s);
            else if (W == 8) *(uint2 *)p = make_uint2(s, s);
            else if (W == 4) *(uint32_t *)p = s;
            else *(uint16_t *)p = (uint16_t)s;
            s = s * 1664525u + 1013904223u;
        }
    }
    if (READ && acc.x == 0x12345678u) sink[wave] = acc;
}

template <int W, bool READ>
static void run(unsigned char *buf, const uint4 *src, uint4 *sink, int piece, int blocks)
{
    const int waves = blocks * 8, iters = 64;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((probe<W, READ>), dim3(blocks), dim3(512), 0, 0, buf, src, sink, piece, 4);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((probe<W, READ>), dim3(blocks), dim3(512), 0, 0, buf, src, sink, piece, iters);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double pieces = (double)iters * 64 * waves;
    const int cus = blocks >= 256 ? 256 : blocks;
    printf("piece %4d B by %2d-B stores (%2d pieces/instr)%s blocks %d: %6.1f clk per piece per CU, %5.2f B/clk/CU written, %.2f TB/s written\n",
           piece, W, 64 / (piece / W), READ ? " + equal streaming read" : "                       ", blocks,
           ms * 1e6 * 2.2 / (pieces / cus), piece / (ms * 1e6 * 2.2 / (pieces / cus)), pieces * piece / ms / 1e9);
}

int main()
{
    const int blocks = 512;
    const size_t cap = (size_t)blocks * 8 * 64 * 8192;   // 2 GiB
    unsigned char *buf; uint4 *src, *sink;
    CHECK(hipMalloc(&buf, cap));
    CHECK(hipMalloc(&src, (size_t)blocks * 8 * 64 * 64 * 1024 + 4096));   // up to 1 KiB per (wave, iter, cursor)
    CHECK(hipMalloc(&sink, (size_t)blocks * 8 * 16));
    CHECK(hipMemset(buf, 0, cap));
    for (int b : {256, 512}) {
        run<16, false>(buf, src, sink, 32, b);
        run<16, false>(buf, src, sink, 64, b);
        run<16, false>(buf, src, sink, 128, b);
        run<16, false>(buf, src, sink, 256, b);
        run<16, false>(buf, src, sink, 1024, b);
        run<8, false>(buf, src, sink, 64, b);
        run<8, false>(buf, src, sink, 128, b);
        run<4, false>(buf, src, sink, 32, b);
        run<4, false>(buf, src, sink, 64, b);
        run<4, false>(buf, src, sink, 128, b);
        run<4, false>(buf, src, sink, 256, b);
        run<2, false>(buf, src, sink, 64, b);
        run<2, false>(buf, src, sink, 128, b);
    }
    run<16, true>(buf, src, sink, 64, 512);
    run<16, true>(buf, src, sink, 128, 512);
    run<4, true>(buf, src, sink, 64, 512);
    run<4, true>(buf, src, sink, 128, 512);
    run<16, true>(buf, src, sink, 1024, 512);
    return 0;
}
