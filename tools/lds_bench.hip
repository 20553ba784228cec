// lds_bench.hip -- microbenchmark of LDS atomic / scatter primitives on gfx950, used to size the
// partition kernels.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/lds_bench tools/lds_bench.hip && /tmp/lds_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            printf("%s: %s\n", #x, hipGetErrorString(e));                             \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

enum Op { ADD32 = 0, ADD32_RTN, ADD64, WRITE16, WRITE32, READ32, ADD32_REP32, ADD32_RTN_WRITE16, ADD32_NOCONF, ADD32_PAIRS, ADD64_NOCONF, READ32_NOCONF };

__device__ __forceinline__ uint32_t rng(uint32_t &s)
{
    s = s * 1664525u + 1013904223u;
    return s >> 8;
}

// (the random bin of every operation comes from eight per-thread offsets advanced by one add + one and: a generator of five VALU
// instructions per operation -- what this file used until round 5 -- bounds the loop at ~6 clk per wave-instruction per CU by itself,
// four SIMDs issuing one VALU instruction per 4 clk each, and hid every LDS rate below that)
template <int OP>
__global__ __launch_bounds__(1024) void bench_kernel(int iters, uint32_t nbins, uint32_t *out)
{
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < 36864; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    uint32_t s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    const uint32_t mask = nbins - 1;
    constexpr bool kNoConf = OP == ADD32_NOCONF || OP == ADD64_NOCONF || OP == READ32_NOCONF;
    uint32_t a[8], step[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        a[u] = rng(s) & mask;
        step[u] = (rng(s) | 1u) & mask;
        if (kNoConf) {                       // lane = bank (64 dwords; 32 qwords): the steps leave the low bits alone
            const uint32_t low = OP == ADD64_NOCONF ? 31u : 63u;
            a[u] = (a[u] & ~low) | (threadIdx.x & low);
            step[u] = (step[u] & ~low) | (low + 1u);
        }
        if (OP == ADD32_PAIRS) {             // neighbouring lanes share an address
            a[u] = __shfl(a[u], (int)(threadIdx.x & 62u));
            step[u] = __shfl(step[u], (int)(threadIdx.x & 62u));
        }
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a[u] = (a[u] + step[u]) & mask;
            const uint32_t b = a[u], r = b;
            if (OP == ADD32 || OP == ADD32_NOCONF || OP == ADD32_PAIRS) atomicAdd(&lds[b], 1u);
            else if (OP == ADD32_RTN) acc += atomicAdd(&lds[b], 1u);
            else if (OP == ADD64) atomicAdd((unsigned long long *)&lds[2 * (b & (mask >> 1))], 1ULL);
            else if (OP == ADD64_NOCONF) atomicAdd((unsigned long long *)&lds[2 * (b & (mask >> 1))], 1ULL);
            else if (OP == WRITE16) ((uint16_t *)lds)[r & 0x7FFF] = (uint16_t)r;
            else if (OP == WRITE32) lds[b] = r;
            else if (OP == READ32 || OP == READ32_NOCONF) acc += lds[b];
            else if (OP == ADD32_REP32) atomicAdd(&lds[(b & 511) * 32 + (threadIdx.x & 31)], 1u);
            else if (OP == ADD32_RTN_WRITE16) {
                const uint32_t slot = atomicAdd(&lds[b & 511], 1u);
                ((uint16_t *)(lds + 512))[(b & 511) * 64 + ((slot + 2 * b) & 63)] = (uint16_t)r;
            }
        }
    }
    __syncthreads();
    if (acc == 0xFFFFFFFFu || lds[threadIdx.x] == 0xFFFFFFFFu) out[0] = acc;
}

template <int OP>
void run(const char *name, int block, int blocks_per_cu, uint32_t nbins, uint32_t *dout)
{
    const int iters = 2000;
    const int grid = 256 * blocks_per_cu;
    const size_t lds_bytes = 147456 / blocks_per_cu > 147456 ? 147456 : 147456 / blocks_per_cu;
    CHECK(hipFuncSetAttribute((const void *)bench_kernel<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(bench_kernel<OP>, dim3(grid), dim3(block), lds_bytes, 0, 10, nbins, dout);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(bench_kernel<OP>, dim3(grid), dim3(block), lds_bytes, 0, iters, nbins, dout);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double wave_instr_per_cu = (double)iters * 8 * (block / 64) * blocks_per_cu;
    const double ns_per = ms * 1e6 / wave_instr_per_cu;
    printf("%-22s block=%4d x%d/CU bins=%6u : %7.2f ns per wave-instr per CU (%.1f clk @2.4GHz), %.2f Glanes/s chip\n", name,
           block, blocks_per_cu, nbins, ns_per, ns_per * 2.4, 64.0 * 256 / ns_per);
}

// in-kernel clock: delta s_memtime (shader cycles) / delta s_memrealtime (100 MHz)
__global__ void clock_kernel(unsigned long long *out, int iters)
{
    __shared__ uint32_t l[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) l[i] = 0;
    __syncthreads();
    uint32_t s = threadIdx.x * 747796405u + blockIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            acc += atomicAdd(&l[(s >> 8) & 4095], 1u);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = t1 - t0;
        out[2 * blockIdx.x + 1] = r1 - r0;
    }
    if (acc == 0xFFFFFFFF) out[0] = acc;
}

int main()
{
    uint32_t *dout;
    CHECK(hipMalloc(&dout, 64));
    {
        unsigned long long *dc;
        CHECK(hipMalloc(&dc, 1024 * 16));
        hipLaunchKernelGGL(clock_kernel, dim3(1024), dim3(512), 0, 0, dc, 20000);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(2048);
        CHECK(hipMemcpy(h.data(), dc, 2048 * 8, hipMemcpyDeviceToHost));
        printf("in-kernel clock under LDS-atomic load: %.0f MHz (block 0), %.0f MHz (block 1000)\n",
               100.0 * h[0] / h[1], 100.0 * h[2000] / h[2001]);
    }
    for (int waves : {8, 16}) {
        const int block = waves * 64 > 1024 ? 1024 : waves * 64;
        const int bpc = waves * 64 / block;
        run<ADD32>("ds_add_u32", block, bpc, 32768, dout);
        run<ADD32>("ds_add_u32", block, bpc, 512, dout);
        run<ADD32_REP32>("ds_add_u32 rep32", block, bpc, 512, dout);
        run<ADD32_RTN>("ds_add_rtn_u32", block, bpc, 512, dout);
        run<ADD32_RTN>("ds_add_rtn_u32", block, bpc, 32768, dout);
        run<ADD64>("ds_add_u64", block, bpc, 32768, dout);
        run<ADD32_NOCONF>("ds_add_u32 lane=bank", block, bpc, 32768, dout);
        run<ADD32_PAIRS>("ds_add_u32 lane pairs", block, bpc, 32768, dout);
        run<ADD64_NOCONF>("ds_add_u64 lane=bank", block, bpc, 32768, dout);
        run<WRITE16>("ds_write_b16", block, bpc, 32768, dout);
        run<WRITE32>("ds_write_b32", block, bpc, 32768, dout);
        run<READ32>("ds_read_b32", block, bpc, 32768, dout);
        run<READ32_NOCONF>("ds_read_b32 lane=bank", block, bpc, 32768, dout);
        run<ADD32_RTN_WRITE16>("add_rtn+write16", block, bpc, 512, dout);
    }
    // two 512-thread blocks per CU (the scatter kernel's shape)
    run<ADD32_RTN_WRITE16>("add_rtn+write16", 512, 2, 512, dout);
    run<ADD32>("ds_add_u32", 512, 2, 512, dout);
    return 0;
}
