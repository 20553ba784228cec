// valu_bench.hip -- issue cost of the vector instructions the counting kernels are made of, on gfx950, at 1, 2 and 4 waves per
// SIMD (review of round 5, item 2: docs/NOTEBOOK.md priced the scatter's VALU stream at 4 cycles per wave-instruction, the
// microarchitecture guide measures v_fma_f32 at 2 with more than one wave on the SIMD).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_bench tools/valu_bench.hip && /tmp/valu_bench
// Every kernel runs ITERS x 32 copies of one instruction on eight independent register chains (a chain's next instruction is
// eight issue slots away), one workgroup per CU of 256 / 512 / 1024 threads = 1 / 2 / 4 waves per SIMD.  Reported: cycles per
// wave-instruction per SIMD = KERNEL time x clock / (instructions per wave x waves per SIMD), with the clock read inside the
// kernel (s_memtime against the 100 MHz s_memrealtime).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                              \
    do {                                                      \
        hipError_t e = (x);                                   \
        if (e != hipSuccess) {                                \
            printf("%s: %s\n", #x, hipGetErrorString(e));     \
            exit(1);                                          \
        }                                                     \
    } while (0)

enum Op {
    FMA_F32 = 0, ADD_F64, FMA_F64, PERM, DOT4, LSHL_OR, AND_OR, BFE, LSHRREV_B64, MOV_DPP_SHR, READLANE, ADD_U32, LSHLREV, XOR, CNDMASK,
    BCNT, MBCNT, CMP_EQ, ADD3, MUL_LO, MUL_U24, MAD_U24, SAD, CVT_F64_U32, ALIGNBIT, PK_ADD_F32, ADD_LSHL, MOV, LSHL_ADD, OR3, MIX_SCATTER, NOPS,
    BITOP3, LSHRREV, AND, OR, SUB, NOT, CNDMASK_SGPR, CNDMASK_ALT, LSHL_ADD_U64, CMP_GT_U64, WRITELANE, CMP_CNDMASK, ABSDIFF_F64, ABSDIFF_F64_FAR
};

#define ONE8(INS)                                                                                                                 \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                                          \
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])                \
                 : "v"(c0), "v"(c1), "s"(sc)                                                                                      \
                 : "vcc")
#define ONE8D(INS)                                                                                                                \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                                          \
                 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])                \
                 : "v"(dc)                                                                                                        \
                 :)

#define I_FMA_F32(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define I_DOT4(n) "v_dot4_u32_u8 %" #n ", %" #n ", %8, %9\n"
#define I_LSHL_OR(n) "v_lshl_or_b32 %" #n ", %" #n ", 3, %9\n"
#define I_AND_OR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define I_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 3, 20\n"
#define I_MOV_DPP(n) "v_mov_b32_dpp %" #n ", %" #n " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_READLANE(n) "v_readlane_b32 s40, %" #n ", 5\n"
#define I_ADD_U32(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_LSHLREV(n) "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
#define I_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define I_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_BCNT(n) "v_bcnt_u32_b32 %" #n ", %8, %" #n "\n"
#define I_MBCNT(n) "v_mbcnt_lo_u32_b32 %" #n ", %8, %" #n "\n"
#define I_CMP_EQ(n) "v_cmp_eq_u32 vcc, %" #n ", %8\n"
#define I_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define I_MUL_LO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define I_MUL_U24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define I_MAD_U24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n"
#define I_SAD(n) "v_sad_u32 %" #n ", %" #n ", %8, %9\n"
#define I_ALIGNBIT(n) "v_alignbit_b32 %" #n ", %" #n ", %8, 7\n"
#define I_PK_ADD_F32(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n"
#define I_ADD_LSHL(n) "v_add_lshl_u32 %" #n ", %" #n ", %8, 3\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define I_LSHL_ADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 2, %9\n"
#define I_OR3(n) "v_or3_b32 %" #n ", %" #n ", %8, %9\n"
#define I_NOP(n) "s_nop 0\n"
#define I_BITOP3(n) "v_bitop3_b32 %" #n ", %" #n ", %8, %9 bitop3:0x96\n"
#define I_LSHRREV(n) "v_lshrrev_b32 %" #n ", 3, %" #n "\n"
#define I_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define I_OR(n) "v_or_b32 %" #n ", %" #n ", %8\n"
#define I_SUB(n) "v_sub_u32 %" #n ", %" #n ", %8\n"
#define I_NOT(n) "v_not_b32 %" #n ", %" #n "\n"
#define I_CNDMASK_SGPR(n) "v_cndmask_b32 %" #n ", %" #n ", %8, s[42:43]\n"
#define I_WRITELANE(n) "v_writelane_b32 %" #n ", s44, 7\n"
#define I_LSHL_ADD_U64(n) "v_lshl_add_u64 %" #n ", %" #n ", 0, %8\n"
#define I_CMP_GT_U64(n) "v_cmp_gt_u64 vcc, %" #n ", %8\n"
#define I_ADD_F64(n) "v_add_f64 %" #n ", %" #n ", %8\n"
#define I_FMA_F64(n) "v_fma_f64 %" #n ", %" #n ", %8, %" #n "\n"
#define I_LSHRREV_B64(n) "v_lshrrev_b64 %" #n ", 3, %" #n "\n"
#define I_CVT_F64_U32(n) "v_cvt_f64_u32 %" #n ", %8\n"

template <int OP>
__global__ __launch_bounds__(1024) void bench_kernel(int iters, uint32_t seed, unsigned long long *out)
{
    uint32_t r[8];
    double d[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        r[u] = seed * (threadIdx.x + 1u) + u;
        d[u] = 1.0 + (double)(threadIdx.x + u) * 1e-9;
    }
    double e[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) e[u] = 1.0 / (double)(threadIdx.x * 4 + u + seed + 1);   // (operands with full mantissas: the power they draw)
    const uint32_t c0 = seed ^ 0x01020304u, c1 = seed | 0x03020100u;
    const uint32_t sc = seed;
    const double dc = 1.0000001;
    const float2 fc = make_float2(1.5f, 0.5f);
    (void)fc;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if constexpr (OP == FMA_F32) ONE8(I_FMA_F32);
            else if constexpr (OP == PERM) ONE8(I_PERM);
            else if constexpr (OP == DOT4) ONE8(I_DOT4);
            else if constexpr (OP == LSHL_OR) ONE8(I_LSHL_OR);
            else if constexpr (OP == AND_OR) ONE8(I_AND_OR);
            else if constexpr (OP == BFE) ONE8(I_BFE);
            else if constexpr (OP == MOV_DPP_SHR) ONE8(I_MOV_DPP);
            else if constexpr (OP == READLANE) {
                asm volatile(I_READLANE(0) I_READLANE(1) I_READLANE(2) I_READLANE(3) I_READLANE(4) I_READLANE(5) I_READLANE(6) I_READLANE(7)
                             :
                             : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7])
                             : "s40");
            } else if constexpr (OP == ADD_U32) ONE8(I_ADD_U32);
            else if constexpr (OP == LSHLREV) ONE8(I_LSHLREV);
            else if constexpr (OP == XOR) ONE8(I_XOR);
            else if constexpr (OP == CNDMASK) ONE8(I_CNDMASK);
            else if constexpr (OP == BCNT) ONE8(I_BCNT);
            else if constexpr (OP == MBCNT) ONE8(I_MBCNT);
            else if constexpr (OP == CMP_EQ) ONE8(I_CMP_EQ);
            else if constexpr (OP == ADD3) ONE8(I_ADD3);
            else if constexpr (OP == MUL_LO) ONE8(I_MUL_LO);
            else if constexpr (OP == MUL_U24) ONE8(I_MUL_U24);
            else if constexpr (OP == MAD_U24) ONE8(I_MAD_U24);
            else if constexpr (OP == SAD) ONE8(I_SAD);
            else if constexpr (OP == ALIGNBIT) ONE8(I_ALIGNBIT);
            else if constexpr (OP == ADD_LSHL) ONE8(I_ADD_LSHL);
            else if constexpr (OP == MOV) ONE8(I_MOV);
            else if constexpr (OP == LSHL_ADD) ONE8(I_LSHL_ADD);
            else if constexpr (OP == OR3) ONE8(I_OR3);
            else if constexpr (OP == NOPS) ONE8(I_NOP);
            else if constexpr (OP == BITOP3) ONE8(I_BITOP3);
            else if constexpr (OP == LSHRREV) ONE8(I_LSHRREV);
            else if constexpr (OP == AND) ONE8(I_AND);
            else if constexpr (OP == OR) ONE8(I_OR);
            else if constexpr (OP == SUB) ONE8(I_SUB);
            else if constexpr (OP == NOT) ONE8(I_NOT);
            else if constexpr (OP == CNDMASK_SGPR) {
                asm volatile(I_CNDMASK_SGPR(0) I_CNDMASK_SGPR(1) I_CNDMASK_SGPR(2) I_CNDMASK_SGPR(3) I_CNDMASK_SGPR(4) I_CNDMASK_SGPR(5) I_CNDMASK_SGPR(6) I_CNDMASK_SGPR(7)
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                             : "v"(c0)
                             : "s42", "s43");
            } else if constexpr (OP == CNDMASK_ALT) {   // every second instruction something else: is it the back-to-back select that is slow?
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_xor_b32 %1, %1, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_xor_b32 %3, %3, %8\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\n v_xor_b32 %5, %5, %8\n v_cndmask_b32 %6, %6, %8, vcc\n v_xor_b32 %7, %7, %8\n"
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                             : "v"(c0)
                             : "vcc");
            } else if constexpr (OP == CMP_CNDMASK) {   // the usual pair: a compare that writes vcc, the select that reads it
                asm volatile("v_cmp_gt_u32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_gt_u32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cmp_gt_u32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_gt_u32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n"
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                             : "v"(c0)
                             : "vcc");
            } else if constexpr (OP == WRITELANE) {
                asm volatile(I_WRITELANE(0) I_WRITELANE(1) I_WRITELANE(2) I_WRITELANE(3) I_WRITELANE(4) I_WRITELANE(5) I_WRITELANE(6) I_WRITELANE(7)
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                             :
                             : "s44");
            } else if constexpr (OP == ABSDIFF_F64 || OP == ABSDIFF_F64_FAR) {
                // the matrix kernels' term: t = x - y; acc += |t| -- three different register pairs per instruction, the second
                // instruction reads the first one's result (ABSDIFF_F64: back to back; _FAR: four instructions later)
                if constexpr (OP == ABSDIFF_F64) {
                    asm volatile("v_add_f64 %0, %8, -%9\n v_add_f64 %4, |%0|, %4\n v_add_f64 %1, %9, -%10\n v_add_f64 %5, |%1|, %5\n"
                                 "v_add_f64 %2, %10, -%11\n v_add_f64 %6, |%2|, %6\n v_add_f64 %3, %11, -%8\n v_add_f64 %7, |%3|, %7\n"
                                 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
                                 : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3])
                                 :);
                } else {
                    asm volatile("v_add_f64 %0, %8, -%9\n v_add_f64 %1, %9, -%10\n v_add_f64 %2, %10, -%11\n v_add_f64 %3, %11, -%8\n"
                                 "v_add_f64 %4, |%0|, %4\n v_add_f64 %5, |%1|, %5\n v_add_f64 %6, |%2|, %6\n v_add_f64 %7, |%3|, %7\n"
                                 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
                                 : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3])
                                 :);
                }
            } else if constexpr (OP == LSHL_ADD_U64) ONE8D(I_LSHL_ADD_U64);
            else if constexpr (OP == CMP_GT_U64) {
                asm volatile(I_CMP_GT_U64(0) I_CMP_GT_U64(1) I_CMP_GT_U64(2) I_CMP_GT_U64(3) I_CMP_GT_U64(4) I_CMP_GT_U64(5) I_CMP_GT_U64(6) I_CMP_GT_U64(7)
                             :
                             : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]), "v"(dc)
                             : "vcc");
            }
            else if constexpr (OP == ADD_F64) ONE8D(I_ADD_F64);
            else if constexpr (OP == FMA_F64) ONE8D(I_FMA_F64);
            else if constexpr (OP == LSHRREV_B64) ONE8D(I_LSHRREV_B64);
            else if constexpr (OP == CVT_F64_U32) {
                asm volatile(I_CVT_F64_U32(0) I_CVT_F64_U32(1) I_CVT_F64_U32(2) I_CVT_F64_U32(3) I_CVT_F64_U32(4) I_CVT_F64_U32(5) I_CVT_F64_U32(6) I_CVT_F64_U32(7)
                             : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
                             : "v"(c0)
                             :);
            } else if constexpr (OP == PK_ADD_F32) {
                asm volatile(I_PK_ADD_F32(0) I_PK_ADD_F32(1) I_PK_ADD_F32(2) I_PK_ADD_F32(3) I_PK_ADD_F32(4) I_PK_ADD_F32(5) I_PK_ADD_F32(6) I_PK_ADD_F32(7)
                             : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
                             : "v"(dc)
                             :);
            } else if constexpr (OP == MIX_SCATTER) {
                // the blend of the scatter's inner loop, eight instructions of the commonest opcodes per chain-round
                asm volatile("v_perm_b32 %0, %0, %8, %9\n"
                             "v_lshl_or_b32 %1, %1, 3, %9\n"
                             "v_and_b32 %2, %2, %8\n"
                             "v_dot4_u32_u8 %3, %3, %8, %9\n"
                             "v_lshrrev_b32 %4, 3, %4\n"
                             "v_bfe_u32 %5, %5, 3, 20\n"
                             "v_or_b32 %6, %6, %8\n"
                             "v_cndmask_b32 %7, %7, %8, vcc\n"
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                             : "v"(c0), "v"(c1)
                             : "vcc");
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= r[u] ^ (uint32_t)__double_as_longlong(d[u]);
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = t1 - t0;
        out[2 * blockIdx.x + 1] = w1 - w0;
    }
    if (acc == 0x12345678u && seed == 77u) out[0] = acc;
}

// (the time is the KERNEL's -- the SIMD issues oldest-first, so the first wave of a workgroup runs as if it were alone whatever
// the others do; its own cycle count says nothing about the issue rate.  The in-kernel counters only calibrate the clock.)
template <int OP>
void run(const char *name, unsigned long long *dout)
{
    const int iters = 20000;
    printf("%-22s", name);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int block : {256, 512, 1024}) {
        hipLaunchKernelGGL(bench_kernel<OP>, dim3(256), dim3(block), 0, 0, 100, 3u, dout);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(bench_kernel<OP>, dim3(256), dim3(block), 0, 0, iters, 3u, dout);
        CHECK(hipEventRecord(b));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        std::vector<unsigned long long> h(512);
        CHECK(hipMemcpy(h.data(), dout, 512 * 8, hipMemcpyDeviceToHost));
        double mhz = 0;
        for (int blk = 0; blk < 256; ++blk) mhz += 100.0 * (double)h[2 * blk] / (double)h[2 * blk + 1];
        mhz /= 256;
        // iters x 32 instructions per wave, (block / 256) waves per SIMD
        const double cyc = (double)ms * 1e-3 * mhz * 1e6 / ((double)iters * 32 * (block / 256));
        printf("  %d/SIMD: %5.2f cyc (%4.0f MHz)", block / 256, cyc, mhz);
    }
    printf("\n");
    CHECK(hipEventDestroy(a));
    CHECK(hipEventDestroy(b));
}

int main()
{
    unsigned long long *dout;
    CHECK(hipMalloc(&dout, 512 * 8));
    printf("cycles per wave-instruction per SIMD (issue cost with W waves on the SIMD; clock measured in the kernel)\n");
    run<NOPS>("s_nop 0", dout);
    run<FMA_F32>("v_fma_f32", dout);
    run<MOV>("v_mov_b32", dout);
    run<ADD_U32>("v_add_u32", dout);
    run<XOR>("v_xor_b32", dout);
    run<LSHLREV>("v_lshlrev_b32", dout);
    run<ADD3>("v_add3_u32", dout);
    run<OR3>("v_or3_b32", dout);
    run<LSHL_OR>("v_lshl_or_b32", dout);
    run<LSHL_ADD>("v_lshl_add_u32", dout);
    run<ADD_LSHL>("v_add_lshl_u32", dout);
    run<AND_OR>("v_and_or_b32", dout);
    run<BFE>("v_bfe_u32", dout);
    run<ALIGNBIT>("v_alignbit_b32", dout);
    run<PERM>("v_perm_b32", dout);
    run<DOT4>("v_dot4_u32_u8", dout);
    run<SAD>("v_sad_u32", dout);
    run<CNDMASK>("v_cndmask_b32", dout);
    run<CMP_EQ>("v_cmp_eq_u32 -> vcc", dout);
    run<BCNT>("v_bcnt_u32_b32", dout);
    run<MBCNT>("v_mbcnt_lo_u32_b32", dout);
    run<MOV_DPP_SHR>("v_mov_b32 dpp wave_shr", dout);
    run<READLANE>("v_readlane_b32", dout);
    run<MUL_U24>("v_mul_u32_u24", dout);
    run<MAD_U24>("v_mad_u32_u24", dout);
    run<MUL_LO>("v_mul_lo_u32", dout);
    run<LSHRREV_B64>("v_lshrrev_b64", dout);
    run<PK_ADD_F32>("v_pk_add_f32", dout);
    run<ADD_F64>("v_add_f64", dout);
    run<FMA_F64>("v_fma_f64", dout);
    run<CVT_F64_U32>("v_cvt_f64_u32", dout);
    run<MIX_SCATTER>("mix of 8 (scatter)", dout);
    run<AND>("v_and_b32", dout);
    run<OR>("v_or_b32", dout);
    run<SUB>("v_sub_u32", dout);
    run<NOT>("v_not_b32", dout);
    run<LSHRREV>("v_lshrrev_b32", dout);
    run<BITOP3>("v_bitop3_b32", dout);
    run<CNDMASK_SGPR>("v_cndmask_b32 (sgpr)", dout);
    run<CNDMASK_ALT>("v_cndmask + v_xor", dout);
    run<CMP_CNDMASK>("v_cmp + v_cndmask", dout);
    run<WRITELANE>("v_writelane_b32", dout);
    run<LSHL_ADD_U64>("v_lshl_add_u64", dout);
    run<CMP_GT_U64>("v_cmp_gt_u64 -> vcc", dout);
    run<ABSDIFF_F64>("t=x-y; acc+=|t| (f64)", dout);
    run<ABSDIFF_F64_FAR>("... 4 apart", dout);
    return 0;
}
