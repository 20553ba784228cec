// matrix_all_kernels.hpp -- the multiset distance matrix (kdistlib.distance_matrix, kpal/kdistlib.py:164-186; multiset,
// metrics.py:101-123; pairwise prod / sum, metrics.py:159-162) for up to 64 profiles with EVERY profile staged ONCE per
// bin range (gfx950).
//
// The super-tile kernels of vec_kernels.hpp (matrix_rdiff / matrix_rsum) give a workgroup one 16 x 16 block of pairs, so a
// profile is loaded, converted and written to LDS by every super-tile of its block row and column: 5 x at 64 profiles
// (43 GB of loads for 8.6 GB of profiles at k = 12), and that loader, not the arithmetic, set their time.  Here ONE
// workgroup stages 64 bins of ALL 4 S rows (S = 16: 64 profiles, 32 KiB of reciprocals per buffer) and its 4 S^2 threads
// cover the whole lower triangle:
//   * the S (S - 1) / 2 off-diagonal 4 x 4 tiles, eight lanes each (a lane takes 8 of the 64 bins);
//   * the S diagonal 4 x 4 tiles hold six pairs each: an eight-lane slot takes TWO of them (rows of block 2q as x, of block
//     2q + 1 as y: the same eight LDS reads as a tile, twelve terms instead of sixteen) -- S / 2 slots.
//   S^2 / 2 slots x 8 lanes = 4 S^2 threads: 1024 for S = 16 (waves 0..14 off-diagonal, wave 15 the diagonal), 256 for S = 8.
// No pair is computed twice and no slot idles: the fp64 work is exactly 2 instructions per term.
//   LDS reads: ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... -- i.e. four
// lanes of each of four slots per LDS cycle, 64 bytes each from a different row: slot q of a wave walks its row's four
// 128-byte pieces in the order u ^ ((q >> 1) & 1), which puts the four pieces on four different quarters of the 64 banks
// whatever the rows are (same row and piece: a broadcast).  Conflict-free, 4 cycles per wave-instruction.
#pragma once
#include "vec_kernels.hpp"

namespace kpal {

#if !defined(KPAL_MALL_BINS)
#define KPAL_MALL_BINS 64
#endif
#if !defined(KPAL_MALL_UNITS)
#define KPAL_MALL_UNITS 1
#endif
constexpr int kMatrixAllBins = KPAL_MALL_BINS;        // bins of a stage at 33..64 profiles (64 or 128; A/B builds)
constexpr int kMatrixAllUnits = KPAL_MALL_UNITS;      // slots a thread works through at 33..64 profiles (1: 1024 threads, 2: 512)

template <int S, int B, int U>
struct MatrixAllGeometry {
    static constexpr int kRows = 4 * S;                  // staged rows (profiles, clamped to P - 1)
    static constexpr int kOff = S * (S - 1) / 2;         // off-diagonal 4 x 4 tiles
    static constexpr int kSlots = kOff + S / 2;          // = S^2 / 2
    static constexpr int kThreads = 8 * kSlots / U;
    static constexpr int kRowPieces = B / 2;             // 16-byte pieces (two bins) of a staged row
    static constexpr int kPieces = kRows * kRowPieces;
    static constexpr int kLoads = kPieces / kThreads;
    static constexpr int kMaskWords = B / 64;
    static constexpr int kPairs = kRows * (kRows - 1) / 2;
    static constexpr int kPairsPerThread = (kPairs + kThreads - 1) / kThreads;
};

// slot -> blocks (xb, yb): off-diagonal tile (xb > yb) in lexicographic order, then the diagonal slots (xb = 2q, yb = 2q + 1)
template <int S>
__device__ __forceinline__ void matrix_all_slot(int slot, int &xb, int &yb, bool &diag)
{
    constexpr int OFF = S * (S - 1) / 2;
    diag = slot >= OFF;
    if (diag) {
        xb = 2 * (slot - OFF);
        yb = xb + 1;
    } else {
        int ti = 1;
        while (ti * (ti + 1) / 2 <= slot) ++ti;
        xb = ti;
        yb = slot - ti * (ti - 1) / 2;
    }
}

// the terms of one slot and 16 bins: DIAG = false: the 16 pairs x[a] : y[b]; true: the six pairs inside x and the six inside y
template <bool DIAG>
__device__ __forceinline__ void matrix_all_terms(double (&s)[4][4], const double2 (&rx)[4], const double2 (&ry)[4])
{
    if constexpr (!DIAG) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                s[a][b] += fabs(rx[a].x - ry[b].x);
                s[a][b] += fabs(rx[a].y - ry[b].y);
            }
    } else {
#pragma unroll
        for (int a = 1; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < a; ++b) {
                s[a][b] += fabs(rx[a].x - rx[b].x);
                s[a][b] += fabs(rx[a].y - rx[b].y);
                s[b][a] += fabs(ry[a].x - ry[b].x);
                s[b][a] += fabs(ry[a].y - ry[b].y);
            }
    }
}

// Multiset 'prod' as | 1/(y+1) - 1/(x+1) | (see matrix_rdiff_kernel for the identity and its accuracy bound); the staged
// values are the reciprocals.  A count >= kRdiffMaxCount (or negative) raises *big: the caller reruns the pair-of-counts kernel.
// B bins per stage; U slots per thread (thread slot q works through the slots q, q + S^2 / (2 U), ...: U = 2 halves the
// waves and doubles their registers -- room to have the LDS reads of the next step in flight during the arithmetic of this one).
// partial layout: that of the super-tile kernels -- [tile t = ti (ti + 1) / 2 + tj][entry a * 4 + b][group]; .s from the
// slot's lane 0, .m (bins seen - bins where both are zero) from the thread that holds the pair's zero masks.
template <int S, int B, int U>
__global__ __launch_bounds__((MatrixAllGeometry<S, B, U>::kThreads), S == 16 ? 4 / U : 3) void matrix_rdiff_all_kernel(const int64_t *__restrict__ prof, int P, uint64_t n,
                                                                                                   Partial *__restrict__ partials, uint32_t *__restrict__ big)
{
    using G = MatrixAllGeometry<S, B, U>;
    constexpr int R = G::kRows, NT = G::kThreads, NL = G::kLoads, RP = G::kRowPieces, MW = G::kMaskWords, ZP = G::kPairsPerThread;
    static_assert(G::kPieces % NT == 0 && (B == 64 || B == 128) && G::kSlots % U == 0, "geometry");
    __shared__ __attribute__((aligned(16))) double rstage[2][R][B];
    __shared__ unsigned long long zmask[2][R][MW];
    __shared__ double rtable[kRdiffTable];
    for (int i = threadIdx.x; i < kRdiffTable; i += NT) rtable[i] = rcp_counts((double)i + 1.0);
    const uint32_t group = blockIdx.x, ngroups = gridDim.x;
    const int tslot = threadIdx.x >> 3, l = threadIdx.x & 7;
    const int side = (P + 3) / 4;
    int xb[U], yb[U];
    bool diag[U];
    bool mine = false;
#pragma unroll
    for (int v = 0; v < U; ++v) {
        matrix_all_slot<S>(tslot + v * (G::kSlots / U), xb[v], yb[v], diag[v]);
        mine |= xb[v] < side;
    }
    // (mine: one of the thread's slots holds a pair of real profiles; the others work on clamped rows and are never written)
    const int flip = (tslot >> 1) & 1;
    double s[U][4][4];
#pragma unroll
    for (int v = 0; v < U; ++v)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) s[v][a][b] = 0.0;
    // term counts: thread t holds the pairs number t, t + NT, ... of the lower triangle (pair p = i (i - 1) / 2 + j, j < i)
    int zi[ZP], zj[ZP];
    uint32_t both_zero[ZP];
#pragma unroll
    for (int h = 0; h < ZP; ++h) {
        const int p = min((int)threadIdx.x + NT * h, G::kPairs - 1);
        int i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
        while (i * (i - 1) / 2 > p) --i;
        while ((i + 1) * i / 2 <= p) ++i;
        zi[h] = i;
        zj[h] = p - i * (i - 1) / 2;
        both_zero[h] = 0u;
    }
    uint32_t hi_seen = 0;                              // OR of the counts' high words and of the low words >= kRdiffMaxCount
    // loader: piece p = t + NT i is the 16 bytes (two bins) p % RP of staged row p / RP
    const int64_t *src[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int p = (int)threadIdx.x + NT * i;
        src[i] = prof + (uint64_t)min(p / RP, P - 1) * n + 2 * (p % RP) + (uint64_t)group * B;
    }
    __syncthreads();                                   // the table
    // Staging a piece = two halves with the table's latency between them: `lookup` (after the piece's global load has landed:
    // clamp the counts to the table, issue the two reads) and `store` (the reciprocals, the zero masks).  The arithmetic of the
    // stage's last step runs between the two.  Counts past the table are rare (a wave-uniform test): their reciprocals are computed.
    auto lookup = [&](const longlong2 &v, double2 &r) {
        const uint32_t x = (uint32_t)v.x, y = (uint32_t)v.y;
        r.x = rtable[min(x, (uint32_t)kRdiffTable - 1u)];
        r.y = rtable[min(y, (uint32_t)kRdiffTable - 1u)];
    };
    auto store = [&](int buf, int i, const longlong2 &v, double2 r) {
        const int p = (int)threadIdx.x + NT * i;
        const int row = p / RP, col = p % RP;
        // (a count with a high word, or >= 2^16, makes the launch void -- *big -- so only the low words are looked at)
        const uint32_t x = (uint32_t)v.x, y = (uint32_t)v.y;
        hi_seen |= (uint32_t)((unsigned long long)v.x >> 32) | (uint32_t)((unsigned long long)v.y >> 32) | ((x | y) & ~(uint32_t)(kRdiffMaxCount - 1));
        if (__builtin_amdgcn_ballot_w64((x | y) >= (uint32_t)kRdiffTable) != 0) {   // wave-uniform, rare
            if (x >= (uint32_t)kRdiffTable) r.x = rcp_counts((double)x + 1.0);
            if (y >= (uint32_t)kRdiffTable) r.y = rcp_counts((double)y + 1.0);
        }
        *reinterpret_cast<double2 *>(&rstage[buf][row][2 * col]) = r;
        // zero masks: the same permutation of the bins in every row
        const unsigned long long z0 = __builtin_amdgcn_ballot_w64(x == 0), z1 = __builtin_amdgcn_ballot_w64(y == 0);
        if constexpr (B == 128) {                      // a wave = one row: word 0 the even bins, word 1 the odd ones
            if ((threadIdx.x & 63) == 0) {
                zmask[buf][row][0] = z0;
                zmask[buf][row][1] = z1;
            }
        } else {                                       // a wave = two rows: bit c = bin 2c, bit 32 + c = bin 2c + 1
            const unsigned long long mlo = (z0 & 0xFFFFFFFFull) | (z1 << 32), mhi = (z0 >> 32) | (z1 & 0xFFFFFFFF00000000ull);
            if ((threadIdx.x & 31) == 0) zmask[buf][row][0] = (threadIdx.x & 32) ? mhi : mlo;
        }
    };
    const uint64_t chunks = n / B;
    // (src[] runs ahead of the arithmetic: it always points at the next stage to request)
    const uint64_t hop = (uint64_t)ngroups * B;
    uint64_t creq = group;                             // the chunk src[] points at
    auto request = [&](longlong2 (&dst)[NL]) {
        if (creq < chunks) {                           // block-uniform
#pragma unroll
            for (int i = 0; i < NL; ++i) {
#if defined(KPAL_MALL_NOLOAD)   // ablation: everything but the global loads (1: a few repeating values, 2: scattered ones)
                {
                    const uint32_t h = ((uint32_t)threadIdx.x * 2654435761u) ^ ((uint32_t)creq * 0x9E3779B1u) ^ (uint32_t)(i * 0x85EBCA6Bu);
                    dst[i] = KPAL_MALL_NOLOAD == 2 ? longlong2{(long long)((h >> 7) & 31), (long long)((h >> 19) & 31)}
                                                   : longlong2{(long long)((threadIdx.x + creq) & 15), (long long)(creq & 7)};
                }
#else
                dst[i] = *reinterpret_cast<const longlong2 *>(src[i]);
#endif
                src[i] += hop;
            }
        }
        creq += ngroups;
    };
    // one step of the arithmetic: 16 bins (a double2 per lane) of the 4 + 4 rows of each of the thread's slots
    auto step = [&](int cur, int u) {
#if !defined(KPAL_MALL_NOCOMPUTE)
        if (!mine) return;
        const int off = 16 * (u ^ flip);
#pragma unroll
        for (int v = 0; v < U; ++v) {
            const double *xrow = &rstage[cur][4 * xb[v]][2 * l], *yrow = &rstage[cur][4 * yb[v]][2 * l];
            double2 rx[4], ry[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                rx[a] = *reinterpret_cast<const double2 *>(xrow + a * B + off);
                ry[a] = *reinterpret_cast<const double2 *>(yrow + a * B + off);
            }
            if (!diag[v]) matrix_all_terms<false>(s[v], rx, ry);
            else matrix_all_terms<true>(s[v], rx, ry);
        }
#endif
    };
#if defined(KPAL_MALL_PRIO)   // A/B: a fixed priority per wave of a SIMD (waves w, w + 4, w + 8, w + 12 share one)
    switch ((threadIdx.x >> 8) & 3u) {
    case 0: __builtin_amdgcn_s_setprio(3); break;
    case 1: __builtin_amdgcn_s_setprio(2); break;
    case 2: __builtin_amdgcn_s_setprio(1); break;
    default: __builtin_amdgcn_s_setprio(0); break;
    }
#endif
#if defined(KPAL_MALL_CLOCK)
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), wall0 = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int STEPS = B / 16;
    longlong2 next[NL];
    uint64_t c = group;
    uint64_t stages = 0;
    request(next);
    if (c < chunks) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            double2 r;
            lookup(next[i], r);
            store(0, i, next[i], r);
        }
    }
    __syncthreads();
    for (int cur = 0; c < chunks; c += ngroups, ++stages, cur ^= 1) {
        const bool more = c + ngroups < chunks;        // block-uniform
        request(next);                                 // the next stage's values: in flight during this stage's arithmetic
#if !defined(KPAL_MALL_NOCOMPUTE) && !defined(KPAL_MALL_NOZMASK)
        unsigned long long zr[ZP][MW], zc[ZP][MW];     // (read here, used after the first step: no wait of their own)
#pragma unroll
        for (int h = 0; h < ZP; ++h)
#pragma unroll
            for (int w = 0; w < MW; ++w) {
                zr[h][w] = zmask[cur][zi[h]][w];
                zc[h][w] = zmask[cur][zj[h]][w];
            }
#endif
        step(cur, 0);
#if !defined(KPAL_MALL_NOCOMPUTE) && !defined(KPAL_MALL_NOZMASK)
#pragma unroll
        for (int h = 0; h < ZP; ++h)
#pragma unroll
            for (int w = 0; w < MW; ++w) both_zero[h] += (uint32_t)__popcll(zr[h][w] & zc[h][w]);
#endif
#pragma unroll
        for (int u = 1; u < STEPS - 1; ++u) step(cur, u);
        double2 rnext[NL];
#if defined(KPAL_MALL_NOPUT)   // ablation: no conversion, no LDS stores (the first stage's values stay)
        const bool stage_it = more && next[0].x == 0x7fffffffffffffffll;
#else
        const bool stage_it = more;
#endif
        if (stage_it) {
#pragma unroll
            for (int i = 0; i < NL; ++i) lookup(next[i], rnext[i]);
        }
        step(cur, STEPS - 1);
        if (stage_it) {
#pragma unroll
            for (int i = 0; i < NL; ++i) store(cur ^ 1, i, next[i], rnext[i]);
        }
        __syncthreads();
    }
#if defined(KPAL_MALL_CLOCK)   // diagnostic builds: shader cycles and 100 MHz ticks of workgroup 0's main loop behind *big
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        reinterpret_cast<unsigned long long *>(big)[1] = __builtin_amdgcn_s_memtime() - clk0;
        reinterpret_cast<unsigned long long *>(big)[2] = __builtin_amdgcn_s_memrealtime() - wall0;
    }
#endif
    if (hi_seen) atomicOr(big, 1u);
    // sums: reduction over the slot's eight lanes (fixed order), lane 0 writes .s
#pragma unroll
    for (int v = 0; v < U; ++v) {
        const uint64_t tx = (uint64_t)xb[v] * (xb[v] + 1) / 2, ty = (uint64_t)yb[v] * (yb[v] + 1) / 2;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                double ps = s[v][a][b];
#pragma unroll
                for (int d = 4; d >= 1; d >>= 1) ps += __shfl_down(ps, d, 8);
                if (l != 0 || (diag[v] && a == b)) continue;
                if (!diag[v]) {
                    if (xb[v] < side) partials[((tx + yb[v]) * 16 + a * 4 + b) * ngroups + group].s = ps;
                } else if (a > b) {                    // pair (a, b) of block xb
                    if (xb[v] < side) partials[((tx + xb[v]) * 16 + a * 4 + b) * ngroups + group].s = ps;
                } else {                               // s[a][b], a < b: pair (b, a) of block yb
                    if (yb[v] < side) partials[((ty + yb[v]) * 16 + b * 4 + a) * ngroups + group].s = ps;
                }
            }
    }
    // term counts
#pragma unroll
    for (int h = 0; h < ZP; ++h) {
        const int i = zi[h], j = zj[h];
        if ((int)threadIdx.x + NT * h < G::kPairs && i < P) {
            const int pti = i / 4, ptj = j / 4;
            const uint64_t t = (uint64_t)pti * (pti + 1) / 2 + ptj;
            partials[(t * 16 + (i % 4) * 4 + (j % 4)) * ngroups + group].m = stages * B - both_zero[h];
        }
    }
}

// Multiset 'sum', |x - y| / (x + y + 1), with the reciprocal of the denominator from a table in LDS (see matrix_rsum_kernel:
// v_sad_u32, the table offset, one ds_read_b64, a conversion and one fused multiply-add per term) over the same slots; the
// staged values are the counts as 32-bit integers, a lane takes 4 bins of a row per step (one ds_read_b128).  A count >=
// kRsumTable / 2 anywhere raises *big and the caller reruns matrix_super_kernel<1>.
template <int S, int B>
__global__ __launch_bounds__((MatrixAllGeometry<S, B, 1>::kThreads), 4) void matrix_rsum_all_kernel(const int64_t *__restrict__ prof, int P, uint64_t n,
                                                                                                  Partial *__restrict__ partials, uint32_t *__restrict__ big)
{
    using G = MatrixAllGeometry<S, B, 1>;
    constexpr int R = G::kRows, NT = G::kThreads, NL = G::kLoads, RP = G::kRowPieces, MW = G::kMaskWords, ZP = G::kPairsPerThread;
    static_assert(G::kPieces % NT == 0 && B == 64, "geometry");
    __shared__ __attribute__((aligned(16))) uint32_t cstage[2][R][B];
    __shared__ unsigned long long zmask[2][R][MW];
    // The staged values are the counts TIMES EIGHT -- the byte offset of a table entry is then one v_add_u32 (a half-price
    // instruction, tools/valu_bench.hip) instead of v_add_lshl_u32 -- and the table holds 1 / (8 (s + 1)): v_sad_u32 of two
    // staged values is 8 |x - y|, and the power of two cancels exactly.
    __shared__ double rtable[kRsumTable];
    for (int i = threadIdx.x; i < kRsumTable; i += NT) rtable[i] = rcp_counts((double)i + 1.0) * 0.125;
    const uint32_t group = blockIdx.x, ngroups = gridDim.x;
    const int slot = threadIdx.x >> 3, l = threadIdx.x & 7;
    const int side = (P + 3) / 4;
    int xb, yb;
    bool diag;
    matrix_all_slot<S>(slot, xb, yb, diag);
    const bool mine = xb < side;
    const int flip = (slot >> 1) & 1;
    double s[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) s[a][b] = 0.0;
    int zi[ZP], zj[ZP];
    uint32_t both_zero[ZP];
#pragma unroll
    for (int h = 0; h < ZP; ++h) {
        const int p = min((int)threadIdx.x + NT * h, G::kPairs - 1);
        int i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
        while (i * (i - 1) / 2 > p) --i;
        while ((i + 1) * i / 2 <= p) ++i;
        zi[h] = i;
        zj[h] = p - i * (i - 1) / 2;
        both_zero[h] = 0u;
    }
    uint32_t hi_seen = 0;                              // OR of the counts' high words and of the low words >= kRsumTable / 2
    const int64_t *src[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int p = (int)threadIdx.x + NT * i;
        src[i] = prof + (uint64_t)min(p / RP, P - 1) * n + 2 * (p % RP) + (uint64_t)group * B;
    }
    __syncthreads();                                   // the table
    auto store = [&](int buf, int i, const longlong2 &v) {
        const int p = (int)threadIdx.x + NT * i;
        const int row = p / RP, col = p % RP;
        const uint32_t x = (uint32_t)v.x, y = (uint32_t)v.y;
        hi_seen |= (uint32_t)((unsigned long long)v.x >> 32) | (uint32_t)((unsigned long long)v.y >> 32) | ((x | y) & ~(uint32_t)(kRsumTable / 2 - 1));
        // (masked: a larger count only ever costs a rerun, never an out-of-range read)
        *reinterpret_cast<uint2 *>(&cstage[buf][row][2 * col]) = make_uint2((x & (uint32_t)(kRsumTable / 2 - 1)) << 3, (y & (uint32_t)(kRsumTable / 2 - 1)) << 3);
        const unsigned long long z0 = __builtin_amdgcn_ballot_w64(x == 0), z1 = __builtin_amdgcn_ballot_w64(y == 0);
        const unsigned long long mlo = (z0 & 0xFFFFFFFFull) | (z1 << 32), mhi = (z0 >> 32) | (z1 & 0xFFFFFFFF00000000ull);
        if ((threadIdx.x & 31) == 0) zmask[buf][row][0] = (threadIdx.x & 32) ? mhi : mlo;
    };
    const uint64_t chunks = n / B;
    const uint64_t hop = (uint64_t)ngroups * B;
    uint64_t creq = group;
    auto request = [&](longlong2 (&dst)[NL]) {
        if (creq < chunks) {                           // block-uniform
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                dst[i] = *reinterpret_cast<const longlong2 *>(src[i]);
                src[i] += hop;
            }
        }
        creq += ngroups;
    };
    const char *tab = reinterpret_cast<const char *>(rtable);
    auto term4 = [&](double &acc, const uint4 &x, const uint4 &y) {
        const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint32_t d;
            asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(xs[e]), "v"(ys[e]));   // 8 |x - y|
            const double r = *reinterpret_cast<const double *>(tab + (xs[e] + ys[e]));
            acc = fma((double)d, r, acc);
        }
    };
    // one step: 32 bins (four per lane) of the slot's 4 + 4 rows
    auto step = [&](int cur, int u) {
        if (!mine) return;
        const uint32_t *xrow = &cstage[cur][4 * xb][4 * l], *yrow = &cstage[cur][4 * yb][4 * l];
        const int off = 32 * (u ^ flip);
        uint4 cx[4], cy[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            cx[a] = *reinterpret_cast<const uint4 *>(xrow + a * B + off);
            cy[a] = *reinterpret_cast<const uint4 *>(yrow + a * B + off);
        }
        if (!diag) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    // (one pair at a time: with the offsets of all 64 terms up front the registers run out; the other waves of the SIMD cover the reads)
                    asm volatile("" : "+v"(cy[b].x), "+v"(cy[b].y), "+v"(cy[b].z), "+v"(cy[b].w));
                    term4(s[a][b], cx[a], cy[b]);
                }
        } else {
#pragma unroll
            for (int a = 1; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < a; ++b) {
                    asm volatile("" : "+v"(cx[b].x), "+v"(cx[b].y), "+v"(cx[b].z), "+v"(cx[b].w));
                    term4(s[a][b], cx[a], cx[b]);
                    asm volatile("" : "+v"(cy[b].x), "+v"(cy[b].y), "+v"(cy[b].z), "+v"(cy[b].w));
                    term4(s[b][a], cy[a], cy[b]);
                }
        }
    };
    constexpr int STEPS = B / 32;
    longlong2 next[NL];
    uint64_t c = group;
    uint64_t stages = 0;
    request(next);
    if (c < chunks) {
#pragma unroll
        for (int i = 0; i < NL; ++i) store(0, i, next[i]);
    }
    __syncthreads();
    for (int cur = 0; c < chunks; c += ngroups, ++stages, cur ^= 1) {
        const bool more = c + ngroups < chunks;        // block-uniform
        request(next);
        unsigned long long zr[ZP][MW], zc[ZP][MW];
#pragma unroll
        for (int h = 0; h < ZP; ++h)
#pragma unroll
            for (int w = 0; w < MW; ++w) {
                zr[h][w] = zmask[cur][zi[h]][w];
                zc[h][w] = zmask[cur][zj[h]][w];
            }
        step(cur, 0);
#pragma unroll
        for (int h = 0; h < ZP; ++h)
#pragma unroll
            for (int w = 0; w < MW; ++w) both_zero[h] += (uint32_t)__popcll(zr[h][w] & zc[h][w]);
#pragma unroll
        for (int u = 1; u < STEPS; ++u) step(cur, u);
        if (more) {
#pragma unroll
            for (int i = 0; i < NL; ++i) store(cur ^ 1, i, next[i]);
        }
        __syncthreads();
    }
    if (hi_seen) atomicOr(big, 1u);
    {
        const uint64_t tx = (uint64_t)xb * (xb + 1) / 2, ty = (uint64_t)yb * (yb + 1) / 2;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                double ps = s[a][b];
#pragma unroll
                for (int d = 4; d >= 1; d >>= 1) ps += __shfl_down(ps, d, 8);
                if (l != 0 || (diag && a == b)) continue;
                if (!diag) {
                    if (xb < side) partials[((tx + yb) * 16 + a * 4 + b) * ngroups + group].s = ps;
                } else if (a > b) {
                    if (xb < side) partials[((tx + xb) * 16 + a * 4 + b) * ngroups + group].s = ps;
                } else {
                    if (yb < side) partials[((ty + yb) * 16 + b * 4 + a) * ngroups + group].s = ps;
                }
            }
    }
#pragma unroll
    for (int h = 0; h < ZP; ++h) {
        const int i = zi[h], j = zj[h];
        if ((int)threadIdx.x + NT * h < G::kPairs && i < P) {
            const int pti = i / 4, ptj = j / 4;
            const uint64_t t = (uint64_t)pti * (pti + 1) / 2 + ptj;
            partials[(t * 16 + (i % 4) * 4 + (j % 4)) * ngroups + group].m = stages * B - both_zero[h];
        }
    }
}

}  // namespace kpal
