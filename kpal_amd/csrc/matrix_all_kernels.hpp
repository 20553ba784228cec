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

template <int S>
struct MatrixAllGeometry {
    static constexpr int kRows = 4 * S;                  // staged rows (profiles, clamped to P - 1)
    static constexpr int kThreads = 4 * S * S;
    static constexpr int kOff = S * (S - 1) / 2;         // off-diagonal 4 x 4 tiles
    static constexpr int kSlots = kOff + S / 2;
    static constexpr int kPieces = kRows * 32;           // 16-byte pieces of a stage
    static constexpr int kLoads = (kPieces + kThreads - 1) / kThreads;
};

// slot -> blocks (xb, yb): off-diagonal tile (xb > yb) in lexicographic order, then the diagonal slots (xb = 2q, yb = 2q + 1)
template <int S>
__device__ __forceinline__ void matrix_all_slot(int slot, int &xb, int &yb, bool &diag)
{
    constexpr int OFF = MatrixAllGeometry<S>::kOff;
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

// SUM = false: multiset 'prod' as | 1/(y+1) - 1/(x+1) | (see matrix_rdiff_kernel for the identity and its accuracy bound);
// the staged values are the reciprocals.  A count >= kRdiffMaxCount (or negative) raises *big: the caller reruns the
// pair-of-counts kernel.
// partial layout: that of the super-tile kernels -- [tile t = ti (ti + 1) / 2 + tj][entry a * 4 + b][group], .s from the
// slot's lane 0, .m (bins seen - bins where both are zero) from thread (i, j).
template <int S>
__global__ __launch_bounds__(4 * S * S) void matrix_rdiff_all_kernel(const int64_t *__restrict__ prof, int P, uint64_t n,
                                                                      Partial *__restrict__ partials, uint32_t *__restrict__ big)
{
    using G = MatrixAllGeometry<S>;
    constexpr int R = G::kRows, NT = G::kThreads;
    __shared__ __attribute__((aligned(16))) double rstage[2][R][kSuperBins];
    __shared__ unsigned long long zmask[2][R];
    __shared__ double rtable[kRdiffTable];
    for (int i = threadIdx.x; i < kRdiffTable; i += NT) rtable[i] = rcp_counts((double)i + 1.0);
    const uint32_t group = blockIdx.x, ngroups = gridDim.x;
    const int slot = threadIdx.x >> 3, l = threadIdx.x & 7;
    int xb, yb;
    bool diag;
    matrix_all_slot<S>(slot, xb, yb, diag);
    const int side = (P + 3) / 4;
    const bool mine = xb < side;                        // (yb < xb off the diagonal; a diagonal slot whose second block is past the end computes clamped rows nobody reads)
    const int flip = (slot >> 1) & 1;
    double s[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) s[a][b] = 0.0;
    // term counts: thread t holds the pairs (i, jq + S m), m = 0..3, of row i = t / S
    const int zi = threadIdx.x / S, zq = threadIdx.x % S;
    uint32_t both_zero[4] = {0u, 0u, 0u, 0u};
    bool saw_big = false;
    // loader: piece p = t + NT i is the 16 bytes (two bins) p & 31 of staged row p >> 5: a wave reads two 512-byte runs
    const int64_t *src[G::kLoads];
#pragma unroll
    for (int i = 0; i < G::kLoads; ++i) {
        const int p = (int)threadIdx.x + NT * i;
        src[i] = prof + (uint64_t)min(p >> 5, P - 1) * n + 2 * (p & 31);
    }
    const int lhalf = (threadIdx.x >> 5) & 1, lcol = threadIdx.x & 31;
    __syncthreads();                                   // the table
    auto recip = [&](int64_t v, bool all_small) -> double {
        if (all_small) return rtable[(uint32_t)v];
        saw_big |= (unsigned long long)v >= kRdiffMaxCount;
        return (unsigned long long)v < (unsigned long long)kRdiffTable ? rtable[(uint32_t)v & (kRdiffTable - 1)] : rcp_counts((double)(uint32_t)v + 1.0);
    };
    auto put = [&](int buf, int i, const longlong2 &v) {
        const int p = (int)threadIdx.x + NT * i;
        if (G::kPieces % NT != 0 && p >= G::kPieces) return;   // (whole half-waves)
        const int row = p >> 5;
        const bool all_small = __all((unsigned long long)v.x < (unsigned long long)kRdiffTable && (unsigned long long)v.y < (unsigned long long)kRdiffTable);
        double2 r;
        r.x = recip(v.x, all_small);
        r.y = recip(v.y, all_small);
        *reinterpret_cast<double2 *>(&rstage[buf][row][2 * lcol]) = r;
        // zero masks: bit c = bin 2c, bit 32 + c = bin 2c + 1 (the same permutation of the bins in every row)
        const unsigned long long z0 = __builtin_amdgcn_ballot_w64(v.x == 0), z1 = __builtin_amdgcn_ballot_w64(v.y == 0);
        if (lcol == 0) zmask[buf][row] = lhalf ? ((z0 >> 32) | (z1 & 0xFFFFFFFF00000000ull)) : ((z0 & 0xFFFFFFFFull) | (z1 << 32));
    };
    const uint64_t chunks = n / kSuperBins;
    auto request = [&](longlong2 (&dst)[G::kLoads], uint64_t chunk) {
        if (chunk < chunks) {                          // block-uniform
#pragma unroll
            for (int i = 0; i < G::kLoads; ++i) {
                if (G::kPieces % NT != 0 && (int)threadIdx.x + NT * i >= G::kPieces) continue;
                dst[i] = *reinterpret_cast<const longlong2 *>(src[i] + chunk * kSuperBins);
            }
        }
    };
    auto compute = [&](int cur) {
        {
            const unsigned long long zr = zmask[cur][zi];
            if (zr != 0) {
#pragma unroll
                for (int m = 0; m < 4; ++m) both_zero[m] += (uint32_t)__popcll(zr & zmask[cur][zq + S * m]);
            }
        }
        if (!mine) return;
        const double *xrow = &rstage[cur][4 * xb][2 * l], *yrow = &rstage[cur][4 * yb][2 * l];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int off = 16 * (u ^ flip);
            double2 rx[4], ry[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                rx[a] = *reinterpret_cast<const double2 *>(xrow + a * kSuperBins + off);
                ry[a] = *reinterpret_cast<const double2 *>(yrow + a * kSuperBins + off);
            }
            if (!diag) {
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
    };
    longlong2 next[G::kLoads];
    uint64_t c = group;
    uint64_t stages = 0;
    request(next, c);
    if (c < chunks) {
#pragma unroll
        for (int i = 0; i < G::kLoads; ++i) put(0, i, next[i]);
    }
    __syncthreads();
    int cur = 0;
    for (; c < chunks; c += ngroups, ++stages) {
        const bool more = c + ngroups < chunks;        // block-uniform
        request(next, c + ngroups);
        compute(cur);
        if (more) {
#pragma unroll
            for (int i = 0; i < G::kLoads; ++i) put(cur ^ 1, i, next[i]);
        }
        __syncthreads();
        cur ^= 1;
    }
    if (saw_big) atomicOr(big, 1u);
    // sums: reduction over the slot's eight lanes (fixed order), lane 0 writes .s
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
                if (mine) partials[((tx + yb) * 16 + a * 4 + b) * ngroups + group].s = ps;
            } else if (a > b) {                        // pair (a, b) of block xb
                if (xb < side) partials[((tx + xb) * 16 + a * 4 + b) * ngroups + group].s = ps;
            } else {                                   // s[a][b], a < b: pair (b, a) of block yb
                if (yb < side) partials[((ty + yb) * 16 + b * 4 + a) * ngroups + group].s = ps;
            }
        }
    // term counts
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int i = zi, j = zq + S * m;
        if (j < i && i < P) {
            const int pti = i / 4, ptj = j / 4;
            const uint64_t t = (uint64_t)pti * (pti + 1) / 2 + ptj;
            partials[(t * 16 + (i % 4) * 4 + (j % 4)) * ngroups + group].m = stages * kSuperBins - both_zero[m];
        }
    }
}

}  // namespace kpal
