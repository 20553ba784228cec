// gram_kernels.hpp -- euclidean distance matrix as an fp64 Gram contraction on the matrix cores (gfx950).
//
// metrics.euclidean (kpal/metrics.py:126-135) of every profile pair: |x - y|^2 = |x|^2 + |y|^2 - 2 x.y, so
// the P(P-1)/2 distances follow from the Gram matrix G = X X^T of the P x 4^k count matrix -- the one dense
// contraction of the hot path, and the only place the matrix cores are used (north_star).  fp64 MFMA keeps
// the reference's integer arithmetic EXACT as long as every partial sum stays below 2^53: the products and
// sums are integers, v_mfma_f64_16x16x4_f64 rounds nothing that fits 53 bits.  That is checked a posteriori on
// the diagonal (|sum of any subset of x_i y_i| <= max(G_ii, G_jj) by Cauchy-Schwarz, and a sum of squares
// that ever reached 2^53 stays >= 2^53 under rounding); if a diagonal entry is >= 2^53 the caller falls back
// to the wrapping-int64 kernel (matrix_super_kernel<2>), which is also the cross-check in the tests.
//
// Layout: profiles in blocks of 64; a workgroup (4 waves) takes one block pair (I >= J) and a stride of
// 64-bin slabs.  A slab of 64 profiles x 64 bins is loaded with 512-byte runs per profile, converted to
// fp64 and staged in LDS (rows padded to 66 so that the 16 lanes x 4 columns of a fragment read hit distinct
// banks).  Fragment of profile group g (16 profiles) for bin quad q: lane l holds X[16 g + l % 16][4 q + l / 16]
// -- the A and the B operand layout of the instruction coincide, so one LDS read serves both.  Wave w takes
// bin quads w, w+4, w+8, w+12 of the slab and all tiles: 10 (diagonal block: gi >= gj) or 16 accumulators
// of 4 fp64 per lane.  C/D: lane l, register r holds row (l >> 4) + 4 r, column l & 15.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vec_kernels.hpp"

namespace kpal {

constexpr int kGramBins = 64;
constexpr int kGramRow = 66;
typedef double gram_v4f64 __attribute__((ext_vector_type(4)));

template <bool DIAG>
__global__ __launch_bounds__(256) void gram_mfma_kernel(const int64_t *__restrict__ prof, int P, uint64_t n,
                                                        const int2 *__restrict__ blocks, Partial *__restrict__ partials)
{
    constexpr int NS = DIAG ? 1 : 2;               // slabs per stage (row block, column block)
    constexpr int NT = DIAG ? 10 : 16;             // 16 x 16 tiles accumulated
    __shared__ double stage[2][NS * 64][kGramRow];
    const int I = blocks[blockIdx.y].x, J = blocks[blockIdx.y].y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // loader: value q of thread t is bin (t & 63) of staged row 4 q + (t >> 6): a wave reads one 512-byte run
    const int lrow = threadIdx.x >> 6, lcol = threadIdx.x & 63;
    auto load_slab = [&](uint64_t c, int64_t (&v)[NS * 16]) {
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int p = (sidx == 0 ? I : J) * 64 + 4 * q + lrow;
                v[sidx * 16 + q] = p < P ? prof[(uint64_t)p * n + c * kGramBins + lcol] : 0;
            }
    };
    auto store_slab = [&](int buf, const int64_t (&v)[NS * 16]) {
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx)
#pragma unroll
            for (int q = 0; q < 16; ++q) stage[buf][sidx * 64 + 4 * q + lrow][lcol] = (double)v[sidx * 16 + q];
    };
    gram_v4f64 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = gram_v4f64{0.0, 0.0, 0.0, 0.0};
    const uint64_t slabs = n / kGramBins;
    uint64_t c = blockIdx.x;
    int64_t next[NS * 16];
    if (c < slabs) {
        load_slab(c, next);
        store_slab(0, next);
    }
    __syncthreads();
    int cur = 0;
    for (; c < slabs; c += gridDim.x) {
        const bool more = c + gridDim.x < slabs;   // block-uniform
        if (more) load_slab(c + gridDim.x, next);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = wave + 4 * u;
            double a[4], b[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                a[g] = stage[cur][16 * g + (lane & 15)][4 * q + (lane >> 4)];
                b[g] = DIAG ? a[g] : stage[cur][64 + 16 * g + (lane & 15)][4 * q + (lane >> 4)];
            }
            int t = 0;
#pragma unroll
            for (int gi = 0; gi < 4; ++gi)
#pragma unroll
                for (int gj = 0; gj < 4; ++gj) {
                    if (DIAG && gj > gi) continue;
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[gi], b[gj], acc[t], 0, 0, 0);
                    ++t;
                }
        }
        if (more) store_slab(cur ^ 1, next);
        __syncthreads();
        cur ^= 1;
    }
    // sum the four waves' accumulators (fixed order) and write this workgroup's partial Gram block
    double *red = &stage[0][0][0];                 // 4 x 256 doubles
    int t = 0;
#pragma unroll
    for (int gi = 0; gi < 4; ++gi)
#pragma unroll
        for (int gj = 0; gj < 4; ++gj) {
            if (DIAG && gj > gi) continue;
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave * 256 + ((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[t][r];
            __syncthreads();
            const int e = threadIdx.x;             // element row * 16 + col of the tile
            const double sum = ((red[e] + red[256 + e]) + red[512 + e]) + red[768 + e];
            partials[(((uint64_t)blockIdx.y * 16 + (gi * 4 + gj)) * 256 + e) * gridDim.x + blockIdx.x] = Partial{sum, 0ULL};
            ++t;
        }
}

}  // namespace kpal
