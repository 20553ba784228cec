// option_kernels.hpp -- the optional steps of ProfileDistance.distance on the device (gfx950):
// positive (kpal/kdistlib.py:143-145), dynamic smoothing (kdistlib.py:53-124), scaling
// (kdistlib.py:149-157) and the final multiset / euclidean / cosine reduction over integer or
// scaled (float64) profiles (kdistlib.py:159-161, kpal/metrics.py:101-147).
//
// Dynamic smoothing is a recursion over base-4 prefixes in the reference: a node (start, length)
// is collapsed -- counts[start] = node sum, the rest zeroed -- iff
// min(f(quarter sums of left), f(quarter sums of right)) <= threshold, else its four quarters are
// visited.  Whether a node collapses depends only on the ORIGINAL counts below it (a node is only
// visited while none of its ancestors collapsed, and then nothing below it has been touched), so
// the recursion becomes: (1) bottom-up, level by level, the node sums and the collapse decision of
// every node; (2) per bin, the topmost deciding ancestor wins.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vec_kernels.hpp"

namespace kpal {

constexpr int kSummaryMin = 0, kSummaryAverage = 1, kSummaryMedian = 2;

// left' = left * bool(right); right' = right * bool(left')  ==  both zero unless both non-zero.
__global__ __launch_bounds__(256) void positive_kernel(const int64_t *__restrict__ l, const int64_t *__restrict__ r,
                                                       int64_t *__restrict__ lo, int64_t *__restrict__ ro, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const int64_t x = l[i], y = r[i];
        const bool both = x != 0 && y != 0;
        lo[i] = both ? x : 0;
        ro[i] = both ? y : 0;
    }
}

// Summary of four int64 quarter sums as NumPy evaluates it on an int64 array of length 4:
// np.min -> the integer; np.mean -> float64 sum of the converted values / 4; np.median -> mean of
// the two middle values.  Returned as double for the comparison with the threshold.
__device__ __forceinline__ double summarise4(const int64_t (&q)[4], int summary)
{
    if (summary == kSummaryMin) return (double)min(min(q[0], q[1]), min(q[2], q[3]));
    if (summary == kSummaryAverage) return ((((double)q[0] + (double)q[1]) + (double)q[2]) + (double)q[3]) / 4.0;
    // median: sort four values with a 5-comparator network, average the middle two
    int64_t a = min(q[0], q[1]), b = max(q[0], q[1]), c = min(q[2], q[3]), d = max(q[2], q[3]);
    const int64_t lo = max(a, c), hi = min(b, d);   // the two middle values are {max of mins, min of maxes}
    return ((double)min(lo, hi) + (double)max(lo, hi)) / 2.0;
}

// One level: node j of `nparent` nodes has the four children child[4j .. 4j+3] (the level below,
// or the counts themselves at the bottom).  Writes the node sums (wrapping int64, like
// ndarray.sum) and the collapse decision.
__global__ __launch_bounds__(256) void smooth_level_kernel(const int64_t *__restrict__ child_l,
                                                           const int64_t *__restrict__ child_r, uint64_t nparent,
                                                           int64_t *__restrict__ sum_l, int64_t *__restrict__ sum_r,
                                                           uint8_t *__restrict__ decide, int summary, double threshold)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nparent; j += (uint64_t)gridDim.x * blockDim.x) {
        const longlong2 *pl = reinterpret_cast<const longlong2 *>(child_l + 4 * j);
        const longlong2 *pr = reinterpret_cast<const longlong2 *>(child_r + 4 * j);
        const longlong2 a0 = pl[0], a1 = pl[1], b0 = pr[0], b1 = pr[1];
        const int64_t ql[4] = {a0.x, a0.y, a1.x, a1.y};
        const int64_t qr[4] = {b0.x, b0.y, b1.x, b1.y};
        sum_l[j] = (int64_t)((uint64_t)ql[0] + (uint64_t)ql[1] + (uint64_t)ql[2] + (uint64_t)ql[3]);
        sum_r[j] = (int64_t)((uint64_t)qr[0] + (uint64_t)qr[1] + (uint64_t)qr[2] + (uint64_t)qr[3]);
        const double f = fmin(summarise4(ql, summary), summarise4(qr, summary));
        decide[j] = f <= threshold ? 1 : 0;
    }
}

struct SmoothLevels {          // level d has 4^d nodes, d = 0 .. k-1
    const int64_t *sum_l[16];
    const int64_t *sum_r[16];
    const uint8_t *decide[16];
};

// Apply: one thread per bottom node (four bins).  Walk from the root; the first deciding ancestor
// (or the node itself) collapses everything below it into its first bin.
__global__ __launch_bounds__(256) void smooth_apply_kernel(const int64_t *__restrict__ l, const int64_t *__restrict__ r,
                                                           int k, SmoothLevels lv, int64_t *__restrict__ lo,
                                                           int64_t *__restrict__ ro)
{
    const uint64_t nnodes = 1ULL << (2 * (k - 1));
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nnodes; j += (uint64_t)gridDim.x * blockDim.x) {
        const longlong2 *pl = reinterpret_cast<const longlong2 *>(l + 4 * j);
        const longlong2 *pr = reinterpret_cast<const longlong2 *>(r + 4 * j);
        longlong2 a0 = pl[0], a1 = pl[1], b0 = pr[0], b1 = pr[1];
        for (int d = 0; d < k; ++d) {
            const int shift = 2 * (k - 1 - d);
            const uint64_t node = j >> shift;
            if (lv.decide[d][node]) {
                const bool first = (node << shift) == j;   // this thread holds the node's first bin
                a0 = make_longlong2(first ? lv.sum_l[d][node] : 0, 0);
                a1 = make_longlong2(0, 0);
                b0 = make_longlong2(first ? lv.sum_r[d][node] : 0, 0);
                b1 = make_longlong2(0, 0);
                break;
            }
        }
        longlong2 *ql = reinterpret_cast<longlong2 *>(lo + 4 * j);
        longlong2 *qr = reinterpret_cast<longlong2 *>(ro + 4 * j);
        ql[0] = a0;
        ql[1] = a1;
        qr[0] = b0;
        qr[1] = b1;
    }
}

// np.sum of both vectors (wrapping int64): partials[b] = left, partials[gridDim.x + b] = right.
__global__ __launch_bounds__(256) void totals_kernel(const int64_t *__restrict__ l, const int64_t *__restrict__ r,
                                                     uint64_t n, Partial *__restrict__ partials)
{
    Partial pl = {0.0, 0ULL}, pr = {0.0, 0ULL};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        pl.m += (uint64_t)l[i];
        pr.m += (uint64_t)r[i];
    }
    pl = block_reduce(pl);
    pr = block_reduce(pr);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = pl;
        partials[gridDim.x + blockIdx.x] = pr;
    }
}

// Final reduction.  METRIC 0/1: multiset prod/sum; 2: euclidean; 3: cosine.  SCALED: the values
// are float64 `count * scale` (kdistlib.py:156-157) and all arithmetic is float64; otherwise the
// int64 arithmetic of NumPy (wrapping).  Partial groups (each gridDim.x long):
//   multiset: [0] = (sum of terms, m)           euclidean: [0] = (float dot, int dot) of l - r
//   cosine:   [0] = l.r, [1] = l.l, [2] = r.r   as (float dot, int dot)
template <int METRIC, bool SCALED>
__global__ __launch_bounds__(256) void option_distance_kernel(const int64_t *__restrict__ l, const int64_t *__restrict__ r,
                                                              uint64_t n, double ls, double rs,
                                                              Partial *__restrict__ partials)
{
    Partial p0 = {0.0, 0ULL}, p1 = {0.0, 0ULL}, p2 = {0.0, 0ULL};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const int64_t xi = l[i], yi = r[i];
        if constexpr (SCALED) {
            const double x = (double)xi * ls, y = (double)yi * rs;
            if constexpr (METRIC <= 1) {
                if (x != 0.0 || y != 0.0) {
                    p0.s += METRIC == 0 ? pw_prod(x, y) : pw_sum(x, y);
                    p0.m += 1;
                }
            } else if constexpr (METRIC == 2) {
                const double d = x - y;
                p0.s += d * d;
            } else {
                p0.s += x * y;
                p1.s += x * x;
                p2.s += y * y;
            }
        } else {
            if constexpr (METRIC <= 1) {
                if (xi != 0 || yi != 0) {
                    p0.s += METRIC == 0 ? pw_prod(xi, yi) : pw_sum(xi, yi);
                    p0.m += 1;
                }
            } else if constexpr (METRIC == 2) {
                const uint64_t d = (uint64_t)xi - (uint64_t)yi;
                p0.m += d * d;
            } else {
                p0.m += (uint64_t)xi * (uint64_t)yi;
                p1.m += (uint64_t)xi * (uint64_t)xi;
                p2.m += (uint64_t)yi * (uint64_t)yi;
            }
        }
    }
    p0 = block_reduce(p0);
    if (threadIdx.x == 0) partials[blockIdx.x] = p0;
    if constexpr (METRIC == 3) {
        p1 = block_reduce(p1);
        p2 = block_reduce(p2);
        if (threadIdx.x == 0) {
            partials[gridDim.x + blockIdx.x] = p1;
            partials[2 * gridDim.x + blockIdx.x] = p2;
        }
    }
}

}  // namespace kpal
