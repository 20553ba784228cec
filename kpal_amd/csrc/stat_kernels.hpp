// Profile summaries, merge and shrink on the device.
//
//   S1  stats_kernel          one pass: wrapping int64 sum (Profile.total, kpal/klib.py:200-204), an exact
//                             128-bit sum (for the mean, which NumPy accumulates in float64 and so never
//                             wraps: klib.py:206-211), non-zero count (klib.py:193-198), min and max
//   S2  stats_var_kernel      sum of (x - mean)^2 in float64 (ndarray.std, klib.py:220-225)
//   S3  select_hist_kernel    one 8-bit digit of a most-significant-digit-first radix select: histogram of the
//                             digit over the elements that match the digits chosen so far (np.median,
//                             klib.py:213-218); the host picks the bin that holds the wanted rank
//   S4  select_next_kernel    smallest element above a value (the upper middle element when it differs)
//   S5  merge_kernel<M>       metrics.mergers sum / xor / int / nint (kpal/metrics.py:174-179; klib.py:269-283)
//   S6  shrink_kernel         sums of 4^factor consecutive counts, int64 wrap (klib.py:329-352)
//
// All are single-pass streaming kernels bound by HBM: 8 bytes read per bin (S5: 16 read + 8 written).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kpal {

struct StatPartial {
    uint64_t sum_lo;     // 128-bit two's complement sum, low word == the wrapping int64 sum
    int64_t sum_hi;
    uint64_t non_zero;
    int64_t mn, mx;
};

__device__ __forceinline__ void stat_combine(StatPartial &a, const StatPartial &b)
{
    const uint64_t lo = a.sum_lo + b.sum_lo;
    a.sum_hi += b.sum_hi + (lo < a.sum_lo ? 1 : 0);
    a.sum_lo = lo;
    a.non_zero += b.non_zero;
    a.mn = b.mn < a.mn ? b.mn : a.mn;
    a.mx = b.mx > a.mx ? b.mx : a.mx;
}

__device__ __forceinline__ uint64_t shfl_down_u64(uint64_t v, int d)
{
    const uint32_t lo = __shfl_down((uint32_t)v, d), hi = __shfl_down((uint32_t)(v >> 32), d);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ StatPartial stat_shfl_down(const StatPartial &p, int d)
{
    StatPartial r;
    r.sum_lo = shfl_down_u64(p.sum_lo, d);
    r.sum_hi = (int64_t)shfl_down_u64((uint64_t)p.sum_hi, d);
    r.non_zero = shfl_down_u64(p.non_zero, d);
    r.mn = (int64_t)shfl_down_u64((uint64_t)p.mn, d);
    r.mx = (int64_t)shfl_down_u64((uint64_t)p.mx, d);
    return r;
}

constexpr int kStatThreads = 256;

// partial[blockIdx.x] = summary of this workgroup's grid-stride share.
__global__ __launch_bounds__(kStatThreads) void stats_kernel(const int64_t *__restrict__ x, uint64_t n,
                                                             StatPartial *__restrict__ partial)
{
    __shared__ StatPartial wsum[kStatThreads / 64];
    StatPartial p = {0, 0, 0, INT64_MAX, INT64_MIN};
    const uint64_t stride = (uint64_t)gridDim.x * kStatThreads;
    for (uint64_t i = (uint64_t)blockIdx.x * kStatThreads + threadIdx.x; i < n; i += stride) {
        const int64_t v = x[i];
        const uint64_t lo = p.sum_lo + (uint64_t)v;
        p.sum_hi += (v >> 63) + (lo < p.sum_lo ? 1 : 0);   // sign extension of v plus the carry
        p.sum_lo = lo;
        p.non_zero += v != 0;
        p.mn = v < p.mn ? v : p.mn;
        p.mx = v > p.mx ? v : p.mx;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const StatPartial o = stat_shfl_down(p, d);
        stat_combine(p, o);
    }
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = p;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kStatThreads / 64; ++w) stat_combine(p, wsum[w]);
        partial[blockIdx.x] = p;
    }
}

// partial[blockIdx.x] = sum over the workgroup's share of ((double)x - mean)^2, fixed order.
__global__ __launch_bounds__(kStatThreads) void stats_var_kernel(const int64_t *__restrict__ x, uint64_t n, double mean,
                                                                 double *__restrict__ partial)
{
    __shared__ double wsum[kStatThreads / 64];
    double s = 0.0;
    const uint64_t stride = (uint64_t)gridDim.x * kStatThreads;
    for (uint64_t i = (uint64_t)blockIdx.x * kStatThreads + threadIdx.x; i < n; i += stride) {
        const double d = (double)x[i] - mean;
        s += d * d;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kStatThreads / 64; ++w) s += wsum[w];
        partial[blockIdx.x] = s;
    }
}

// Order-preserving map int64 -> uint64.
__device__ __host__ __forceinline__ uint64_t select_key(int64_t v) { return (uint64_t)v ^ 0x8000000000000000ULL; }

// hist[d] += number of elements whose key matches `prefix` on the bits of `mask` and has digit d at
// `shift`.  Counts are typically concentrated in one or two digits, so the wave first counts the
// digit of its first lane with a ballot; only the other lanes use one LDS atomic each.
__global__ __launch_bounds__(kStatThreads) void select_hist_kernel(const int64_t *__restrict__ x, uint64_t n, uint64_t mask,
                                                                   uint64_t prefix, int shift,
                                                                   unsigned long long *__restrict__ hist)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * kStatThreads;
    const uint64_t rounds = (n + stride - 1) / stride;   // every lane runs every round: the ballots need whole waves
    for (uint64_t r = 0; r < rounds; ++r) {
        const uint64_t i = r * stride + (uint64_t)blockIdx.x * kStatThreads + threadIdx.x;
        uint32_t d = 0xFFFFFFFFu;
        if (i < n) {
            const uint64_t key = select_key(x[i]);
            if ((key & mask) == prefix) d = (uint32_t)(key >> shift) & 255u;
        }
        const uint32_t hot = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
        const bool eq = d == hot && d != 0xFFFFFFFFu;
        const uint32_t same = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(eq));
        if (same && (threadIdx.x & 63) == 0) atomicAdd(&h[hot], same);
        if (!eq && d != 0xFFFFFFFFu) atomicAdd(&h[d], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// partial[blockIdx.x] = smallest key above `key0` in the workgroup's share (UINT64_MAX if none).
__global__ __launch_bounds__(kStatThreads) void select_next_kernel(const int64_t *__restrict__ x, uint64_t n, uint64_t key0,
                                                                   unsigned long long *__restrict__ partial)
{
    __shared__ uint64_t wmin[kStatThreads / 64];
    uint64_t m = UINT64_MAX;
    const uint64_t stride = (uint64_t)gridDim.x * kStatThreads;
    for (uint64_t i = (uint64_t)blockIdx.x * kStatThreads + threadIdx.x; i < n; i += stride) {
        const uint64_t key = select_key(x[i]);
        if (key > key0 && key < m) m = key;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint64_t o = shfl_down_u64(m, d);
        m = o < m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) wmin[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kStatThreads / 64; ++w) m = wmin[w] < m ? wmin[w] : m;
        partial[blockIdx.x] = m;
    }
}

// metrics.mergers (kpal/metrics.py:174-179) on int64 vectors; products with a boolean are selections.
template <int MERGER>
__global__ __launch_bounds__(256) void merge_kernel(const int64_t *__restrict__ x, const int64_t *__restrict__ y, uint64_t n,
                                                    int64_t *__restrict__ out)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const uint64_t a = (uint64_t)x[i], b = (uint64_t)y[i];
        uint64_t r;
        if (MERGER == 0) r = a + b;                               // sum
        else if (MERGER == 1) r = ((a != 0) != (b != 0)) ? a + b : 0;   // xor: (x + y) * logical_xor(x, y)
        else if (MERGER == 2) r = b != 0 ? a : 0;                 // int:  x * bool(y)
        else r = b == 0 ? a : 0;                                  // nint: x * logical_not(y)
        out[i] = (int64_t)r;
    }
}

// out[j] = sum of in[j*m .. j*m + m), m = 4^factor (int64 wrap).  Thread t adds the input pair
// (2t, 2t+1) with one 16-byte load; for m <= 128 the m/2 lanes of an output combine with shuffles,
// for larger m a wave walks its output's inputs.
__global__ __launch_bounds__(256) void shrink_small_kernel(const int64_t *__restrict__ in, uint64_t n_pairs, int lanes_per_out,
                                                           int64_t *__restrict__ out)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    const uint64_t rounds = (n_pairs + stride - 1) / stride;   // whole waves take part in the shuffles
    const longlong2 *in2 = reinterpret_cast<const longlong2 *>(in);
    for (uint64_t r = 0; r < rounds; ++r) {
        const uint64_t t = r * stride + (uint64_t)blockIdx.x * 256 + threadIdx.x;
        uint64_t s = 0;
        if (t < n_pairs) {
            const longlong2 v = in2[t];
            s = (uint64_t)v.x + (uint64_t)v.y;
        }
        for (int d = lanes_per_out >> 1; d >= 1; d >>= 1) s += shfl_down_u64(s, d);
        if (t < n_pairs && (t & (uint64_t)(lanes_per_out - 1)) == 0) out[t / (uint64_t)lanes_per_out] = (int64_t)s;
    }
}

__global__ __launch_bounds__(256) void shrink_large_kernel(const int64_t *__restrict__ in, uint64_t n_out, uint64_t m,
                                                           int64_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const uint64_t waves = (uint64_t)gridDim.x * 4;
    for (uint64_t j = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); j < n_out; j += waves) {
        uint64_t s = 0;
        for (uint64_t i = lane; i < m; i += 64) s += (uint64_t)in[j * m + i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s += shfl_down_u64(s, d);
        if (lane == 0) out[j] = (int64_t)s;
    }
}

}  // namespace kpal
