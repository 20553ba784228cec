// vec_kernels.hpp -- streaming kernels over 4^k int64 count vectors (gfx950).
//   balance            Profile.balance            kpal/klib.py:285-298
//   split              Profile.split              kpal/klib.py:300-327
//   strand balance     kmer.get_balance score     kpal/kmer.py:243-245
//   pair distance      metrics.multiset/euclidean kpal/metrics.py:101-135
//   distance matrix    kdistlib.distance_matrix   kpal/kdistlib.py:179-186
// All are HBM-bandwidth kernels except the matrix, which is fp64-VALU bound (register tiles fed from
// LDS-staged 16 x 16 super-tiles; a 6-instruction division for the operands of real profiles).
// fp64 sums are reduced in a FIXED order (per-thread serial, wave shuffle tree, block tree,
// then a single-workgroup pass over the per-block partials) so results are run-to-run
// reproducible; they agree with NumPy's pairwise summation to ~1e-15 relative.
#pragma once
#include "kpal_device.hpp"

namespace kpal {

// ---- pairwise functions, kpal/metrics.py:159-162, int64 wrap-around like NumPy -------------
__device__ __forceinline__ int64_t wrap_abs_diff(int64_t x, int64_t y)
{
    const uint64_t d = (uint64_t)x - (uint64_t)y;
    return (int64_t)d < 0 ? (int64_t)(0ULL - d) : (int64_t)d;
}
__device__ __forceinline__ double pw_prod(int64_t x, int64_t y)
{
    const int64_t den = (int64_t)(((uint64_t)x + 1ULL) * ((uint64_t)y + 1ULL));
    return (double)wrap_abs_diff(x, y) / (double)den;
}
__device__ __forceinline__ double pw_sum(int64_t x, int64_t y)
{
    const int64_t den = (int64_t)((uint64_t)x + (uint64_t)y + 1ULL);
    return (double)wrap_abs_diff(x, y) / (double)den;
}
__device__ __forceinline__ double pw_prod(double x, double y) { return fabs(x - y) / ((x + 1.0) * (y + 1.0)); }
__device__ __forceinline__ double pw_sum(double x, double y) { return fabs(x - y) / (x + y + 1.0); }

struct Partial {
    double s;            // sum of pairwise terms
    unsigned long long m;  // multiset: bins with l!=0 or r!=0; euclidean: wrapping int64 dot
};

__device__ __forceinline__ Partial block_reduce(Partial p)
{
    __shared__ double sh_s[16];
    __shared__ unsigned long long sh_m[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        p.s += __shfl_down(p.s, d);
        p.m += __shfl_down(p.m, d);
    }
    __syncthreads();
    if (lane == 0) {
        sh_s[wave] = p.s;
        sh_m[wave] = p.m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int w = 1; w < nw; ++w) {
            p.s += sh_s[w];
            p.m += sh_m[w];
        }
    }
    return p;  // valid in thread 0
}

// ---- balance ------------------------------------------------------------------------------
// out[i] = in[i] + in[rc(i)] (i == rc(i) gives 2*in[i], klib.py:297-298).  Out of place.
__global__ __launch_bounds__(256) void balance_oop_kernel(const int64_t *__restrict__ in, int64_t *__restrict__ out,
                                                          int k, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = (int64_t)((uint64_t)in[i] + (uint64_t)in[revcomp(i, k)]);
}

// In place: the thread owning i < rc(i) updates both ends of the pair (klib.py:290-296).
__global__ __launch_bounds__(256) void balance_inplace_kernel(int64_t *__restrict__ c, int k, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = revcomp(i, k);
        if (i < r) {
            const uint64_t v = (uint64_t)c[i] + (uint64_t)c[r];
            c[i] = (int64_t)v;
            c[r] = (int64_t)v;
        } else if (i == r) {
            c[i] = (int64_t)((uint64_t)c[i] * 2ULL);
        }
    }
}

// LDS-tiled balance for k >= 6.  Write i = (H, M, L) with H / L the top / bottom three digits
// and M the k-6 middle digits; then rc(i) = (rc(L), rc(M), rc(H)): the 64x64 tile {(H, M, L)} maps
// onto the tile of rc(M), transposed and with rows/columns permuted by the 3-digit reverse
// complement.  One workgroup owns the tile pair (M, rc(M)), M <= rc(M): it reads both tiles as 64
// runs of 512 B, keeps them in LDS (rows padded to 65 to spread banks on the transposed read),
// and writes out[i] = in[i] + in[rc(i)] for both tiles -- 16 B of HBM traffic per bin instead of
// scattered 8-byte partner accesses.  in == out is allowed (all reads precede the barrier).
// The tile pairs come from a LIST (canon[c] = M of the c-th canonical pair, built by the host: kpal_vec.hip, canon_tiles) and the
// persistent workgroups take them round robin -- every workgroup gets the same number of pairs to within one.  (Striding through
// M itself and skipping the non-canonical ones left the work badly spread: a workgroup's M share their low digits, and those
// decide whether M <= rc(M) for nearly all of them -- a quarter of the workgroups had eight pairs, a quarter none.)
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8))) void balance_tiled_kernel(const int64_t *in, int64_t *out, int k,
                                                                                                     const uint32_t *__restrict__ canon, uint32_t ncanon)
{
    constexpr int T = 3, S = 64;
    __shared__ unsigned long long A[S][S + 1];
    __shared__ unsigned long long B[S][S + 1];
    const int md = k - 2 * T;
    const uint64_t nM = 1ULL << (2 * md);
    const uint64_t rowstride = 1ULL << (2 * (k - T));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int rl = (int)revcomp((uint64_t)lane, T);
    const unsigned long long *uin = reinterpret_cast<const unsigned long long *>(in);
    unsigned long long *uout = reinterpret_cast<unsigned long long *>(out);
    // persistent workgroups over the canonical tile pairs (M <= rc(M)); the next pair's eight values
    // per thread are loaded before the current pair is exchanged through LDS and written back.
    // The ORDER of the list (k >= 13) is page-aware: see canon_tiles.
    (void)nM;
    auto fetch = [&](uint64_t M, unsigned long long (&a)[4], unsigned long long (&b)[4]) {
        const uint64_t Mr = md > 0 ? revcomp(M, md) : 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint64_t row = (uint64_t)(w + 16 * q) * rowstride;
            a[q] = uin[row + M * S + lane];
            b[q] = uin[row + Mr * S + lane];
        }
    };
    unsigned long long a[4], b[4], na[4], nb[4];
    uint32_t seq = blockIdx.x;
    if (seq < ncanon) fetch(canon[seq], a, b);
    while (seq < ncanon) {
        const uint64_t M = canon[seq];
        const uint32_t seqn = seq + gridDim.x;
        if (seqn < ncanon) fetch(canon[seqn], na, nb);
        const uint64_t Mr = md > 0 ? revcomp(M, md) : 0;
        const bool self = M == Mr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            A[w + 16 * q][lane] = a[q];
            B[w + 16 * q][lane] = b[q];   // self: B == A
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rh = (int)revcomp((uint64_t)(w + 16 * q), T);
            const uint64_t row = (uint64_t)(w + 16 * q) * rowstride;
            uout[row + M * S + lane] = a[q] + B[rl][rh];
            if (!self) uout[row + Mr * S + lane] = b[q] + A[rl][rh];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a[q] = na[q];
            b[q] = nb[q];
        }
        seq = seqn;
    }
}

// ---- split --------------------------------------------------------------------------------
// Order-preserving compaction of i <= rc(i).  Pass 1 counts canonical indices per block-sized
// segment; the host scans the (small) count array; pass 2 writes.
constexpr int kSplitSeg = 4096;  // indices per block

__global__ __launch_bounds__(256) void split_count_kernel(int k, uint64_t n, uint32_t *__restrict__ seg_count)
{
    const uint64_t base = (uint64_t)blockIdx.x * kSplitSeg;
    uint32_t c = 0;
    for (int t = threadIdx.x; t < kSplitSeg; t += blockDim.x) {
        const uint64_t i = base + t;
        if (i < n && i <= revcomp(i, k)) ++c;
    }
    __shared__ uint32_t sh[4];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) seg_count[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void split_write_kernel(const int64_t *__restrict__ c, int k, uint64_t n,
                                                          const uint64_t *__restrict__ seg_offset,
                                                          int64_t *__restrict__ fwd, int64_t *__restrict__ rev)
{
    __shared__ uint32_t wave_tot[4];
    const uint64_t base = (uint64_t)blockIdx.x * kSplitSeg;
    uint64_t out = seg_offset[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t0 = 0; t0 < kSplitSeg; t0 += 256) {
        const uint64_t i = base + t0 + threadIdx.x;
        uint64_t r = 0;
        bool keep = false;
        if (i < n) {
            r = revcomp(i, k);
            keep = i <= r;
        }
        const unsigned long long bal = __ballot(keep);
        const uint32_t before = __popcll(bal & ((1ULL << lane) - 1ULL));
        if (lane == 0) wave_tot[wave] = __popcll(bal);
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) wbase += wave_tot[w];
            tot += wave_tot[w];
        }
        if (keep) {
            const uint64_t o = out + wbase + before;
            if (i < r) {
                fwd[o] = (int64_t)((uint64_t)c[i] * 2ULL);   // klib.py:319-320
                rev[o] = (int64_t)((uint64_t)c[r] * 2ULL);
            } else {
                fwd[o] = c[i];                               // klib.py:322-323
                rev[o] = c[i];
            }
        }
        out += tot;
        __syncthreads();
    }
}

// ---- strand balance: multiset(*split()) fused ------------------------------------------------
template <int PW>
__global__ __launch_bounds__(256) void strand_balance_kernel(const int64_t *__restrict__ c, int k, uint64_t n,
                                                             Partial *__restrict__ partials)
{
    Partial p = {0.0, 0ULL};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = revcomp(i, k);
        if (i > r) continue;
        int64_t f, v;
        if (i < r) {
            f = (int64_t)((uint64_t)c[i] * 2ULL);
            v = (int64_t)((uint64_t)c[r] * 2ULL);
        } else {
            f = v = c[i];
        }
        if (f != 0 || v != 0) {
            p.s += PW == 0 ? pw_prod(f, v) : pw_sum(f, v);
            p.m += 1;
        }
    }
    p = block_reduce(p);
    if (threadIdx.x == 0) partials[blockIdx.x] = p;
}

// ---- pair distance --------------------------------------------------------------------------
// METRIC 0/1: multiset prod/sum (metrics.py:121-123); 2: euclidean (int64 dot, metrics.py:135,46).
template <int METRIC, typename T>
__global__ __launch_bounds__(256) void pair_distance_kernel(const T *__restrict__ l, const T *__restrict__ r,
                                                            uint64_t n, Partial *__restrict__ partials)
{
    Partial p = {0.0, 0ULL};
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    // two elements per 16-byte load
    const uint64_t n2 = n >> 1;
    using V2 = typename std::conditional<std::is_same<T, double>::value, double2, longlong2>::type;
    const V2 *l2 = reinterpret_cast<const V2 *>(l);
    const V2 *r2 = reinterpret_cast<const V2 *>(r);
    auto term = [&](T x, T y) {
        if constexpr (METRIC == 2) {
            const uint64_t d = (uint64_t)x - (uint64_t)y;
            p.m += d * d;
        } else {
            if (x != 0 || y != 0) {
                p.s += METRIC == 0 ? pw_prod(x, y) : pw_sum(x, y);
                p.m += 1;
            }
        }
    };
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        const V2 a = l2[i], b = r2[i];
        term((T)a.x, (T)b.x);
        term((T)a.y, (T)b.y);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) term(l[n - 1], r[n - 1]);
    p = block_reduce(p);
    if (threadIdx.x == 0) partials[blockIdx.x] = p;
}

// ---- fused balance + distance, fused split + multiset (k >= 6) ----------------------------------
// Same tiling as balance_tiled_kernel: the workgroup of tile pair (M, rc(M)) forms the balanced
// values x = l[i] + l[rc(i)], y = r[i] + r[rc(i)] on the fly (kpal/kdistlib.py:139-141) and
// reduces the metric over the 2 x 4096 bins of the pair -- 16 B of HBM traffic per bin, no balanced
// copies.  The two LDS tiles (66 KiB: two workgroups per CU) are used twice: first for the left
// profile, whose balanced values stay in registers (8 per thread), then for the right profile,
// whose global loads are already in flight while the left one is transposed.
// PREFETCH: persistent workgroups, the next pair's values requested before the current pair is worked on -- ~120 registers, so
// ONE 1024-thread workgroup per CU (the launcher sizes the grid for that).  Without (KPAL_PDB_PREFETCH=0, A/B only): one pair at
// a time; forced into 64 registers for two workgroups per CU it spills and ran at half the rate (k = 12: 0.126 vs 0.059 ms).
template <int METRIC, bool PREFETCH>
__device__ __forceinline__ void pair_distance_balanced_body(const int64_t *__restrict__ l, const int64_t *__restrict__ r, int k,
                                                            const uint32_t *__restrict__ canon, uint32_t ncanon,
                                                            Partial *__restrict__ partials, unsigned long long (*A)[65], unsigned long long (*B)[65])
{
    constexpr int T = 3, S = 64;
    const int md = k - 2 * T;
    const uint64_t nM = 1ULL << (2 * md);
    const uint64_t rowstride = 1ULL << (2 * (k - T));
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (wave-uniform: row addresses are scalar)
    const int rl = (int)revcomp((uint64_t)lane, T);
    const unsigned long long *ul = reinterpret_cast<const unsigned long long *>(l);
    const unsigned long long *ur = reinterpret_cast<const unsigned long long *>(r);
    Partial p = {0.0, 0ULL};
    // persistent workgroups over the list of canonical tile pairs (M <= rc(M): balance_tiled_kernel), round robin; the next
    // pair's 16 values per thread are loaded before the current pair is transposed and reduced
    (void)nM;
    auto fetch = [&](uint64_t M, unsigned long long (&la)[4], unsigned long long (&lb)[4], unsigned long long (&ra)[4],
                     unsigned long long (&rb)[4]) {
        // (everything but the lane is wave-uniform: said explicitly, the loads take a scalar base + the lane's offset)
        const uint64_t Mu = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)M);
        const uint64_t Mr = md > 0 ? (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)revcomp(Mu, md)) : 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint64_t row = (uint64_t)(w + 16 * q) * rowstride;
            const unsigned long long *pla = ul + (row + Mu * S), *plb = ul + (row + Mr * S);
            const unsigned long long *pra = ur + (row + Mu * S), *prb = ur + (row + Mr * S);
            la[q] = pla[lane];
            lb[q] = plb[lane];
            ra[q] = pra[lane];
            rb[q] = prb[lane];
        }
    };
    auto term = [&](unsigned long long xu, unsigned long long yu) {
        const int64_t x = (int64_t)xu, y = (int64_t)yu;
        if constexpr (METRIC == 2) {
            const uint64_t d = (uint64_t)x - (uint64_t)y;
            p.m += d * d;
        } else {
            if (x != 0 || y != 0) {
                p.s += METRIC == 0 ? pw_prod(x, y) : pw_sum(x, y);
                p.m += 1;
            }
        }
    };
    unsigned long long la[4], lb[4], ra[4], rb[4], nla[4], nlb[4], nra[4], nrb[4];
    uint32_t seq = blockIdx.x;
    if (PREFETCH && seq < ncanon) fetch(canon[seq], la, lb, ra, rb);
    while (seq < ncanon) {
        const uint64_t M = canon[seq];
        const uint32_t seqn = seq + gridDim.x;
        if constexpr (PREFETCH) {
            if (seqn < ncanon) fetch(canon[seqn], nla, nlb, nra, nrb);
        } else {
            fetch(M, la, lb, ra, rb);
        }
        const bool self = md == 0 || M == revcomp(M, md);
        auto exchange = [&](unsigned long long (&a)[4], unsigned long long (&b)[4]) {
            // a/b: this thread's bins of tile M / rc(M); on return each holds bin + bin[rc]
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                A[w + 16 * q][lane] = a[q];
                B[w + 16 * q][lane] = b[q];   // self: B == A
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rh = (int)revcomp((uint64_t)(w + 16 * q), T);
                a[q] += B[rl][rh];
                b[q] += A[rl][rh];
            }
            __syncthreads();
        };
        exchange(la, lb);
        exchange(ra, rb);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            term(la[q], ra[q]);
            if constexpr (!PREFETCH) __builtin_amdgcn_sched_barrier(0);   // (64 registers: one division's temporaries at a time)
            if (!self) term(lb[q], rb[q]);
            if constexpr (!PREFETCH) __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (PREFETCH) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                la[q] = nla[q];
                lb[q] = nlb[q];
                ra[q] = nra[q];
                rb[q] = nrb[q];
            }
        }
        seq = seqn;
    }
    p = block_reduce(p);
    if (threadIdx.x == 0) partials[blockIdx.x] = p;
}

template <int METRIC>
__global__ __launch_bounds__(1024) void pair_distance_balanced_kernel(const int64_t *__restrict__ l, const int64_t *__restrict__ r, int k,
                                                                      const uint32_t *__restrict__ canon, uint32_t ncanon,
                                                                      Partial *__restrict__ partials)
{
    __shared__ unsigned long long A[64][65], B[64][65];
    pair_distance_balanced_body<METRIC, true>(l, r, k, canon, ncanon, partials, A, B);
}

// kmer.get_balance score (kpal/kmer.py:243-245) with the split halves never materialised: every
// unordered pair {i, rc(i)} is visited once -- from the tile of the smaller M, or, inside a
// self-paired tile, from its smaller index (palindromes contribute f = r = c[i], klib.py:322-323).
template <int PW>
__global__ __launch_bounds__(1024) void strand_balance_tiled_kernel(const int64_t *__restrict__ c, int k,
                                                                    Partial *__restrict__ partials)
{
    constexpr int T = 3, S = 64;
    __shared__ unsigned long long A[S][S + 1], B[S][S + 1];
    const int md = k - 2 * T;
    const uint64_t M = blockIdx.x;
    const uint64_t Mr = md > 0 ? revcomp(M, md) : 0;
    Partial p = {0.0, 0ULL};
    if (M <= Mr) {
        const bool self = M == Mr;
        const uint64_t rowstride = 1ULL << (2 * (k - T));
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const unsigned long long *uc = reinterpret_cast<const unsigned long long *>(c);
        for (int h = w; h < S; h += 16) {
            A[h][lane] = uc[(uint64_t)h * rowstride + M * S + lane];
            if (!self) B[h][lane] = uc[(uint64_t)h * rowstride + Mr * S + lane];
        }
        __syncthreads();
        const int rl = (int)revcomp((uint64_t)lane, T);
        for (int h = w; h < S; h += 16) {
            const int rh = (int)revcomp((uint64_t)h, T);
            const unsigned long long mine = A[h][lane];
            const unsigned long long other = self ? A[rl][rh] : B[rl][rh];
            int64_t f, v;
            bool take = true;
            if (self) {
                const int i_loc = h * S + lane, r_loc = rl * S + rh;   // order inside the tile == global order
                take = i_loc <= r_loc;
                if (i_loc == r_loc) {
                    f = v = (int64_t)mine;
                } else {
                    f = (int64_t)(mine * 2ULL);
                    v = (int64_t)(other * 2ULL);
                }
            } else {
                f = (int64_t)(mine * 2ULL);
                v = (int64_t)(other * 2ULL);
            }
            if (take && (f != 0 || v != 0)) {
                p.s += PW == 0 ? pw_prod(f, v) : pw_sum(f, v);
                p.m += 1;
            }
        }
    }
    p = block_reduce(p);
    if (threadIdx.x == 0) partials[blockIdx.x] = p;
}

// Final fixed-order reduction of per-block partials: out[q] = sum over blocks of partials[q*nblocks + b].
__global__ __launch_bounds__(256) void reduce_partials_kernel(const Partial *__restrict__ partials, uint32_t nblocks,
                                                              Partial *__restrict__ out)
{
    const Partial *src = partials + (uint64_t)blockIdx.x * nblocks;
    Partial p = {0.0, 0ULL};
    for (uint32_t b = threadIdx.x; b < nblocks; b += blockDim.x) {
        p.s += src[b].s;
        p.m += src[b].m;
    }
    p = block_reduce(p);
    if (threadIdx.x == 0) out[blockIdx.x] = p;
}

// num / den for den in [1, 2^63) and num >= 0 (the float path of the matrix kernel: counts < 2^31, so no
// zero, infinite, NaN or denormal operands and no scaling): v_rcp_f64, one Newton step on the reciprocal,
// the product, and one residual correction of the quotient -- the correction multiplies the error of the
// quotient by the error of the reciprocal, so the result is within 1 ulp of the correctly rounded quotient
// whenever v_rcp_f64 is good to 14 bits.  6 full-rate instructions instead of the ~13 of the IEEE
// division sequence (v_div_scale x2, two Newton steps, v_div_fmas, v_div_fixup), 29.0 -> 21.4 ms for the
// 64-profile k=12 matrix with register tiles.  The parity contract for fp64 results is 1e-9 relative.
// KPAL_MATRIX_DIV: 0 = IEEE division, 1 = two Newton steps without the correction, 2 = two steps with it.
#ifndef KPAL_MATRIX_DIV
#define KPAL_MATRIX_DIV 3
#endif
__device__ __forceinline__ double div_counts(double num, double den)
{
#if KPAL_MATRIX_DIV == 0
    return num / den;
#else
    double r = __builtin_amdgcn_rcp(den);   // (an fp32 v_rcp_f32 seed is as exact and not faster: the reciprocal is not the limit)
    r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
#if KPAL_MATRIX_DIV != 3
    r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
#endif
    const double q = num * r;
#if KPAL_MATRIX_DIV == 1
    return q;
#else
    return __builtin_fma(__builtin_fma(-den, q, num), r, q);
#endif
#endif
}

// 1 / d for d in [1, 2^32]: v_rcp_f64 and two Newton steps (within 1 ulp; no zero, infinite, NaN or denormal operand)
__device__ __forceinline__ double rcp_counts(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    return __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
}

// ---- distance matrix --------------------------------------------------------------------------
// Lower triangle of P x P in TILE x TILE register tiles.  blockIdx.y = tile (ti >= tj),
// blockIdx.x strides over bins.  Each thread streams one bin at a time for the TILE row
// profiles and TILE column profiles (coalesced 512-B wave loads per profile), accumulating
// TILE^2 (sum, m) pairs in registers.  partial layout: [tile][entry a*TILE+b][blockIdx.x].
// The TILE x TILE terms of one bin: row values x[], column values y[]; s = fp64 sums, mf = number of
// multiset terms, m = exact int64 dots (euclidean).
// Term counts of the float path: per row a one word of four byte counters (column b in byte b) of the bins in
// which x[a] or y[b] is non-zero -- 19 instead of 48 instructions per bin for the 16 counts; the bytes are
// added to the 32-bit totals every 255 bins.
template <int TILE>
struct TermBytes {
    uint32_t packed[TILE];
    uint32_t bins;
};

template <int TILE>
__device__ __forceinline__ void term_bytes_flush(TermBytes<TILE> &tb, uint32_t (&mf)[TILE][TILE])
{
    static_assert(TILE == 4, "four byte counters per word");
#pragma unroll
    for (int a = 0; a < TILE; ++a) {
#pragma unroll
        for (int b = 0; b < TILE; ++b) mf[a][b] += (tb.packed[a] >> (8 * b)) & 255u;
        tb.packed[a] = 0u;
    }
    tb.bins = 0u;
}

template <int METRIC, int TILE>
__device__ __forceinline__ void matrix_accumulate(const int64_t (&x)[TILE], const int64_t (&y)[TILE], double (&s)[TILE][TILE],
                                                  unsigned long long (&m)[TILE][TILE], uint32_t (&mf)[TILE][TILE],
                                                  TermBytes<TILE> &tb)
{
    if constexpr (METRIC != 2) {
        // Counts below 2^31 (any real profile): |x-y|, (x+1)(y+1) and x+y+1 are exact in float64 or
        // round exactly like the int64 value NumPy converts, so the terms are bit-identical to the
        // int64 formulation -- with 8 cheap 32-bit conversions per bin instead of 32 64-bit ones.
        uint64_t any = 0;
#pragma unroll
        for (int a = 0; a < TILE; ++a) any |= (uint64_t)x[a] | (uint64_t)y[a];
        if (__all((any >> 31) == 0)) {   // wave-uniform
            double xd[TILE], yd[TILE];
#pragma unroll
            for (int a = 0; a < TILE; ++a) {
                xd[a] = (double)(uint32_t)x[a];
                yd[a] = (double)(uint32_t)y[a];
            }
            // branch-free: a pair of zeros contributes |0 - 0| / 1 = +0.0 to the sum and nothing to the count, so
            // the 16 division chains of a bin are independent straight-line code that the scheduler interleaves
#pragma unroll
            for (int a = 0; a < TILE; ++a)
#pragma unroll
                for (int b = 0; b < TILE; ++b) {
                    const double num = fabs(xd[a] - yd[b]);
                    const double den = METRIC == 0 ? (xd[a] + 1.0) * (yd[b] + 1.0) : xd[a] + yd[b] + 1.0;
                    s[a][b] += div_counts(num, den);
                }
            uint32_t ynz = 0u;   // byte b = 1 iff y[b] != 0
#pragma unroll
            for (int b = 0; b < TILE; ++b) ynz |= min((uint32_t)y[b], 1u) << (8 * b);
#pragma unroll
            for (int a = 0; a < TILE; ++a) tb.packed[a] += (uint32_t)x[a] != 0u ? 0x01010101u : ynz;
            if (++tb.bins == 255u) term_bytes_flush(tb, mf);   // wave-uniform
            return;
        }
    }
#pragma unroll
    for (int a = 0; a < TILE; ++a)
#pragma unroll
        for (int b = 0; b < TILE; ++b) {
            if constexpr (METRIC == 2) {
                const uint64_t d = (uint64_t)x[a] - (uint64_t)y[b];
                m[a][b] += d * d;
            } else {
                if (x[a] != 0 || y[b] != 0) {
                    s[a][b] += METRIC == 0 ? pw_prod(x[a], y[b]) : pw_sum(x[a], y[b]);
                    mf[a][b] += 1u;
                }
            }
        }
}

// Multiset 'prod' terms with the reciprocals 1 / (x + 1) of the staged values precomputed ONCE per value by the
// loader of matrix_super_kernel instead of one division per pair: |x - y| / ((x + 1)(y + 1)) = |x - y| * rx * ry --
// a subtraction, a multiplication and a fused multiply-add per term (3 fp64 issue slots instead of ~12).  Each
// factor is within 1 ulp, so a term is within ~2 ulp of the reference's quotient and the sum of the non-negative
// terms within ~5e-16 relative -- the contract for fp64 results is 1e-9 (metrics.py:101-123).  Counts >= 2^31
// anywhere in the wave's values take the int64 formulation (matrix_accumulate), like before.
template <int TILE>
__device__ __forceinline__ void matrix_accumulate_prod_rcp(const int64_t (&x)[TILE], const int64_t (&y)[TILE],
                                                           const double (&rx)[TILE], const double (&ry)[TILE],
                                                           double (&s)[TILE][TILE], unsigned long long (&m)[TILE][TILE],
                                                           uint32_t (&mf)[TILE][TILE], TermBytes<TILE> &tb)
{
    uint64_t any = 0;
#pragma unroll
    for (int a = 0; a < TILE; ++a) any |= (uint64_t)x[a] | (uint64_t)y[a];
    if (!__all((any >> 31) == 0)) {   // wave-uniform
        matrix_accumulate<0, TILE>(x, y, s, m, mf, tb);
        return;
    }
    double xd[TILE], yd[TILE];
#pragma unroll
    for (int a = 0; a < TILE; ++a) {
        xd[a] = (double)(uint32_t)x[a];
        yd[a] = (double)(uint32_t)y[a];
    }
#pragma unroll
    for (int a = 0; a < TILE; ++a)
#pragma unroll
        for (int b = 0; b < TILE; ++b) s[a][b] = __builtin_fma(fabs(xd[a] - yd[b]) * rx[a], ry[b], s[a][b]);
    uint32_t ynz = 0u;   // byte b = 1 iff y[b] != 0
#pragma unroll
    for (int b = 0; b < TILE; ++b) ynz |= min((uint32_t)y[b], 1u) << (8 * b);
#pragma unroll
    for (int a = 0; a < TILE; ++a) tb.packed[a] += (uint32_t)x[a] != 0u ? 0x01010101u : ynz;
    if (++tb.bins == 255u) term_bytes_flush(tb, mf);   // wave-uniform
}

template <int METRIC, int TILE>
__global__ __launch_bounds__(256) void matrix_tile_kernel(const int64_t *__restrict__ prof, int P, uint64_t n,
                                                          const int2 *__restrict__ tiles,
                                                          Partial *__restrict__ partials)
{
    const int ti = tiles[blockIdx.y].x, tj = tiles[blockIdx.y].y;
    double s[TILE][TILE];
    unsigned long long m[TILE][TILE];
    uint32_t mf[TILE][TILE];   // multiset: number of terms (a thread sees fewer than 2^32 bins); m holds the euclidean dots
#pragma unroll
    for (int a = 0; a < TILE; ++a)
#pragma unroll
        for (int b = 0; b < TILE; ++b) {
            s[a][b] = 0.0;
            m[a][b] = 0ULL;
            mf[a][b] = 0u;
        }
    TermBytes<TILE> tb = {{0u, 0u, 0u, 0u}, 0u};
    const int64_t *rowp[TILE];
    const int64_t *colp[TILE];
#pragma unroll
    for (int a = 0; a < TILE; ++a) {
        rowp[a] = prof + (uint64_t)min(ti * TILE + a, P - 1) * n;
        colp[a] = prof + (uint64_t)min(tj * TILE + a, P - 1) * n;
    }
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        int64_t x[TILE], y[TILE];
#pragma unroll
        for (int a = 0; a < TILE; ++a) {
            x[a] = rowp[a][i];
            y[a] = colp[a][i];
        }
        matrix_accumulate<METRIC, TILE>(x, y, s, m, mf, tb);
    }
    term_bytes_flush(tb, mf);
#pragma unroll
    for (int a = 0; a < TILE; ++a)
#pragma unroll
        for (int b = 0; b < TILE; ++b) {
            Partial p = {s[a][b], METRIC != 2 ? (unsigned long long)mf[a][b] : m[a][b]};
            p = block_reduce(p);
            if (threadIdx.x == 0)
                partials[((uint64_t)blockIdx.y * TILE * TILE + a * TILE + b) * gridDim.x + blockIdx.x] = p;
        }
}

// The same lower triangle in 16 x 16 SUPER-tiles staged through LDS (k >= 6, P > 8): a workgroup loads 64 bins
// of its 16 row and 16 column profiles once (512-byte runs, the next stage's loads in flight during the
// arithmetic) and its 16 groups of 16 lanes compute the sixteen 4 x 4 register tiles from LDS -- a quarter of
// the global loads per term of matrix_tile_kernel, whose 146 GB of (cached) loads bound the euclidean matrix
// and nearly bound the multiset one.  Group g owns tile (4 si + g/4, 4 sj + g%4); tiles above the diagonal
// idle.  Rows are padded to 68 bins so that the column rows of the two groups of a half-wave (4 rows apart)
// sit 32 banks apart for ds_read_b64.  Partials: the layout of matrix_tile_kernel, tile index ti(ti+1)/2 + tj.
constexpr int kSuperBins = 64;
constexpr int kSuperRow = 68;
template <int METRIC>
__global__ __launch_bounds__(256) void matrix_super_kernel(const int64_t *__restrict__ prof, int P, uint64_t n,
                                                           const int2 *__restrict__ supers,
                                                           Partial *__restrict__ partials)
{
    constexpr int TILE = 4;
    constexpr bool RCP = METRIC == 0;              // 'prod': reciprocals 1 / (x + 1) staged next to the values
    __shared__ int64_t stage[2][32][kSuperRow];
    __shared__ double rstage[RCP ? 2 : 1][RCP ? 32 : 1][RCP ? kSuperRow : 1];
    auto put = [&](int buf, int row, int col, int64_t v) {
        stage[buf][row][col] = v;
        if constexpr (RCP) rstage[buf][row][col] = rcp_counts((double)(uint32_t)v + 1.0);   // (unused when v >= 2^31)
    };
    const int si = supers[blockIdx.y].x, sj = supers[blockIdx.y].y;
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
    const int ti = si * 4 + (g >> 2), tj = sj * 4 + (g & 3);
    const int side = (P + TILE - 1) / TILE;
    const bool mine = ti < side && tj <= ti;           // this group's 4 x 4 tile is part of the lower triangle
    double s[TILE][TILE];
    unsigned long long m[TILE][TILE];
    uint32_t mf[TILE][TILE];
#pragma unroll
    for (int a = 0; a < TILE; ++a)
#pragma unroll
        for (int b = 0; b < TILE; ++b) {
            s[a][b] = 0.0;
            m[a][b] = 0ULL;
            mf[a][b] = 0u;
        }
    TermBytes<TILE> tb = {{0u, 0u, 0u, 0u}, 0u};
    // loader: value q of thread t is bin (t & 63) of staged row 4 q + (t >> 6): a wave reads one 512-byte run
    const int lrow = threadIdx.x >> 6, lcol = threadIdx.x & 63;
    const int64_t *src[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int r = 4 * q + lrow;                    // 0..15 rows of the super-tile, 16..31 its columns
        const int profile = r < 16 ? si * 16 + r : sj * 16 + (r - 16);
        src[q] = prof + (uint64_t)min(profile, P - 1) * n + lcol;
    }
    const uint64_t chunks = n / kSuperBins;
    int64_t next[8];
    uint64_t c = blockIdx.x;
    if (c < chunks) {
#pragma unroll
        for (int q = 0; q < 8; ++q) put(0, 4 * q + lrow, lcol, src[q][c * kSuperBins]);
    }
    __syncthreads();
    int cur = 0;
    for (; c < chunks; c += gridDim.x) {
        const bool more = c + gridDim.x < chunks;      // block-uniform
        if (more) {
#pragma unroll
            for (int q = 0; q < 8; ++q) next[q] = src[q][(c + gridDim.x) * kSuperBins];
        }
        if (mine) {
#pragma unroll 1   // (unrolled 2 / 4 times: 21.3 / 20.4 ms against 19.9)
            for (int u = 0; u < kSuperBins / 16; ++u) {
                int64_t x[TILE], y[TILE];
#pragma unroll
                for (int a = 0; a < TILE; ++a) {
                    x[a] = stage[cur][4 * (g >> 2) + a][16 * u + l];
                    y[a] = stage[cur][16 + 4 * (g & 3) + a][16 * u + l];
                }
                if constexpr (RCP) {
                    double rx[TILE], ry[TILE];
#pragma unroll
                    for (int a = 0; a < TILE; ++a) {
                        rx[a] = rstage[cur][4 * (g >> 2) + a][16 * u + l];
                        ry[a] = rstage[cur][16 + 4 * (g & 3) + a][16 * u + l];
                    }
                    matrix_accumulate_prod_rcp<TILE>(x, y, rx, ry, s, m, mf, tb);
                } else {
                    matrix_accumulate<METRIC, TILE>(x, y, s, m, mf, tb);
                }
            }
        }
        if (more) {
#pragma unroll
            for (int q = 0; q < 8; ++q) put(cur ^ 1, 4 * q + lrow, lcol, next[q]);
        }
        __syncthreads();
        cur ^= 1;
    }
    term_bytes_flush(tb, mf);
    // per-group reduction over its 16 lanes (fixed order), lane 0 of the group writes
#pragma unroll
    for (int a = 0; a < TILE; ++a)
#pragma unroll
        for (int b = 0; b < TILE; ++b) {
            double ps = s[a][b];
            unsigned long long pm = METRIC != 2 ? (unsigned long long)mf[a][b] : m[a][b];
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) {
                ps += __shfl_down(ps, d, 16);
                pm += __shfl_down(pm, d, 16);
            }
            if (mine && l == 0) {
                const uint64_t t = (uint64_t)ti * (ti + 1) / 2 + tj;
                partials[(t * TILE * TILE + a * TILE + b) * gridDim.x + blockIdx.x] = Partial{ps, pm};
            }
        }
}

// Wave priority by progress (quad_kernels.hpp: quad_tile_priority): the workgroups of the staged matrix kernels run a few thousand
// stages each, four to a CU, and the arbiter's oldest-first order let them finish one after the other -- the last one of a CU
// alone.  A workgroup's priority falls with the share of its stages it has done.
__device__ __forceinline__ void matrix_stage_priority(uint64_t done, uint64_t total)
{
#if !defined(KPAL_MATRIX_NO_PRIO)   // A/B builds
    switch ((uint32_t)(done * 4u / total)) {     // (block-uniform scalars)
    case 0: __builtin_amdgcn_s_setprio(3); break;
    case 1: __builtin_amdgcn_s_setprio(2); break;
    case 2: __builtin_amdgcn_s_setprio(1); break;
    default: __builtin_amdgcn_s_setprio(0); break;
    }
#endif
}

// Multiset with the 'prod' pairwise function (the default of kpal distance / matrix; metrics.py:101-123, 159-162) as a
// difference of reciprocals:
//        |x - y| / ((x + 1)(y + 1))  =  |(x + 1) - (y + 1)| / ((x + 1)(y + 1))  =  | 1/(y + 1) - 1/(x + 1) |.
// With r = 1 / (count + 1) staged instead of the counts a term is ONE subtraction and ONE add of an absolute value -- two
// fp64 instructions (matrix_super_kernel<0>: three, plus the conversions; the plain division: ~12) -- and the number of
// terms (bins where x != 0 or y != 0) leaves the fp64 loop entirely: the loader's waves read 64 bins of one profile at a
// time, so ONE ballot gives that row's zero mask, and the bins where BOTH profiles are zero are popcount(mask_i & mask_j),
// two v_bcnt per pair and stage, accumulated by thread (i, j) of the 16 x 16 super-tile.
//   Accuracy: r is within 1 ulp of 1 / (x + 1) (rcp_counts), so a term's error is at most 2^-52 (r_x + r_y) against a term
// of at least r_x r_y (x != y: |x - y| >= 1): relative 2^-52 (x + y + 2) -- below 2.9e-11 while both counts are below 2^16
// (kRdiffMaxCount; at 2^20 the bound would be 4.7e-10, half the contract with nothing left for the accumulation),
// and every term being non-negative that bounds the relative error of the sum as well; the contract for fp64 results is
// 1e-9 (typical: 1e-15; the cancellation-dominated worst case -- all counts just below the limit, differing by 1 -- is
// tests/test_gpu_vec.py::test_matrix_rdiff_worst_case).  A count >= 2^16 (or negative) anywhere raises *big and the caller
// reruns the pair-of-counts kernel.
//   Reciprocals of counts below 512 come from a table in LDS (one ds_read_b64 instead of v_rcp_f64 + four fused
// multiply-adds per staged value -- the loader would cost 60 % of the arithmetic otherwise); larger counts are computed.
constexpr unsigned long long kRdiffMaxCount = 1ull << 16;   // counts the difference form is accurate for (see above)
constexpr int kRdiffTable = 512;    // reciprocals 1 / (c + 1) of counts c < 512 (4 KiB: four workgroups per CU)
constexpr int kRdiffRow = 64;       // staged row: 64 bins, unpadded -- with 16-byte reads a 16-lane group covers all 64 banks, and
                                    // rows a multiple of 8 doubles apart keep the lanes of two groups that share a read pass apart
__global__ __launch_bounds__(256) void matrix_rdiff_kernel(const int64_t *__restrict__ prof, int P, uint64_t n,
                                                           const int2 *__restrict__ supers, uint32_t nsuper,
                                                           Partial *__restrict__ partials, uint32_t *__restrict__ big)
{
    constexpr int TILE = 4;
    __shared__ __attribute__((aligned(16))) double rstage[2][32][kRdiffRow];
    __shared__ unsigned long long zmask[2][32];
    __shared__ double rtable[kRdiffTable];
    for (int i = threadIdx.x; i < kRdiffTable; i += 256) rtable[i] = rcp_counts((double)i + 1.0);
    // Every profile is staged by several super-tiles (64 profiles: 10 super-tiles x 32 rows = 5 x the profiles' bytes, 43 GB at
    // k = 12 -- more than the arithmetic takes).  Workgroups are dispatched round-robin over the 8 XCDs, each with its own L2:
    // the 1-D grid is cut so that the `nsuper` workgroups that stage the SAME bins are neighbours on ONE XCD -- linear id
    // L = (c * nsuper + s) * 8 + x  ->  super-tile s, bin-group c * 8 + x -- and the second to tenth reader of a line hits that L2.
    const uint32_t lin = blockIdx.x, xcd = lin & 7u, sidx = (lin >> 3) % nsuper, cgrp = (lin >> 3) / nsuper;
    const uint32_t group = cgrp * 8u + xcd, ngroups = gridDim.x / nsuper;   // (the host launches nsuper * a multiple of 8 workgroups)
    const int si = supers[sidx].x, sj = supers[sidx].y;
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
    const int ti = si * 4 + (g >> 2), tj = sj * 4 + (g & 3);
    const int side = (P + TILE - 1) / TILE;
    const bool mine = ti < side && tj <= ti;           // this group's 4 x 4 tile is part of the lower triangle
    double s[TILE][TILE];
#pragma unroll
    for (int a = 0; a < TILE; ++a)
#pragma unroll
        for (int b = 0; b < TILE; ++b) s[a][b] = 0.0;
    uint32_t both_zero = 0;                            // pair (row threadIdx.x >> 4, column threadIdx.x & 15) of the super-tile
    bool saw_big = false;
    // loader: values 2 q', 2 q' + 1 of thread t are the bins 2 (t & 31), 2 (t & 31) + 1 of staged row 8 q' + (t >> 5): a wave reads
    // two 512-byte runs with 16-byte loads (as 8-byte loads the 43 GB of staged reads moved at 0.6 of the rate: MI355X_MICROARCH.md).
    // The row addresses of a wave's two halves are scalar; a lane selects its half's.
    const int lrow = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lhalf = (threadIdx.x >> 5) & 1, lcol = threadIdx.x & 31;
    const int64_t *src[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = 8 * q + 2 * lrow + h;            // 0..15 rows of the super-tile, 16..31 its columns
            const int profile = r < 16 ? si * 16 + r : sj * 16 + (r - 16);
            src[q][h] = prof + (uint64_t)min(profile, P - 1) * n;   // (uniform)
        }
    __syncthreads();                                   // the table
    auto recip = [&](int64_t v, bool all_small) -> double {
        if (all_small) return rtable[(uint32_t)v];
        saw_big |= (unsigned long long)v >= kRdiffMaxCount;
        return (unsigned long long)v < (unsigned long long)kRdiffTable ? rtable[(uint32_t)v & (kRdiffTable - 1)] : rcp_counts((double)(uint32_t)v + 1.0);
    };
    auto put = [&](int buf, int q, const longlong2 &v) {
        const int row = 8 * q + 2 * lrow + lhalf;
        const bool all_small = __all((unsigned long long)v.x < (unsigned long long)kRdiffTable && (unsigned long long)v.y < (unsigned long long)kRdiffTable);   // wave-uniform
        double2 r;
        r.x = recip(v.x, all_small);
        r.y = recip(v.y, all_small);
        *reinterpret_cast<double2 *>(&rstage[buf][row][2 * lcol]) = r;
        // zero masks: bit c = bin 2c, bit 32 + c = bin 2c + 1 (the same permutation of the bins in every row)
        const unsigned long long z0 = __builtin_amdgcn_ballot_w64(v.x == 0), z1 = __builtin_amdgcn_ballot_w64(v.y == 0);
        if (lcol == 0) zmask[buf][row] = lhalf ? ((z0 >> 32) | (z1 & 0xFFFFFFFF00000000ull)) : ((z0 & 0xFFFFFFFFull) | (z1 << 32));
    };
    const uint64_t chunks = n / kSuperBins;
    auto request = [&](longlong2 (&dst)[4], uint64_t chunk) {
        if (chunk < chunks) {                          // block-uniform
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t *p = lhalf ? src[q][1] : src[q][0];
                dst[q] = *reinterpret_cast<const longlong2 *>(p + chunk * kSuperBins + 2 * lcol);
            }
        }
    };
    auto compute = [&](int cur) {
        both_zero += (uint32_t)__popcll(zmask[cur][threadIdx.x >> 4] & zmask[cur][16 + (threadIdx.x & 15)]);
        if (mine) {
            // lane l takes the bin pairs (2l, 2l+1) and (32 + 2l, 32 + 2l + 1): 16-byte LDS reads (ds_read_b128 moves 256 B/clk per
            // CU; the ds_read2_b64 the 8-byte form compiled to, half of that -- and the LDS, not the fp64 pipe, set the pace)
#pragma unroll
            for (int u = 0; u < kSuperBins / 32; ++u) {
                double2 rx[TILE], ry[TILE];
#pragma unroll
                for (int a = 0; a < TILE; ++a) {
                    rx[a] = *reinterpret_cast<const double2 *>(&rstage[cur][4 * (g >> 2) + a][32 * u + 2 * l]);
                    ry[a] = *reinterpret_cast<const double2 *>(&rstage[cur][16 + 4 * (g & 3) + a][32 * u + 2 * l]);
                }
#pragma unroll
                for (int a = 0; a < TILE; ++a)
#pragma unroll
                    for (int b = 0; b < TILE; ++b) {
                        s[a][b] += fabs(rx[a].x - ry[b].x);
                        s[a][b] += fabs(rx[a].y - ry[b].y);
                    }
            }
        }
    };
    // the values of the next stage are requested before this stage's arithmetic and staged after it.  (Requesting TWO stages
    // ahead -- a stage's arithmetic takes ~0.3 us, a load 1-2 us -- needs 16 more registers than four waves per SIMD leave:
    // the compiler parked the prefetched values in scratch memory and the kernel was slower.  Round 4 tried the register-free
    // way to that depth: the raw counts by LDS-DMA (global_load_lds_dwordx4, inline assembly so that hipcc does not drain it
    // before every LDS read) into a three-slot ring, converted to reciprocals in place one iteration later, one raw s_barrier
    // per stage, 52 KiB of LDS = three workgroups per CU -- correct, and 8.4 ms against this kernel's 6.0: the DMA pieces cost
    // 100-185 cycles of issue each (MI355X_MICROARCH.md) -- four per wave and stage, as much as the stage's 136 fp64
    // instructions -- and the in-place pass adds a third to the LDS traffic, which already runs level with the fp64 pipe.)
    longlong2 next[4];
    uint64_t c = group;
    uint64_t stages = 0;
    request(next, c);
    if (c < chunks) {
#pragma unroll
        for (int q = 0; q < 4; ++q) put(0, q, next[q]);
    }
    __syncthreads();
    int cur = 0;
    const uint64_t my_stages = group < chunks ? (chunks - group + ngroups - 1) / ngroups : 1;
    for (; c < chunks; c += ngroups, ++stages) {
        const bool more = c + ngroups < chunks;        // block-uniform
        if ((stages & 15u) == 0) matrix_stage_priority(stages, my_stages);
        request(next, c + ngroups);
        compute(cur);
        if (more) {
#pragma unroll
            for (int q = 0; q < 4; ++q) put(cur ^ 1, q, next[q]);
        }
        __syncthreads();
        cur ^= 1;
    }
    if (saw_big) atomicOr(big, 1u);
    // sums: per-group reduction over its 16 lanes (fixed order), lane 0 of the group writes .s
#pragma unroll
    for (int a = 0; a < TILE; ++a)
#pragma unroll
        for (int b = 0; b < TILE; ++b) {
            double ps = s[a][b];
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) ps += __shfl_down(ps, d, 16);
            if (mine && l == 0) {
                const uint64_t t = (uint64_t)ti * (ti + 1) / 2 + tj;
                partials[(t * TILE * TILE + a * TILE + b) * ngroups + group].s = ps;
            }
        }
    // term counts: thread (i, j) of the super-tile writes .m = bins seen - bins where both are zero
    {
        const int i = si * 16 + (int)(threadIdx.x >> 4), j = sj * 16 + (int)(threadIdx.x & 15);
        const int pti = i / TILE, ptj = j / TILE;
        if (pti < side && ptj <= pti) {
            const uint64_t t = (uint64_t)pti * (pti + 1) / 2 + ptj;
            partials[(t * TILE * TILE + (i % TILE) * TILE + (j % TILE)) * ngroups + group].m = stages * kSuperBins - both_zero;
        }
    }
}

// Multiset with the 'sum' pairwise function, |x - y| / (x + y + 1) (metrics.py:101-123, 159-162): no difference form as for
// 'prod', but the denominator is a small integer -- its reciprocal comes from a table in LDS, R[s] = 1 / (s + 1) for
// s = x + y < kRsumTable, and a term is v_sad_u32 (|x - y|), v_add_lshl_u32 (the table offset), one ds_read_b64, a conversion
// and one fused multiply-add (matrix_super_kernel<1>: the two conversions, the sum and a ~10-instruction division).  Profiles of
// one sample have counts within a narrow range, so the 64 lanes of a table read touch a few dozen consecutive entries: few bank
// conflicts.  Same super-tile structure, loader, zero masks / popcounts for the term count and partial layout as
// matrix_rdiff_kernel; the staged values are the counts themselves as 32-bit integers (16 KiB instead of 32).
//   A count >= kRsumTable / 2 anywhere raises *big and the caller reruns matrix_super_kernel<1> (an inline second path for
// such stages cost the kernel its occupancy: 200 registers).
//   Accuracy: R within 1 ulp, |x - y| exact: a term within 1.5 ulp of the correctly rounded quotient the reference computes.
constexpr int kRsumTable = 2048;
__global__ __launch_bounds__(256) void matrix_rsum_kernel(const int64_t *__restrict__ prof, int P, uint64_t n,
                                                          const int2 *__restrict__ supers, uint32_t nsuper,
                                                          Partial *__restrict__ partials, uint32_t *__restrict__ big)
{
    constexpr int TILE = 4;
    __shared__ __attribute__((aligned(16))) uint32_t cstage[2][32][kSuperBins];
    __shared__ unsigned long long zmask[2][32];
    __shared__ double rtable[kRsumTable];
    for (int i = threadIdx.x; i < kRsumTable; i += 256) rtable[i] = rcp_counts((double)i + 1.0);
    // (grid: see matrix_rdiff_kernel -- the workgroups that stage the same bins are neighbours on one XCD)
    const uint32_t lin = blockIdx.x, xcd = lin & 7u, sidx = (lin >> 3) % nsuper, cgrp = (lin >> 3) / nsuper;
    const uint32_t group = cgrp * 8u + xcd, ngroups = gridDim.x / nsuper;
    const int si = supers[sidx].x, sj = supers[sidx].y;
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
    const int ti = si * 4 + (g >> 2), tj = sj * 4 + (g & 3);
    const int side = (P + TILE - 1) / TILE;
    const bool mine = ti < side && tj <= ti;
    double s[TILE][TILE];
#pragma unroll
    for (int a = 0; a < TILE; ++a)
#pragma unroll
        for (int b = 0; b < TILE; ++b) s[a][b] = 0.0;
    uint32_t both_zero = 0;
    bool saw_big = false;
    // (loader: see matrix_rdiff_kernel -- 16-byte loads, two rows per wave)
    const int lrow = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lhalf = (threadIdx.x >> 5) & 1, lcol = threadIdx.x & 31;
    const int64_t *src[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = 8 * q + 2 * lrow + h;
            const int profile = r < 16 ? si * 16 + r : sj * 16 + (r - 16);
            src[q][h] = prof + (uint64_t)min(profile, P - 1) * n;
        }
    __syncthreads();
    auto put = [&](int buf, int q, const longlong2 &v) {
        const int row = 8 * q + 2 * lrow + lhalf;
        saw_big |= (unsigned long long)v.x >= (unsigned long long)(kRsumTable / 2) || (unsigned long long)v.y >= (unsigned long long)(kRsumTable / 2);   // (x + y must stay inside the table)
        // (masked: a larger count only ever costs a rerun, never an out-of-range read)
        *reinterpret_cast<uint2 *>(&cstage[buf][row][2 * lcol]) =
            make_uint2((uint32_t)v.x & (uint32_t)(kRsumTable / 2 - 1), (uint32_t)v.y & (uint32_t)(kRsumTable / 2 - 1));
        const unsigned long long z0 = __builtin_amdgcn_ballot_w64(v.x == 0), z1 = __builtin_amdgcn_ballot_w64(v.y == 0);
        if (lcol == 0) zmask[buf][row] = lhalf ? ((z0 >> 32) | (z1 & 0xFFFFFFFF00000000ull)) : ((z0 & 0xFFFFFFFFull) | (z1 << 32));
    };
    const uint64_t chunks = n / kSuperBins;
    auto request = [&](longlong2 (&dst)[4], uint64_t chunk) {
        if (chunk < chunks) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t *p = lhalf ? src[q][1] : src[q][0];
                dst[q] = *reinterpret_cast<const longlong2 *>(p + chunk * kSuperBins + 2 * lcol);
            }
        }
    };
    auto compute = [&](int cur) {
        both_zero += (uint32_t)__popcll(zmask[cur][threadIdx.x >> 4] & zmask[cur][16 + (threadIdx.x & 15)]);
        if (!mine) return;
        // lane l takes the bins 4l .. 4l+3 of every row: one 16-byte LDS read per row
        uint4 cx[TILE], cy[TILE];
#pragma unroll
        for (int a = 0; a < TILE; ++a) {
            cx[a] = *reinterpret_cast<const uint4 *>(&cstage[cur][4 * (g >> 2) + a][4 * l]);
            cy[a] = *reinterpret_cast<const uint4 *>(&cstage[cur][16 + 4 * (g & 3) + a][4 * l]);
        }
        const char *tab = reinterpret_cast<const char *>(rtable);
#pragma unroll
        for (int a = 0; a < TILE; ++a)
#pragma unroll
            for (int b = 0; b < TILE; ++b) {
                // one pair (four terms) at a time: everything a term needs before its table read depends only on the staged counts,
                // and with the offsets and differences of all 64 terms computed up front the kernel needed 190 registers (two
                // waves per SIMD); the other three waves of the SIMD cover the latency of the four reads
                asm volatile("" : "+v"(cy[b].x), "+v"(cy[b].y), "+v"(cy[b].z), "+v"(cy[b].w));
                const uint32_t x[4] = {cx[a].x, cx[a].y, cx[a].z, cx[a].w}, y[4] = {cy[b].x, cy[b].y, cy[b].z, cy[b].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    uint32_t d;
                    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(x[e]), "v"(y[e]));   // |x - y|
                    const double r = *reinterpret_cast<const double *>(tab + ((x[e] + y[e]) << 3));
                    s[a][b] = fma((double)d, r, s[a][b]);
                }
            }
    };
    longlong2 next[4];
    uint64_t c = group;
    uint64_t stages = 0;
    request(next, c);
    if (c < chunks) {
#pragma unroll
        for (int q = 0; q < 4; ++q) put(0, q, next[q]);
    }
    __syncthreads();
    int cur = 0;
    const uint64_t my_stages = group < chunks ? (chunks - group + ngroups - 1) / ngroups : 1;
    for (; c < chunks; c += ngroups, ++stages) {
        const bool more = c + ngroups < chunks;
        if ((stages & 15u) == 0) matrix_stage_priority(stages, my_stages);
        request(next, c + ngroups);
        compute(cur);
        if (more) {
#pragma unroll
            for (int q = 0; q < 4; ++q) put(cur ^ 1, q, next[q]);
        }
        __syncthreads();
        cur ^= 1;
    }
    if (saw_big) atomicOr(big, 1u);
#pragma unroll
    for (int a = 0; a < TILE; ++a)
#pragma unroll
        for (int b = 0; b < TILE; ++b) {
            double ps = s[a][b];
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) ps += __shfl_down(ps, d, 16);
            if (mine && l == 0) {
                const uint64_t t = (uint64_t)ti * (ti + 1) / 2 + tj;
                partials[(t * TILE * TILE + a * TILE + b) * ngroups + group].s = ps;
            }
        }
    {
        const int i = si * 16 + (int)(threadIdx.x >> 4), j = sj * 16 + (int)(threadIdx.x & 15);
        const int pti = i / TILE, ptj = j / TILE;
        if (pti < side && ptj <= pti) {
            const uint64_t t = (uint64_t)pti * (pti + 1) / 2 + ptj;
            partials[(t * TILE * TILE + (i % TILE) * TILE + (j % TILE)) * ngroups + group].m = stages * kSuperBins - both_zero;
        }
    }
}

}  // namespace kpal
