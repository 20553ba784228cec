// count_kernels.hpp -- k-mer counting kernels for gfx950 (MI355X).
//
// Replaces the per-character Python loop of Profile.from_sequences (kpal/klib.py:154-168).
// Three strategies, all bit-exact (integer adds commute):
//   * global atomic   : one 64-bit global_atomic_add_x2 per k-mer straight into the 4^k table
//                       (any k; the k >= 13 path, where the table is 0.5 - 32 GiB).
//   * LDS direct      : k <= 7, the whole table is privatised per workgroup in LDS as u32
//                       (bank-replicated for tiny k), merged with one atomic per non-zero bin.
//   * partition       : 8 <= k <= 12.  Keys are radix-partitioned on their top 9 bits into 512
//                       buckets (count -> scan -> scatter, scatter staged and sorted in LDS so
//                       global writes are runs), then each bucket's <= 2^15 bins are histogrammed
//                       in LDS with ds_add_u32 and merged into the table.  HBM traffic per base:
//                       1 B read (count) + 1 B read (scatter) + 2 B written + 2 B read (keys).
#pragma once
#include "kpal_device.hpp"

namespace kpal {

constexpr int kPartBits = 9;
constexpr int kNumBuckets = 1 << kPartBits;  // 512
constexpr int kScatterThreads = 512;         // 8 waves
constexpr int kScatterWaves = kScatterThreads / 64;
constexpr int kScatterSteps = 4;             // wave-steps per wave per sub-tile
constexpr int kTileChunks = kScatterWaves * kScatterSteps * 64;  // 2048 chunks = 32 KiB per sub-tile
constexpr int kTileKeys = kTileChunks * 16;                      // <= 32768 k-mers per sub-tile

// ------------------------------------------------------------------------------------------
// Strategy 1: global atomics.  Each wave walks `steps_per_wave` consecutive 1-KiB steps.
// ------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void count_global_atomic_kernel(Span s, uint64_t steps_per_wave,
                                                                  unsigned long long *__restrict__ table)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t step0 = wave * steps_per_wave;
    const uint64_t total_steps = (s.nchunks + 63) / 64;
    if (step0 >= total_steps) return;
    const uint64_t step1 = min(step0 + steps_per_wave, total_steps);
    Chunk carry = load_chunk(s, (int64_t)(step0 * 64) - 1);
    for (uint64_t st = step0; st < step1; ++st) {
        uint64_t window;
        uint32_t mask;
        wave_step<K>(s, (int64_t)(st * 64 + lane), carry, window, mask);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (mask & (1u << (15 - j))) atomicAdd(&table[kmer_at<K>(window, j)], 1ULL);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Strategy 2: LDS-direct histogram for k <= 7 (4^k <= 16384 bins).  R bank-interleaved
// replicas (R*4^k*4 B = 64 KiB at most) so that for tiny k the 64 lanes of a wave do not
// serialise on a handful of addresses: replica r = lane % R lives at dword bin*R + r.
// A workgroup never sees 2^32 k-mers in one launch (the host caps a launch at 2^31 bytes).
// ------------------------------------------------------------------------------------------
template <int K>
struct LdsDirectCfg {
    static constexpr int kBins = 1 << (2 * K);
    static constexpr int kRep = (16384 / kBins) > 32 ? 32 : (16384 / kBins);
};

template <int K>
__global__ __launch_bounds__(512) void count_lds_direct_kernel(Span s, uint64_t steps_per_wave,
                                                               unsigned long long *__restrict__ table)
{
    constexpr int BINS = LdsDirectCfg<K>::kBins;
    constexpr int R = LdsDirectCfg<K>::kRep;
    __shared__ uint32_t h[BINS * R];
    for (int i = threadIdx.x; i < BINS * R; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int rep = lane % R;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t step0 = wave * steps_per_wave;
    const uint64_t total_steps = (s.nchunks + 63) / 64;
    if (step0 < total_steps) {
        const uint64_t step1 = min(step0 + steps_per_wave, total_steps);
        Chunk carry = load_chunk(s, (int64_t)(step0 * 64) - 1);
        for (uint64_t st = step0; st < step1; ++st) {
            uint64_t window;
            uint32_t mask;
            wave_step<K>(s, (int64_t)(st * 64 + lane), carry, window, mask);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (mask & (1u << (15 - j))) atomicAdd(&h[kmer_at<K>(window, j) * R + rep], 1u);
            }
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < BINS; b += blockDim.x) {
        unsigned long long v = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) v += h[b * R + r];
        if (v) atomicAdd(&table[b], v);
    }
}

// ------------------------------------------------------------------------------------------
// Strategy 3: partition.  Block `blk` of G owns the chunk range [blk*chunks_per_block, ...),
// a whole number of sub-tiles; within a sub-tile wave w owns kScatterSteps consecutive steps.
// ------------------------------------------------------------------------------------------
template <int K>
struct PartCfg {
    static constexpr int kKeyBits = 2 * K - kPartBits;  // 7 (k=8) .. 15 (k=12)
    static constexpr uint32_t kKeyMask = (1u << kKeyBits) - 1u;
};

// Visit every countable k-mer of one sub-tile held in registers.
template <int K, typename F>
__device__ __forceinline__ void for_each_kmer(const uint64_t (&window)[kScatterSteps],
                                              const uint32_t (&mask)[kScatterSteps], F f)
{
#pragma unroll
    for (int st = 0; st < kScatterSteps; ++st) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (mask[st] & (1u << (15 - j))) f(kmer_at<K>(window[st], j));
        }
    }
}

template <int K>
__device__ __forceinline__ void load_subtile(const Span &s, uint64_t tile_chunk0,
                                             uint64_t (&window)[kScatterSteps], uint32_t (&mask)[kScatterSteps])
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint64_t c0 = tile_chunk0 + (uint64_t)wave * kScatterSteps * 64;
    Chunk carry = load_chunk(s, (int64_t)c0 - 1);
#pragma unroll
    for (int st = 0; st < kScatterSteps; ++st)
        wave_step<K>(s, (int64_t)(c0 + st * 64 + lane), carry, window[st], mask[st]);
}

// A1: per-(bucket, block) key counts.  cntmat is bucket-major: cntmat[b * G + blk].
template <int K>
__global__ __launch_bounds__(kScatterThreads) void part_count_kernel(Span s, uint64_t tiles_per_block,
                                                                     uint32_t *__restrict__ cntmat)
{
    __shared__ uint32_t cnt[kNumBuckets];
    for (int i = threadIdx.x; i < kNumBuckets; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    const uint64_t tile0 = (uint64_t)blockIdx.x * tiles_per_block;
    for (uint64_t t = 0; t < tiles_per_block; ++t) {
        const uint64_t chunk0 = (tile0 + t) * kTileChunks;
        if (chunk0 >= s.nchunks) break;
        uint64_t window[kScatterSteps];
        uint32_t mask[kScatterSteps];
        load_subtile<K>(s, chunk0, window, mask);
        for_each_kmer<K>(window, mask, [&](uint32_t kmer) { atomicAdd(&cnt[kmer >> PartCfg<K>::kKeyBits], 1u); });
    }
    __syncthreads();
    for (int b = threadIdx.x; b < kNumBuckets; b += blockDim.x)
        cntmat[(uint64_t)b * gridDim.x + blockIdx.x] = cnt[b];
}

// A2: exclusive scan of the M = 512*G counts (flat, bucket-major) into 64-bit offsets; also
// bucket_start[b] (b = 0..512, last = total).  Single workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void part_scan_kernel(const uint32_t *__restrict__ cntmat, uint32_t G,
                                                         uint64_t *__restrict__ offs,
                                                         uint64_t *__restrict__ bucket_start)
{
    __shared__ uint64_t wave_tot[16];
    __shared__ uint64_t wave_base[17];
    const uint64_t M = (uint64_t)kNumBuckets * G;
    const uint64_t per = (M + 1023) / 1024;
    const uint64_t i0 = min((uint64_t)threadIdx.x * per, M);
    const uint64_t i1 = min(i0 + per, M);
    uint64_t sum = 0;
    for (uint64_t i = i0; i < i1; ++i) sum += cntmat[i];
    // inclusive scan of `sum` across the workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (int w = 0; w < 16; ++w) {
            wave_base[w] = run;
            run += wave_tot[w];
        }
        wave_base[16] = run;
    }
    __syncthreads();
    uint64_t run = wave_base[wave] + incl - sum;
    for (uint64_t i = i0; i < i1; ++i) {
        offs[i] = run;
        if (i % G == 0) bucket_start[i / G] = run;
        run += cntmat[i];
    }
    if (threadIdx.x == 0) bucket_start[kNumBuckets] = wave_base[16];
}

// A3: scatter.  Per sub-tile: count buckets in LDS, scan, place 16-bit keys bucket-sorted in
// LDS, then copy each bucket's run to its global cursor.  Output keys are bucket-major and
// contiguous; order inside a bucket is irrelevant to the histogram.
template <int K>
__global__ __launch_bounds__(kScatterThreads) void part_scatter_kernel(Span s, uint64_t tiles_per_block,
                                                                       const uint64_t *__restrict__ offs,
                                                                       uint16_t *__restrict__ keys_out)
{
    __shared__ uint16_t keys[kTileKeys];          // 64 KiB
    __shared__ uint32_t cnt[kNumBuckets];
    __shared__ uint32_t start[kNumBuckets];
    __shared__ uint32_t pos[kNumBuckets];
    __shared__ uint64_t gcur[kNumBuckets];
    __shared__ uint32_t wsum[kScatterWaves];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int KB = PartCfg<K>::kKeyBits;

    for (int b = threadIdx.x; b < kNumBuckets; b += blockDim.x) gcur[b] = offs[(uint64_t)b * gridDim.x + blockIdx.x];
    const uint64_t tile0 = (uint64_t)blockIdx.x * tiles_per_block;
    for (uint64_t t = 0; t < tiles_per_block; ++t) {
        const uint64_t chunk0 = (tile0 + t) * kTileChunks;
        if (chunk0 >= s.nchunks) break;
        cnt[threadIdx.x] = 0;  // blockDim.x == kNumBuckets == 512
        __syncthreads();
        uint64_t window[kScatterSteps];
        uint32_t mask[kScatterSteps];
        load_subtile<K>(s, chunk0, window, mask);
        for_each_kmer<K>(window, mask, [&](uint32_t kmer) { atomicAdd(&cnt[kmer >> KB], 1u); });
        __syncthreads();
        // exclusive scan of cnt[512] (one value per thread)
        {
            const uint32_t v = cnt[threadIdx.x];
            uint32_t incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t o = __shfl_up(incl, d);
                if (lane >= d) incl += o;
            }
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            uint32_t base = 0;
#pragma unroll
            for (int w = 0; w < kScatterWaves; ++w) base += (w < wave) ? wsum[w] : 0u;
            const uint32_t ex = base + incl - v;
            start[threadIdx.x] = ex;
            pos[threadIdx.x] = ex;
        }
        __syncthreads();
        for_each_kmer<K>(window, mask, [&](uint32_t kmer) {
            const uint32_t slot = atomicAdd(&pos[kmer >> KB], 1u);
            keys[slot] = (uint16_t)(kmer & PartCfg<K>::kKeyMask);
        });
        __syncthreads();
        // copy-out: wave w owns buckets [64w, 64w+64); lane l holds bucket 64w+l's metadata
        {
            const int b = wave * 64 + lane;
            const uint32_t my_n = cnt[b];
            const uint32_t my_st = start[b];
            const uint64_t my_g = gcur[b];
            for (int i = 0; i < 64; ++i) {
                const uint32_t n = __shfl(my_n, i);
                const uint32_t st = __shfl(my_st, i);
                const uint64_t g = __shfl(my_g, i);
                for (uint32_t e = lane; e < n; e += 64) keys_out[g + e] = keys[st + e];
            }
            gcur[b] = my_g + my_n;
        }
        __syncthreads();
    }
}

// B: histogram one slice of one bucket in LDS, merge into the table.
// grid = 512 buckets x slices; block 1024 threads.
template <int K>
__global__ __launch_bounds__(1024) void part_hist_kernel(const uint16_t *__restrict__ keys,
                                                         const uint64_t *__restrict__ bucket_start,
                                                         uint32_t slices, unsigned long long *__restrict__ table)
{
    constexpr int KB = PartCfg<K>::kKeyBits;
    constexpr int BINS = 1 << KB;
    __shared__ __attribute__((aligned(16))) uint32_t hist[BINS];  // 128 KiB at k = 12
    const uint32_t b = blockIdx.x / slices;
    const uint32_t sl = blockIdx.x % slices;
    for (int i = threadIdx.x; i < BINS; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const uint64_t bs = bucket_start[b], be = bucket_start[b + 1];
    const uint64_t len = be - bs;
    const uint64_t per = (len + slices - 1) / slices;
    uint64_t e0 = bs + min((uint64_t)sl * per, len);
    const uint64_t e1 = bs + min((uint64_t)(sl + 1) * per, len);
    // head: up to the next 16-byte boundary (8 keys)
    uint64_t a0 = min((e0 + 7) & ~7ULL, e1);
    for (uint64_t e = e0 + threadIdx.x; e < a0; e += blockDim.x) atomicAdd(&hist[keys[e]], 1u);
    const uint64_t a1 = a0 + ((e1 - a0) & ~7ULL);
    const uint4 *kv = reinterpret_cast<const uint4 *>(keys + a0);
    const uint64_t nvec = (a1 - a0) >> 3;
    for (uint64_t v = threadIdx.x; v < nvec; v += blockDim.x) {
        const uint4 q = kv[v];
        atomicAdd(&hist[q.x & 0xFFFFu], 1u);
        atomicAdd(&hist[q.x >> 16], 1u);
        atomicAdd(&hist[q.y & 0xFFFFu], 1u);
        atomicAdd(&hist[q.y >> 16], 1u);
        atomicAdd(&hist[q.z & 0xFFFFu], 1u);
        atomicAdd(&hist[q.z >> 16], 1u);
        atomicAdd(&hist[q.w & 0xFFFFu], 1u);
        atomicAdd(&hist[q.w >> 16], 1u);
    }
    for (uint64_t e = a1 + threadIdx.x; e < e1; e += blockDim.x) atomicAdd(&hist[keys[e]], 1u);
    __syncthreads();
    unsigned long long *dst = table + ((uint64_t)b << KB);
    for (int i = threadIdx.x; i < BINS; i += blockDim.x) {
        const uint32_t v = hist[i];
        if (v) atomicAdd(&dst[i], (unsigned long long)v);
    }
}

// ------------------------------------------------------------------------------------------
// Synthetic reads.  One thread per 16 output bytes.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void synth_reads_kernel(uint64_t seed, uint64_t first_read, uint64_t n_reads,
                                                          uint32_t read_len, int noisy, uint8_t *__restrict__ out)
{
    const uint64_t total = n_reads * (uint64_t)(read_len + 1);
    const uint64_t nvec = (total + 15) / 16;
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec;
         v += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t o0 = v * 16;
        uint64_t r = o0 / (read_len + 1);
        uint32_t pos = (uint32_t)(o0 - r * (read_len + 1));
        uint64_t cached_word = ~0ULL, w = 0;
        uint8_t bytes[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            uint8_t c;
            if (pos == read_len) {
                c = '\n';
            } else {
                const uint64_t g = (first_read + r) * (uint64_t)read_len + pos;
                if ((g >> 5) != cached_word) {
                    cached_word = g >> 5;
                    w = mix64(seed * 0xD1342543DE82EF95ULL + cached_word);
                }
                const uint32_t code = (uint32_t)(w >> (2 * (g & 31))) & 3u;
                c = (uint8_t)((0x54474341u >> (8 * code)) & 0xFFu);  // "ACGT"
                if (noisy) {
                    const uint64_t hsh = mix64(~seed + g);
                    if (hsh % 1000 == 0) c = 'N';
                    else if (hsh % 100 == 1) c |= 0x20;
                }
            }
            bytes[q] = c;
            if (++pos > read_len) {
                pos = 0;
                ++r;
            }
        }
        if (o0 + 16 <= total) {
            uint4 pk;
            pk.x = bytes[0] | (bytes[1] << 8) | (bytes[2] << 16) | ((uint32_t)bytes[3] << 24);
            pk.y = bytes[4] | (bytes[5] << 8) | (bytes[6] << 16) | ((uint32_t)bytes[7] << 24);
            pk.z = bytes[8] | (bytes[9] << 8) | (bytes[10] << 16) | ((uint32_t)bytes[11] << 24);
            pk.w = bytes[12] | (bytes[13] << 8) | (bytes[14] << 16) | ((uint32_t)bytes[15] << 24);
            *reinterpret_cast<uint4 *>(out + o0) = pk;
        } else {
            for (int q = 0; q < 16 && o0 + q < total; ++q) out[o0 + q] = bytes[q];
        }
    }
}

}  // namespace kpal
