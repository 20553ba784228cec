// count_kernels.hpp -- k-mer counting kernels for gfx950 (MI355X).
//
// Replaces the per-character Python loop of Profile.from_sequences (kpal/klib.py:154-168).
// Three strategies, all bit-exact (integer adds commute):
//   * global atomic   : one 64-bit global_atomic_add_x2 per k-mer straight into the 4^k table
//                       (any k; the k >= 13 path, where the table is 0.5 - 32 GiB).
//   * LDS direct      : k <= 7, the whole table is privatised per workgroup in LDS as u32
//                       (bank-replicated for tiny k), merged with one atomic per non-zero bin.
//   * partition       : 8 <= k <= 12.  Keys are radix-partitioned on their top 9 bits into 512
//                       buckets (count -> scan -> scatter, scatter staged per bucket in LDS so
//                       global writes are runs), then each bucket's <= 2^15 bins are histogrammed
//                       in LDS with ds_add_u32 and merged into the table.  HBM traffic per base:
//                       1 B read (count) + 1 B read (scatter) + 2 B written + 2 B read (keys).
#pragma once
#include "kpal_device.hpp"

namespace kpal {

constexpr int kPartBits = 9;
constexpr int kNumBuckets = 1 << kPartBits;  // 512
constexpr int kScatterThreads = 512;         // 8 waves; two workgroups per CU (LDS 70 KiB each)
constexpr int kScatterWaves = kScatterThreads / 64;
constexpr int kScatterSteps = 3;             // wave-steps per wave per sub-tile
constexpr int kTileChunks = kScatterWaves * kScatterSteps * 64;  // 1536 chunks = 24 KiB per sub-tile
constexpr int kBucketsPerWave = (kNumBuckets + kScatterWaves - 1) / kScatterWaves;  // 43: copy-out share of a wave
constexpr int kSlotCap = 64;  // LDS staging slots per bucket per sub-tile (mean fill 44 for 150 bp reads)

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
        if (interior_range(s, st * 64, st * 64 + 64)) wave_step<K, false>(s, (int64_t)(st * 64 + lane), carry, window, mask);
        else wave_step<K, true>(s, (int64_t)(st * 64 + lane), carry, window, mask);
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
            if (interior_range(s, st * 64, st * 64 + 64)) wave_step<K, false>(s, (int64_t)(st * 64 + lane), carry, window, mask);
            else wave_step<K, true>(s, (int64_t)(st * 64 + lane), carry, window, mask);
            // branch-free: a k-mer that must not be counted adds 0 to whatever bin its bits name
#pragma unroll
            for (int j = 0; j < 16; ++j) atomicAdd(&h[kmer_at<K>(window, j) * R + rep], (mask >> (15 - j)) & 1u);
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
// Strategy 3: partition.  Block `blk` of G owns steps_per_block (a multiple of 24) consecutive
// wave-steps; the streaming waves of a block split that range into contiguous sub-ranges, so a
// wave's left-neighbour chunk is carried in registers from step to step and never re-encoded.
// The count (A1) and scatter (A3) kernels use the same block ranges, hence agree on every
// per-(bucket, block) count even though they organise their waves differently.
// ------------------------------------------------------------------------------------------
template <int K>
struct PartCfg {
    static constexpr int kKeyBits = 2 * K - kPartBits;  // 7 (k=8) .. 15 (k=12)
    static constexpr uint32_t kKeyMask = (1u << kKeyBits) - 1u;
};

// One wave-step at absolute step index `step` (interior fast path chosen per wave).
template <int K>
__device__ __forceinline__ void part_step(const Span &s, uint64_t step, Chunk &carry, uint64_t &window, uint32_t &mask)
{
    const int lane = threadIdx.x & 63;
    if (interior_range(s, step * 64, step * 64 + 64)) wave_step<K, false>(s, (int64_t)(step * 64 + lane), carry, window, mask);
    else wave_step<K, true>(s, (int64_t)(step * 64 + lane), carry, window, mask);
}

// A1: per-(bucket, block) key counts.  cntmat is bucket-major: cntmat[b * G + blk].
// The 512 counters are kept in 32 bank-interleaved replicas (replica = lane % 32 lives in LDS
// bank lane % 32), so a wave's 64 ds_add_u32 never conflict; a k-mer that must not be counted
// adds 0 (branch-free).
template <int K>
__global__ __launch_bounds__(kScatterThreads) void part_count_kernel(Span s, uint64_t steps_per_block,
                                                                     uint32_t *__restrict__ cntmat)
{
    __shared__ uint32_t cnt[kNumBuckets * 32];  // 64 KiB
    for (int i = threadIdx.x; i < kNumBuckets * 32; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    const uint32_t rep = threadIdx.x & 31;
    const uint64_t total_steps = (s.nchunks + 63) / 64;
    const uint64_t steps_per_wave = steps_per_block / kScatterWaves;
    const uint64_t step0 = (uint64_t)blockIdx.x * steps_per_block + (uint64_t)(threadIdx.x >> 6) * steps_per_wave;
    if (step0 < total_steps) {
        const uint64_t step1 = min(step0 + steps_per_wave, total_steps);
        Chunk carry = load_chunk(s, (int64_t)(step0 * 64) - 1);
        for (uint64_t st = step0; st < step1; ++st) {
            uint64_t window;
            uint32_t mask;
            part_step<K>(s, st, carry, window, mask);
#pragma unroll
            for (int j = 0; j < 16; ++j)
                atomicAdd(&cnt[(kmer_at<K>(window, j) >> PartCfg<K>::kKeyBits) * 32 + rep], (mask >> (15 - j)) & 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < kNumBuckets; b += blockDim.x) {
        uint32_t v = 0;
#pragma unroll
        for (int r = 0; r < 32; ++r) v += cnt[b * 32 + ((r + b) & 31)];
        cntmat[(uint64_t)b * gridDim.x + blockIdx.x] = v;
    }
}

// A2a: one workgroup per bucket: exclusive scan of that bucket's G per-block counts into
// offs32 (position of each block's keys inside the bucket) and the bucket total.
__global__ __launch_bounds__(256) void part_rowscan_kernel(const uint32_t *__restrict__ cntmat, uint32_t G,
                                                           uint32_t *__restrict__ offs32,
                                                           uint64_t *__restrict__ bucket_total)
{
    __shared__ uint32_t wsum[4];
    const uint32_t *row = cntmat + (uint64_t)blockIdx.x * G;
    uint32_t *orow = offs32 + (uint64_t)blockIdx.x * G;
    const uint32_t per = (G + 255) / 256;
    const uint32_t i0 = min(threadIdx.x * per, G), i1 = min(i0 + per, G);
    uint32_t sum = 0;
    for (uint32_t i = i0; i < i1; ++i) sum += row[i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) base += (w < wave) ? wsum[w] : 0u;
    uint32_t run = base + incl - sum;
    for (uint32_t i = i0; i < i1; ++i) {
        orow[i] = run;
        run += row[i];
    }
    if (threadIdx.x == 255) bucket_total[blockIdx.x] = (uint64_t)run;  // a launch holds < 2^32 keys
}

// A2b: exclusive scan of the 512 bucket totals -> bucket_start[0..512].
__global__ __launch_bounds__(kNumBuckets) void part_bucketscan_kernel(const uint64_t *__restrict__ bucket_total,
                                                                      uint64_t *__restrict__ bucket_start)
{
    __shared__ uint64_t wsum[kNumBuckets / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t v = bucket_total[threadIdx.x];
    uint64_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint64_t base = 0;
#pragma unroll
    for (int w = 0; w < kNumBuckets / 64; ++w) base += (w < wave) ? wsum[w] : 0ULL;
    bucket_start[threadIdx.x] = base + incl - v;
    if (threadIdx.x == kNumBuckets - 1) bucket_start[kNumBuckets] = base + incl;
}

// A3: scatter.  Every bucket has kSlotCap 16-bit staging slots in LDS.  A k-mer takes the next
// slot of its bucket with one returning ds_add (an uncounted k-mer adds 0 and its store is
// diverted to a dummy halfword): per step 16 atomics are in flight, then 16 ds_write_b16.  The
// slot index is rotated by the bucket index so that buckets filling in lock-step hit different
// banks.  Slots >= kSlotCap -- rare for unskewed input, the whole stream for a homopolymer -- are
// written straight to their final global position after the step.  After the tile (3 steps per
// wave), wave w copies the staged runs of buckets [64w, 64w+64) to the buckets' global cursors,
// one masked 2-byte-per-lane store per bucket, and advances the cursors by the full slot
// count.  Output keys are bucket-major and contiguous; order inside a bucket is irrelevant to
// the histogram.
// Diagnostic stamp (only in the STAMP build of the scatter kernel; never in the product launch).
__device__ __forceinline__ unsigned long long phase_stamp()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

// MODE: 0 product; 1 per-phase stamps; 2..4 timing ablations (wrong results, diagnostics only):
// 2 = no global stores in the copy-out, 3 = no copy-out work at all, 4 = no placement,
// 5 = placement only (synthetic windows, no loads, no copy-out).
template <int K, int MODE = 0>
__global__ __launch_bounds__(kScatterThreads) void part_scatter_kernel(Span s, uint64_t steps_per_block,
                                                                       const uint32_t *__restrict__ offs32,
                                                                       const uint64_t *__restrict__ bucket_start,
                                                                       uint16_t *__restrict__ keys_out,
                                                                       unsigned long long *__restrict__ dbg = nullptr)
{
    constexpr bool STAMP = MODE == 1;
    unsigned long long acc[5] = {0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
    // byte layout: keys rows [512][64] u16 | dummy halfword (+pad) | pos[512] u32 | gcur[512] u64
    constexpr uint32_t kKeysBytes = kNumBuckets * kSlotCap * 2;
    constexpr uint32_t kDummyByte = kKeysBytes;
    __shared__ __attribute__((aligned(16))) unsigned char lds[kKeysBytes + 16];
    __shared__ uint32_t pos[kNumBuckets];
    __shared__ uint64_t gcur[kNumBuckets];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int KB = PartCfg<K>::kKeyBits;
    static_assert(kSlotCap == 64 && kBucketsPerWave <= 64, "copy-out: one lane per slot, one lane per owned bucket");

    for (int b = threadIdx.x; b < kNumBuckets; b += blockDim.x) {
        gcur[b] = bucket_start[b] + offs32[(uint64_t)b * gridDim.x + blockIdx.x];
        pos[b] = 0;
    }
    __syncthreads();
    const uint64_t total_steps = (s.nchunks + 63) / 64;
    const uint64_t steps_per_wave = steps_per_block / kScatterWaves;
    const uint64_t block_step0 = (uint64_t)blockIdx.x * steps_per_block;
    const uint64_t step0 = block_step0 + (uint64_t)wave * steps_per_wave;
    Chunk carry = load_chunk(s, (int64_t)(step0 * 64) - 1);
    for (uint64_t t = 0; t < steps_per_wave; t += kScatterSteps) {
        if (block_step0 + t >= total_steps) break;  // block-uniform: wave 0 owns the lowest addresses
        uint64_t window[kScatterSteps];
        uint32_t mask[kScatterSteps];
        if constexpr (STAMP) t0 = phase_stamp();
        if constexpr (MODE == 5) {   // placement-only timing: pseudo-random windows, no global traffic
#pragma unroll
            for (int st = 0; st < kScatterSteps; ++st) {
                window[st] = mix64((step0 + t + st) * 64 + lane);
                mask[st] = 0xFFFFu & ~(uint32_t)((window[st] >> 60) == 0 ? 0xFFF0u : 0u);
            }
        } else {
#pragma unroll
        for (int st = 0; st < kScatterSteps; ++st) part_step<K>(s, step0 + t + st, carry, window[st], mask[st]);
        }
        if constexpr (STAMP) { asm volatile("" ::"v"(window[0]), "v"(window[kScatterSteps - 1])); t1 = phase_stamp(); acc[0] += t1 - t0; t0 = t1; }
        if constexpr (MODE != 4) {
#pragma unroll
            for (int st = 0; st < kScatterSteps; ++st) {
                const uint64_t w = window[st];
                const uint32_t m = mask[st];
                uint32_t slot[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint32_t b4 = (kmer_at<K>(w, j) >> (KB - 2)) & 0x7FCu;  // 4 * bucket
                    slot[j] = atomicAdd((uint32_t *)((unsigned char *)pos + b4), (m >> (15 - j)) & 1u);
                }
                uint32_t smax = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint32_t kmer = kmer_at<K>(w, j);
                    const uint32_t b4 = (kmer >> (KB - 2)) & 0x7FCu;
                    const uint32_t x = slot[j] | (((~m >> (15 - j)) & 1u) << 6);    // >= 64: not counted, or row full
                    const uint32_t at = ((2u * slot[j] + b4) & 126u) | (b4 << 5);      // byte offset of the rotated slot
                    *(uint16_t *)(lds + (x < 64u ? at : kDummyByte)) = (uint16_t)(kmer & PartCfg<K>::kKeyMask);
                    smax = max(smax, slot[j]);
                }
                if (smax >= (uint32_t)kSlotCap) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        if (((m >> (15 - j)) & 1u) && slot[j] >= (uint32_t)kSlotCap) {
                            const uint32_t kmer = kmer_at<K>(w, j);
                            keys_out[gcur[kmer >> KB] + slot[j]] = (uint16_t)(kmer & PartCfg<K>::kKeyMask);
                        }
                    }
                }
            }
        }
        if constexpr (STAMP) { t1 = phase_stamp(); acc[1] += t1 - t0; t0 = t1; }
        __syncthreads();
        if constexpr (STAMP) { t1 = phase_stamp(); acc[2] += t1 - t0; t0 = t1; }
        if constexpr (MODE == 5) {
            if (threadIdx.x < kNumBuckets) pos[threadIdx.x] = 0;
        } else if constexpr (MODE != 3) {
            // Copy-out.  Wave w owns buckets [first, first + kBucketsPerWave); lane l holds bucket
            // first+l's byte count and global byte address.  Per bucket: three v_readlane build a
            // buffer descriptor {base = run start, num_records = run bytes} in SGPRs and ONE
            // buffer_store_short writes the staged row -- lanes beyond the run are dropped by the
            // hardware range check, so there is no exec-mask juggling and no branch.
            const int first = wave * kBucketsPerWave;
            const int mine = first + lane;
            const bool own = lane < kBucketsPerWave && mine < kNumBuckets;
            const uint32_t my_n = own ? pos[mine] : 0u;
            const uint64_t my_g = own ? gcur[mine] : 0ULL;
            const uint64_t my_addr = (uint64_t)keys_out + 2ULL * my_g;
            const uint32_t my_lo = (uint32_t)my_addr, my_hi = (uint32_t)(my_addr >> 32);
            const uint32_t my_bytes = 2u * min(my_n, (uint32_t)kSlotCap);
            const uint32_t r0 = 2u * lane + 4u * first;
            const unsigned char *rows = lds + (uint32_t)first * 128u;
#pragma unroll
            for (int i0 = 0; i0 < kBucketsPerWave; i0 += 8) {
                uint16_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {   // unconditional LDS reads first (8 in flight)
                    const int i = (i0 + u) < kBucketsPerWave ? (i0 + u) : (kBucketsPerWave - 1);
                    v[u] = *(const uint16_t *)(rows + i * 128 + ((r0 + 4u * i) & 126u));
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (i0 + u < kBucketsPerWave) {
                        const uint32_t lo = __builtin_amdgcn_readlane(my_lo, i0 + u);
                        const uint32_t hi = __builtin_amdgcn_readlane(my_hi, i0 + u);
                        const uint32_t nb = __builtin_amdgcn_readlane(my_bytes, i0 + u);
                        if constexpr (MODE != 2) {
                            __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                                (void *)(((uint64_t)hi << 32) | lo), (short)0, (int)nb, 0x00020000);
                            __builtin_amdgcn_raw_buffer_store_b16((short)v[u], rsrc, 2 * lane, 0, 0);
                        } else {
                            asm volatile("" ::"s"(lo), "s"(hi), "s"(nb), "v"(v[u]));
                        }
                    }
                }
            }
            if (own) {
                gcur[mine] = my_g + my_n;
                pos[mine] = 0;
            }
        }
        if constexpr (STAMP) { t1 = phase_stamp(); acc[3] += t1 - t0; t0 = t1; }
        __syncthreads();
        if constexpr (STAMP) { t1 = phase_stamp(); acc[4] += t1 - t0; }
    }
    if constexpr (STAMP) {
        if (lane == 0)
            for (int i = 0; i < 5; ++i) dbg[((uint64_t)blockIdx.x * kScatterWaves + wave) * 5 + i] = acc[i];
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
    auto add8 = [&](const uint4 q) {
        atomicAdd(&hist[q.x & 0xFFFFu], 1u);
        atomicAdd(&hist[q.x >> 16], 1u);
        atomicAdd(&hist[q.y & 0xFFFFu], 1u);
        atomicAdd(&hist[q.y >> 16], 1u);
        atomicAdd(&hist[q.z & 0xFFFFu], 1u);
        atomicAdd(&hist[q.z >> 16], 1u);
        atomicAdd(&hist[q.w & 0xFFFFu], 1u);
        atomicAdd(&hist[q.w >> 16], 1u);
    };
    // four 16-byte loads in flight per lane (64 KiB per workgroup) to cover HBM latency
    uint64_t v = threadIdx.x;
    const uint64_t B = blockDim.x;
    for (; v + 3 * B < nvec; v += 4 * B) {
        const uint4 q0 = kv[v], q1 = kv[v + B], q2 = kv[v + 2 * B], q3 = kv[v + 3 * B];
        add8(q0);
        add8(q1);
        add8(q2);
        add8(q3);
    }
    for (; v < nvec; v += blockDim.x) add8(kv[v]);
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
