// partition_kernels.hpp -- radix-partition counting for gfx950 (MI355X).
//
// k = 8..12 (one level).  A k-mer's 2k bits split into a 9-bit BUCKET (top) and a KB = 2k-9 bit
// KEY.  Keys are scattered bucket-major into a u16 array, then each bucket's <= 2^15 bins are
// histogrammed in LDS and merged into the int64 table.  This file holds the EXACT-OFFSET pipeline
// (KPAL_STRATEGY_PARTITION); the default for these k is the one-pass chunked pipeline of
// chunk_kernels.hpp, which reuses the staging and the histogram design from here:
//     A1 part_count      per-(bucket, block) counts          reads the input
//     A2 rowscan+bucketscan  exact, deterministic offsets
//     A3 part_scatter    keys -> bucket runs                 reads the input, writes 2 B per k-mer
//     B  part_hist       LDS histogram per bucket            reads 2 B per k-mer
// k = 13..16 (two levels).  The top 2k-24 bits select one of 4/16/64/256 COARSE buckets; the
// 24-bit residual is exactly a k=12 k-mer, so after a coarse count/scan/scatter into u32 residuals
// (C1..C3) every coarse bucket runs a one-level pipeline on its residual stream -- by default the
// chunked aligned-line scatter of chunk_kernels.hpp, optionally (KPAL_LEVEL2=0) key_count, scans,
// key_scatter, part_hist from this file -- all coarse buckets in one 2-D launch per stage.
//
// Everything is integer and order-independent: every k-mer gets a unique slot from a returning
// atomic, count and scatter kernels share one block->range mapping, so results are bit-exact.
#pragma once
#include "kpal_device.hpp"

namespace kpal {

constexpr int kPartBits = 9;
constexpr int kNumBuckets = 1 << kPartBits;  // 512
constexpr int kScatterThreads = 512;         // 8 waves; two workgroups per CU
constexpr int kScatterWaves = kScatterThreads / 64;
constexpr int kScatterSteps = 3;             // wave-steps per wave per tile (24 KiB of input per tile)
constexpr int kBucketsPerWave = kNumBuckets / kScatterWaves;  // 64: copy-out share of a wave
constexpr int kSlotCap = 64;  // LDS staging slots per bucket per tile (mean fill 44 for 150 bp reads)
constexpr int kStepsPerBlockQuantum = kScatterWaves * kScatterSteps;  // block ranges are multiples of 24 steps

template <int K>
struct PartCfg {
    static constexpr int kKeyBits = 2 * K - kPartBits;  // 7 (k=8) .. 15 (k=12)
    static constexpr uint32_t kKeyMask = (1u << kKeyBits) - 1u;
};

// (part_step<K>: kpal_device.hpp)

// ------------------------------------------------------------------------------------------
// LDS staging of the ASCII scatter kernel (per-tile rows, one run per bucket per tile).
//   rows : 512 rows x 64 u16 slots (64 KiB), slot index rotated by the bucket so that buckets
//          filling in lock-step hit different banks; one dummy halfword after the rows
//   pos  : slots taken per bucket in the current tile
//   gcur : global cursor (key index) per bucket
// ------------------------------------------------------------------------------------------
constexpr uint32_t kRowsBytes = kNumBuckets * kSlotCap * 2;

// Place 16 values v[j] = (bucket << KB) | key; bit (15-j) of `valid` says whether v[j] counts.
// 16 returning ds_add are in flight (an uncounted value adds 0 and its store is diverted to the
// dummy halfword), then 16 ds_write_b16.  Slots >= kSlotCap -- rare for unskewed input, the
// whole stream for a homopolymer -- are stored straight to their final global position.
template <int KB>
__device__ __forceinline__ void place16(unsigned char *rows, uint32_t *pos, const uint64_t *gcur,
                                        uint16_t *__restrict__ keys_out, const uint32_t (&v)[16], uint32_t valid)
{
    constexpr uint32_t kKeyMask = (1u << KB) - 1u;
    uint32_t slot[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t b4 = (v[j] >> (KB - 2)) & 0x7FCu;  // 4 * bucket
        slot[j] = atomicAdd((uint32_t *)((unsigned char *)pos + b4), (valid >> (15 - j)) & 1u);
    }
    uint32_t smax = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t b4 = (v[j] >> (KB - 2)) & 0x7FCu;
        const uint32_t x = slot[j] | (((~valid >> (15 - j)) & 1u) << 6);      // >= 64: not counted, or row full
        const uint32_t at = ((2u * slot[j] + b4) & 126u) | (b4 << 5);        // byte offset of the rotated slot
        *(uint16_t *)(rows + (x < 64u ? at : kRowsBytes)) = (uint16_t)(v[j] & kKeyMask);
        smax = max(smax, slot[j]);
    }
    if (smax >= (uint32_t)kSlotCap) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (((valid >> (15 - j)) & 1u) && slot[j] >= (uint32_t)kSlotCap)
                keys_out[gcur[v[j] >> KB] + slot[j]] = (uint16_t)(v[j] & kKeyMask);
        }
    }
}

// Copy-out of one tile.  Wave w owns buckets [64w, 64w+64); lane l holds bucket 64w+l's byte
// count and global byte address.  Per bucket: three v_readlane build a buffer descriptor
// {base = run start, num_records = run bytes} in SGPRs and ONE buffer_store_short writes the
// staged row -- lanes beyond the run are dropped by the hardware range check, so there is no
// exec-mask juggling and no branch.  Then the cursors advance by the full slot count.
__device__ __forceinline__ void copy_out_tile(const unsigned char *rows, uint32_t *pos, uint64_t *gcur,
                                              uint16_t *__restrict__ keys_out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int first = wave * kBucketsPerWave;
    const int mine = first + lane;
    const uint32_t my_n = pos[mine];
    const uint64_t my_g = gcur[mine];
    const uint64_t my_addr = (uint64_t)keys_out + 2ULL * my_g;
    const uint32_t my_lo = (uint32_t)my_addr, my_hi = (uint32_t)(my_addr >> 32);
    const uint32_t my_bytes = 2u * min(my_n, (uint32_t)kSlotCap);
    const uint32_t r0 = 2u * lane + 4u * first;
    const unsigned char *wrows = rows + (uint32_t)first * 128u;
#pragma unroll
    for (int i0 = 0; i0 < kBucketsPerWave; i0 += 8) {
        uint16_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)   // unconditional LDS reads first (8 in flight)
            v[u] = *(const uint16_t *)(wrows + (i0 + u) * 128 + ((r0 + 4u * (i0 + u)) & 126u));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t lo = __builtin_amdgcn_readlane(my_lo, i0 + u);
            const uint32_t hi = __builtin_amdgcn_readlane(my_hi, i0 + u);
            const uint32_t nb = __builtin_amdgcn_readlane(my_bytes, i0 + u);
            __amdgpu_buffer_rsrc_t rsrc =
                __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), (short)0, (int)nb, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b16((short)v[u], rsrc, 2 * lane, 0, 0);
        }
    }
    gcur[mine] = my_g + my_n;
    pos[mine] = 0;
}

// ------------------------------------------------------------------------------------------
// A1: per-(bucket, block) key counts of an ASCII span.  cntmat is bucket-major: cntmat[b*G + blk].
// Block `blk` owns steps [blk*SPB, (blk+1)*SPB); wave w streams the w-th eighth of it, carrying its
// left-neighbour chunk in registers.  The 512 counters are kept in 16 bank-interleaved replicas
// (replica = lane % 16), so a wave's 64 ds_add_u32 rarely conflict;
// a k-mer that must not be counted adds 0 (branch-free).
// ------------------------------------------------------------------------------------------
constexpr int kCountReplicas = 16;   // bank-interleaved copies of the 512 bucket counters (32 KiB: four workgroups per CU)

__device__ __forceinline__ void flush_bucket_counts(const uint32_t *cnt, uint32_t *__restrict__ cntmat_block,
                                                    uint64_t stride)
{
    for (int b = threadIdx.x; b < kNumBuckets; b += blockDim.x) {
        uint32_t v = 0;
#pragma unroll
        for (int r = 0; r < kCountReplicas; ++r) v += cnt[b * kCountReplicas + ((r + b) & (kCountReplicas - 1))];
        cntmat_block[(uint64_t)b * stride] = v;
    }
}

template <int K>
__global__ __launch_bounds__(kScatterThreads) void part_count_kernel(Span s, uint64_t steps_per_block,
                                                                     uint32_t *__restrict__ cntmat)
{
    __shared__ uint32_t cnt[kNumBuckets * kCountReplicas];
    for (int i = threadIdx.x; i < kNumBuckets * kCountReplicas; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    const uint32_t rep = threadIdx.x & (kCountReplicas - 1);
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
                atomicAdd(&cnt[(kmer_at<K>(window, j) >> PartCfg<K>::kKeyBits) * kCountReplicas + rep], (mask >> (15 - j)) & 1u);
        }
    }
    __syncthreads();
    flush_bucket_counts(cnt, cntmat + blockIdx.x, gridDim.x);
}

// A2a: one workgroup per row: exclusive scan of the row's G per-block counts into offs32
// (position of each block's keys inside the bucket) and the row total.  Rows are buckets; with a
// 2-D grid, blockIdx.y selects the coarse bucket's matrix.
__global__ __launch_bounds__(256) void part_rowscan_kernel(const uint32_t *__restrict__ cntmat, uint32_t G,
                                                           uint32_t *__restrict__ offs32,
                                                           uint64_t *__restrict__ row_total)
{
    __shared__ uint32_t wsum[4];
    const uint64_t rowid = (uint64_t)blockIdx.y * gridDim.x + blockIdx.x;
    const uint32_t *row = cntmat + rowid * G;
    uint32_t *orow = offs32 + rowid * G;
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
    if (threadIdx.x == 255) row_total[rowid] = (uint64_t)run;  // a launch holds < 2^32 keys
}

// A2b: exclusive scan of n <= 512 row totals -> start[0..n] (start[n] = grand total), offset by
// base[blockIdx.x] when given.  With a 1-D grid of NB blocks this scans NB independent groups of
// n rows (the 512 fine buckets of every coarse bucket).
// It also plans the histogram launch (B): bucket b is cut into slices[b] = ceil(size / T) slices,
// T = max(2 x mean bucket size, kMinSliceKeys) -- one slice per bucket for unskewed input, up to
// n/2 extra slices spread over the oversized buckets of skewed input (a homopolymer puts every key
// into ONE bucket).  slice_start[0..n] is the exclusive scan of slices; sum <= n + n/2.
constexpr uint32_t kMinSliceKeys = 1u << 16;
constexpr uint32_t kHistGridX = kNumBuckets + kNumBuckets / 2;   // upper bound of the slice count

__global__ __launch_bounds__(kNumBuckets) void part_bucketscan_kernel(const uint64_t *__restrict__ row_total, uint32_t n,
                                                                      const uint64_t *__restrict__ base,
                                                                      uint64_t *__restrict__ start,
                                                                      uint32_t *__restrict__ slice_start)
{
    __shared__ uint64_t wsum[kNumBuckets / 64];
    __shared__ uint32_t ssum[kNumBuckets / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t v = threadIdx.x < n ? row_total[(uint64_t)blockIdx.x * n + threadIdx.x] : 0ULL;
    uint64_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint64_t off = base ? base[blockIdx.x] : 0ULL;
    uint64_t total = 0;
#pragma unroll
    for (int w = 0; w < kNumBuckets / 64; ++w) {
        off += (w < wave) ? wsum[w] : 0ULL;
        total += wsum[w];
    }
    uint64_t *out = start + (uint64_t)blockIdx.x * (n + 1);
    if (threadIdx.x < n) out[threadIdx.x] = off + incl - v;
    if (threadIdx.x == n - 1) out[n] = off + incl;
    if (!slice_start) return;
    // slices per bucket and their exclusive scan
    const uint64_t target = max((uint64_t)kMinSliceKeys, 2 * ((total + n - 1) / n));
    const uint32_t slices = threadIdx.x < n ? (uint32_t)max((uint64_t)1, (v + target - 1) / target) : 0u;
    uint32_t sincl = slices;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(sincl, d);
        if (lane >= d) sincl += o;
    }
    if (lane == 63) ssum[wave] = sincl;
    __syncthreads();
    uint32_t soff = 0;
#pragma unroll
    for (int w = 0; w < kNumBuckets / 64; ++w) soff += (w < wave) ? ssum[w] : 0u;
    uint32_t *sout = slice_start + (uint64_t)blockIdx.x * (n + 1);
    if (threadIdx.x < n) sout[threadIdx.x] = soff + sincl - slices;
    if (threadIdx.x == n - 1) sout[n] = soff + sincl;
}

// ------------------------------------------------------------------------------------------
// Aligned-line staging (used by the level-2 key scatter of the two-level path).  Measured (tools/store_bench2.hip): a scattered, unaligned ~88-byte run
// costs ~26-38 clk per CU in the store path, a full aligned 128-byte line ~14 clk.  So the rows
// persist across tiles and only whole aligned lines leave the CU:
//   * one 1024-thread workgroup per CU; 512 rows x 128 slots (256 B) = 128 KiB of LDS;
//     row slot i corresponds to global key index gline[b] + i, gline[b] a multiple of 64;
//   * a tile is ONE step per wave (16 KiB of input, ~30 new keys per bucket), so a row that
//     holds < 64 leftover keys cannot overflow its 128 slots on unskewed input;
//   * after every tile wave w flushes, for each of its 32 rows that holds >= 64 keys, line 0
//     (64 keys = one aligned 128-byte line: 8 lanes x ds_read_b128 + global_store_dwordx4, eight
//     rows per instruction), moves the < 64 leftover keys down and advances gline by 64;
//   * lo (kept in the low six bits of gline[b]) marks slots of the current line that must not be stored (the part of the first line
//     that belongs to the previous block's segment, or keys that went out directly);
//   * slots >= 128 (skew) are stored directly at gline[b] + slot; the row logic accounts for them.
// ------------------------------------------------------------------------------------------
constexpr int kLineThreads = 1024;
constexpr int kLineWaves = kLineThreads / 64;
constexpr uint32_t kLineSlots = 128;
constexpr uint32_t kLineRowBytes = kLineSlots * 2;
constexpr uint32_t kLineRowsBytes = kNumBuckets * kLineRowBytes;  // 128 KiB

template <int KB>
__device__ __forceinline__ void place16_lines(unsigned char *rows, uint32_t *pos, const uint64_t *gline,
                                              uint16_t *__restrict__ keys_out, const uint32_t (&v)[16], uint32_t valid)
{
    constexpr uint32_t kKeyMask = (1u << KB) - 1u;
    uint32_t slot[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t b4 = (v[j] >> (KB - 2)) & 0x7FCu;  // 4 * bucket
        slot[j] = atomicAdd((uint32_t *)((unsigned char *)pos + b4), (valid >> (15 - j)) & 1u);
    }
    uint32_t smax = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t b4 = (v[j] >> (KB - 2)) & 0x7FCu;
        const uint32_t x = slot[j] | (((~valid >> (15 - j)) & 1u) << 7);     // >= 128: not counted, or row full
        const uint32_t rot = (b4 << 2) & 0xF0u;                               // 16 * (bucket % 16): bank spread
        const uint32_t at = ((2u * slot[j] + rot) & 254u) | (b4 << 6);        // row base = bucket * 256
        *(uint16_t *)(rows + (x < kLineSlots ? at : kLineRowsBytes)) = (uint16_t)(v[j] & kKeyMask);
        smax = max(smax, slot[j]);
    }
    if (smax >= kLineSlots) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (((valid >> (15 - j)) & 1u) && slot[j] >= kLineSlots)
                keys_out[(gline[v[j] >> KB] & ~63ULL) + slot[j]] = (uint16_t)(v[j] & kKeyMask);
        }
    }
}

// Flush phase of one tile (see above).  Eight lanes share a row: lane = (row_in_group << 3) | piece,
// a piece is 8 slots = 16 bytes.  gline[row] holds the row's aligned base key index with lo (< 64)
// in its low six bits.  The common case -- exactly one complete line, lo == 0 -- is branch-free
// apart from the store predicate; everything else (first line of a segment, rows that overflowed,
// the final partial line) goes through flush_rows_slow under a wave-uniform test.
__device__ __forceinline__ void flush_rows_slow(unsigned char *rb, uint32_t rot, uint32_t piece, uint32_t n, uint64_t glw,
                                                uint32_t *pos_row, uint64_t *gline_row, uint16_t *__restrict__ keys_out,
                                                bool final)
{
    const uint32_t l0 = (uint32_t)(glw & 63);
    const uint64_t gl = glw & ~63ULL;
    const uint32_t L = n >> 6, r = n & 63;
    const uint4 line0 = *(const uint4 *)(rb + ((16u * piece + rot) & 255u));
    const uint4 line1 = *(const uint4 *)(rb + ((16u * (piece + 8) + rot) & 255u));
    if (L >= 1) {
        const uint16_t *k = (const uint16_t *)&line0;
#pragma unroll
        for (uint32_t e = 0; e < 8; ++e)
            if (8 * piece + e >= l0) keys_out[gl + 8 * piece + e] = k[e];
        if (L >= 2) *(uint4 *)(keys_out + gl + 64 + 8 * piece) = line1;
        if (L == 1 && 8 * piece < r) *(uint4 *)(rb + ((16u * piece + rot) & 255u)) = line1;   // leftover moves down
    }
    uint32_t new_n = n, new_lo = l0;
    uint64_t new_gl = gl;
    if (L >= 1) {
        new_n = r;
        new_lo = L >= 2 ? r : 0u;      // with L >= 2 the remainder went out directly
        new_gl = gl + 64ull * L;
    }
    if (final && new_n > new_lo && L <= 1) {
        // unfinished last line of this block's segment: slots [new_lo, new_n)
        const uint4 cur = (L == 1) ? line1 : line0;
        const uint16_t *k = (const uint16_t *)&cur;
#pragma unroll
        for (uint32_t e = 0; e < 8; ++e) {
            const uint32_t sl = 8 * piece + e;
            if (sl >= new_lo && sl < new_n) keys_out[new_gl + sl] = k[e];
        }
    }
    if (piece == 0) {
        *pos_row = new_n;
        *gline_row = new_gl | new_lo;
    }
}

__device__ __forceinline__ void flush_lines(unsigned char *rows, uint32_t *pos, uint64_t *gline,
                                            uint16_t *__restrict__ keys_out, bool final)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t piece = lane & 7;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint32_t row = wave * 32 + g * 8 + (lane >> 3);
        const uint32_t n = pos[row];
        const uint64_t glw = gline[row];
        const uint32_t rot = (row & 15u) << 4;
        unsigned char *rb = rows + row * kLineRowBytes;
        const bool fast = (n >> 6) == 1 && (glw & 63) == 0;
        const bool slow = !fast && ((n >> 6) >= 1 || final);
        if (__builtin_expect(__any(slow), 0)) {          // wave-uniform
            if (slow || fast) flush_rows_slow(rb, rot, piece, n, glw, &pos[row], &gline[row], keys_out, final);
            continue;
        }
        const uint32_t r = n & 63;
        if (fast) {   // only the lanes of flushing rows touch LDS (roughly half of them per tile)
            const uint4 line0 = *(const uint4 *)(rb + ((16u * piece + rot) & 255u));
            *(uint4 *)(keys_out + glw + 8 * piece) = line0;                     // one aligned 128-byte line per 8 lanes
            if (8 * piece < r) {                                                // leftover moves down
                const uint4 line1 = *(const uint4 *)(rb + ((16u * (piece + 8) + rot) & 255u));
                *(uint4 *)(rb + ((16u * piece + rot) & 255u)) = line1;
            }
            if (piece == 0) {
                pos[row] = r;
                gline[row] = glw + 64;
            }
        }
    }
}

// (lds_barrier: kpal_device.hpp)

// A3: scatter of an ASCII span.  Per tile (3 steps per wave, 24 KiB per workgroup): place, barrier,
// copy-out, barrier.  Bound by the global store-run rate (one ~88-byte run per bucket per tile).
template <int K>
__global__ __launch_bounds__(kScatterThreads) void part_scatter_kernel(Span s, uint64_t steps_per_block,
                                                                       const uint32_t *__restrict__ offs32,
                                                                       const uint64_t *__restrict__ bucket_start,
                                                                       uint16_t *__restrict__ keys_out)
{
    __shared__ __attribute__((aligned(16))) unsigned char rows[kRowsBytes + 16];
    __shared__ uint32_t pos[kNumBuckets];
    __shared__ uint64_t gcur[kNumBuckets];
    static_assert(kScatterThreads == kNumBuckets && kSlotCap == 64, "one lane per slot, one thread per bucket");
    const int wave = threadIdx.x >> 6;
    gcur[threadIdx.x] = bucket_start[threadIdx.x] + offs32[(uint64_t)threadIdx.x * gridDim.x + blockIdx.x];
    pos[threadIdx.x] = 0;
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
#pragma unroll
        for (int st = 0; st < kScatterSteps; ++st) part_step<K>(s, step0 + t + st, carry, window[st], mask[st]);
#pragma unroll
        for (int st = 0; st < kScatterSteps; ++st) {
            uint32_t v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = kmer_at<K>(window[st], j);
            place16<PartCfg<K>::kKeyBits>(rows, pos, gcur, keys_out, v, mask[st]);
        }
        __syncthreads();
        copy_out_tile(rows, pos, gcur, keys_out);
        __syncthreads();
    }
}

// B: histogram one slice of one bucket in LDS, merge into the table.
// grid = (kHistGridX, coarse buckets); block 1024 threads.  bucket_start has 513 entries per coarse
// bucket, slice_start (from the bucket scan) maps workgroup w to (bucket, slice); the table slice of
// coarse bucket c starts at c << (9 + KB).
// Skew: low-complexity sequence sends long runs of ONE key to a bucket, and 64 lanes adding to one
// LDS address serialise (~2 clk per lane).  Per 16-byte vector the wave picks a "hot" key (lane
// 0's first key, or the first dissenting lane's if that one is more frequent), counts its
// occurrences with ballots in a scalar register and diverts those lanes to private dummy words; one lane adds the
// scalar count.  Unskewed input pays two extra VALU per key (B is memory-bound).
template <int KB>
__global__ __launch_bounds__(1024) void part_hist_kernel(const uint16_t *__restrict__ keys,
                                                         const uint64_t *__restrict__ bucket_start,
                                                         const uint32_t *__restrict__ slice_start,
                                                         unsigned long long *__restrict__ table)
{
    constexpr int BINS = 1 << KB;
    __shared__ __attribute__((aligned(16))) uint32_t hist[BINS + 64];  // 128 KiB at KB = 15, + one dummy word per lane
    const uint32_t *ss = slice_start + (uint64_t)blockIdx.y * (kNumBuckets + 1);
    if (blockIdx.x >= ss[kNumBuckets]) return;
    uint32_t b = 0;   // largest b with ss[b] <= blockIdx.x (block-uniform binary search)
#pragma unroll
    for (int step = kNumBuckets / 2; step >= 1; step >>= 1)
        if (ss[b + step] <= blockIdx.x) b += step;
    const uint32_t sl = blockIdx.x - ss[b];
    const uint32_t slices = ss[b + 1] - ss[b];
    for (int i = threadIdx.x; i < BINS; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const uint64_t *bstart = bucket_start + (uint64_t)blockIdx.y * (kNumBuckets + 1);
    const uint64_t bs = bstart[b], be = bstart[b + 1];
    const uint64_t len = be - bs;
    const uint64_t per = (len + slices - 1) / slices;
    const uint64_t e0 = bs + min((uint64_t)sl * per, len);
    const uint64_t e1 = bs + min((uint64_t)(sl + 1) * per, len);
    const uint64_t a0 = min((e0 + 7) & ~7ULL, e1);  // head: up to the next 16-byte boundary (8 keys)
    for (uint64_t e = e0 + threadIdx.x; e < a0; e += blockDim.x) atomicAdd(&hist[keys[e]], 1u);
    const uint64_t a1 = a0 + ((e1 - a0) & ~7ULL);
    const uint4 *kv = reinterpret_cast<const uint4 *>(keys + a0);
    const uint64_t nvec = (a1 - a0) >> 3;
    const int lane = threadIdx.x & 63;
    auto add8 = [&](const uint4 q) {
        const uint32_t k[8] = {q.x & 0xFFFFu, q.x >> 16, q.y & 0xFFFFu, q.y >> 16,
                               q.z & 0xFFFFu, q.z >> 16, q.w & 0xFFFFu, q.w >> 16};
        // hot key: the first lane's first key, or -- if fewer than half of the lanes agree with it --
        // the first dissenting lane's, whichever is more frequent (all wave-uniform, scalar)
        const unsigned long long active = __builtin_amdgcn_ballot_w64(true);
        uint32_t hot = __builtin_amdgcn_readfirstlane(k[0]);
        const unsigned long long agree = __builtin_amdgcn_ballot_w64(k[0] == hot);
        if (__popcll(agree) < 32 && (active & ~agree)) {
            const uint32_t other = __builtin_amdgcn_readlane(k[0], __ffsll((long long)(active & ~agree)) - 1);
            if (__popcll(__builtin_amdgcn_ballot_w64(k[0] == other)) > __popcll(agree)) hot = other;
        }
        uint32_t same = 0;   // wave-uniform: stays in a scalar register
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool eq = k[j] == hot;
            same += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(eq));
            atomicAdd(&hist[eq ? (uint32_t)(BINS + lane) : k[j]], 1u);   // hot lanes: a private dummy word (no conflict)
        }
        if (lane == __ffsll((long long)active) - 1) atomicAdd(&hist[hot], same);
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
    unsigned long long *dst = table + ((uint64_t)blockIdx.y << (kPartBits + KB)) + ((uint64_t)b << KB);
    if (slices == 1) {
        // this workgroup is the only writer of the bucket's table slice during the launch: plain
        // coalesced read-modify-write instead of one atomic per non-zero bin
        for (int i = threadIdx.x; i < BINS; i += blockDim.x) {
            const uint32_t c = hist[i];
            if (c) dst[i] += (unsigned long long)c;
        }
    } else {
        for (int i = threadIdx.x; i < BINS; i += blockDim.x) {
            const uint32_t c = hist[i];
            if (c) atomicAdd(&dst[i], (unsigned long long)c);
        }
    }
}

// ==========================================================================================
// Two-level path, k = 13..16.
// ==========================================================================================
constexpr int kResidualBits = 24;        // residual = a k=12 k-mer
constexpr int kResKeyBits = kResidualBits - kPartBits;  // 15
constexpr int kCoarseSlots = 16384;      // u32 staging slots per workgroup (64 KiB)
constexpr int kCoarseThreads = 512;      // 8 waves, one step per wave per tile (8 KiB of input)

template <int K>
struct CoarseCfg {
    static constexpr int kBits = 2 * K - kResidualBits;  // 2, 4, 6, 8
    static constexpr int kBuckets = 1 << kBits;          // 4, 16, 64, 256
    // virtual rows of the staging = bucket x replica.  Replicas spread the LDS slot counters of
    // a few buckets over more addresses; a wave's rows always hold whole buckets.
    // (measured at k = 15: 64 rows x 256 slots 25.1 ms, 128 x 128 22.3 ms, 256 x 64 20.5 ms -- fewer
    // lanes per slot counter beats longer runs)
    static constexpr int kRowsAlloc = kBuckets >= 64 ? 256 : 64;   // 64, 64, 256, 256
    static constexpr int kRowsPerWave = kRowsAlloc / (kCoarseThreads / 64);   // 8, 8, 32, 32
    static constexpr int kRep = (kRowsAlloc / kBuckets) > kRowsPerWave ? kRowsPerWave : (kRowsAlloc / kBuckets);   // 8, 4, 4, 1
    static constexpr int kRows = kBuckets * kRep;        // 32, 64, 256, 256
    static constexpr int kCap = kCoarseSlots / kRowsAlloc;   // u32 slots per row: 256, 256, 64, 64
};

// C1: per-(coarse bucket, block) counts; cnt1[c * G + blk].  One step range per wave like A1.
template <int K>
__global__ __launch_bounds__(kCoarseThreads) void coarse_count_kernel(Span s, uint64_t steps_per_block,
                                                                      uint32_t *__restrict__ cnt1)
{
    constexpr int NB = CoarseCfg<K>::kBuckets;
    __shared__ uint32_t cnt[NB * 32];
    for (int i = threadIdx.x; i < NB * 32; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    const uint32_t rep = threadIdx.x & 31;
    const uint64_t total_steps = (s.nchunks + 63) / 64;
    const uint64_t steps_per_wave = steps_per_block / (kCoarseThreads / 64);
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
                atomicAdd(&cnt[(kmer_at<K>(window, j) >> kResidualBits) * 32 + rep], (mask >> (15 - j)) & 1u);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < NB; c += blockDim.x) {
        uint32_t v = 0;
#pragma unroll
        for (int r = 0; r < 32; ++r) v += cnt[c * 32 + ((r + c) & 31)];
        cnt1[(uint64_t)c * gridDim.x + blockIdx.x] = v;
    }
}

// C3: coarse scatter: 24-bit residuals (u32) into coarse-bucket-major order.  Per tile (one step
// per wave): every k-mer takes a slot in virtual row vrow = bucket * REP + lane % REP; slots < 256
// are staged; after the barrier wave w turns its 8 rows' counts into global positions (rows of
// one bucket are appended one after another), publishes them, stores the staged rows with
// range-checked buffer_store_dword, and -- only if some row overflowed (skewed input) -- the
// lanes that hold overflowed k-mers store them directly once the positions are published.
template <int K>
__global__ __launch_bounds__(kCoarseThreads) void coarse_scatter_kernel(Span s, uint64_t steps_per_block,
                                                                        const uint32_t *__restrict__ offs1,
                                                                        const uint64_t *__restrict__ start1,
                                                                        uint32_t *__restrict__ res_out)
{
    using CFG = CoarseCfg<K>;
    constexpr int NB = CFG::kBuckets, REP = CFG::kRep, ROWS = CFG::kRows, RALLOC = CFG::kRowsAlloc, CAP = CFG::kCap;
    constexpr int RPW = CFG::kRowsPerWave;
    __shared__ __attribute__((aligned(16))) uint32_t rows[kCoarseSlots + 4];  // 64 KiB + dummy
    __shared__ uint32_t pos[RALLOC];
    __shared__ uint64_t rowbase[RALLOC];  // global position of each virtual row's run in this tile
    __shared__ uint64_t gcur[RALLOC];     // per coarse bucket (first NB entries)
    __shared__ uint32_t overflowed;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < NB) gcur[threadIdx.x] = start1[threadIdx.x] + offs1[(uint64_t)threadIdx.x * gridDim.x + blockIdx.x];
    if (threadIdx.x < RALLOC) pos[threadIdx.x] = 0;
    if (threadIdx.x == 0) overflowed = 0;
    __syncthreads();
    const uint64_t total_steps = (s.nchunks + 63) / 64;
    const uint64_t steps_per_wave = steps_per_block / (kCoarseThreads / 64);
    const uint64_t block_step0 = (uint64_t)blockIdx.x * steps_per_block;
    const uint64_t step0 = block_step0 + (uint64_t)wave * steps_per_wave;
    Chunk carry = load_chunk(s, (int64_t)(step0 * 64) - 1);
    const uint32_t myrep = lane & (REP - 1);
    uint4 raw = fetch_chunk(s, (int64_t)(step0 * 64 + lane));   // fetched one tile ahead: the load flies during the copy-out
    for (uint64_t t = 0; t < steps_per_wave; ++t) {
        if (block_step0 + t >= total_steps) break;  // block-uniform
        uint64_t window;
        uint32_t mask;
        encode_step<K>(s, step0 + t, raw, carry, window, mask);
        uint32_t slot[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t vrow = (kmer_at<K>(window, j) >> kResidualBits) * REP + myrep;
            slot[j] = atomicAdd(&pos[vrow], (mask >> (15 - j)) & 1u);
        }
        uint32_t smax = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t kmer = kmer_at<K>(window, j);
            const uint32_t vrow = (kmer >> kResidualBits) * REP + myrep;
            const bool staged = ((mask >> (15 - j)) & 1u) && slot[j] < (uint32_t)CAP;
            const uint32_t at = staged ? vrow * CAP + ((slot[j] + vrow) & (CAP - 1)) : (uint32_t)kCoarseSlots;
            rows[at] = kmer & ((1u << kResidualBits) - 1u);
            smax = max(smax, slot[j]);
        }
        if (smax >= (uint32_t)CAP) overflowed = 1;   // benign race: every writer stores 1
        __syncthreads();
        if (t + 1 < steps_per_wave) raw = fetch_chunk(s, (int64_t)((step0 + t + 1) * 64 + lane));
        const uint32_t any_overflow = overflowed;            // stable until the barrier below
        // wave w owns virtual rows [RPW*w, RPW*w + RPW) = whole buckets; lanes 0..RPW-1 turn counts into positions
        {
            const int myrow = wave * RPW + (lane & (RPW - 1));
            const uint32_t n = (myrow < ROWS) ? pos[myrow] : 0u;
            uint32_t incl = n;                               // inclusive scan inside each group of REP lanes
#pragma unroll
            for (int d = 1; d < REP; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d);
                if ((lane & (REP - 1)) >= d) incl += o;
            }
            const int bucket = myrow / REP;
            const uint64_t g = (myrow < ROWS) ? gcur[bucket] + (incl - n) : 0ULL;
            if (lane < RPW && myrow < ROWS) {
                rowbase[myrow] = g;
                if ((lane & (REP - 1)) == REP - 1) gcur[bucket] += incl;   // last replica advances the bucket cursor
                pos[myrow] = 0;
            }
            const uint64_t addr = (uint64_t)res_out + 4ULL * g;
            const uint32_t a_lo = (uint32_t)addr, a_hi = (uint32_t)(addr >> 32);
            const uint32_t nbytes = 4u * min(n, (uint32_t)CAP);
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                const uint32_t lo = __builtin_amdgcn_readlane(a_lo, i);
                const uint32_t hi = __builtin_amdgcn_readlane(a_hi, i);
                const uint32_t nb = __builtin_amdgcn_readlane(nbytes, i);
                __amdgpu_buffer_rsrc_t rsrc =
                    __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), (short)0, (int)nb, 0x00020000);
                const uint32_t vrow = wave * RPW + i;
#pragma unroll
                for (int q = 0; q < CAP / 64; ++q) {
                    const uint32_t sl = q * 64 + lane;
                    const uint32_t val = rows[vrow * CAP + ((sl + vrow) & (CAP - 1))];
                    __builtin_amdgcn_raw_buffer_store_b32((int)val, rsrc, 4 * (int)sl, 0, 0);
                }
            }
        }
        __syncthreads();
        if (any_overflow) {   // block-uniform; skewed input only
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (((mask >> (15 - j)) & 1u) && slot[j] >= (uint32_t)CAP) {
                    const uint32_t kmer = kmer_at<K>(window, j);
                    const uint32_t vrow = (kmer >> kResidualBits) * REP + myrep;
                    res_out[rowbase[vrow] + slot[j]] = kmer & ((1u << kResidualBits) - 1u);
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) overflowed = 0;
            __syncthreads();
        }
    }
}

// Level 2 on residual streams.  blockIdx.y = coarse bucket c, whose residuals are
// res[start1[c] .. start1[c+1]).  Block g owns keys [g*KPB, (g+1)*KPB) of that stream (KPB keys per
// block, a multiple of 8 waves x 3 macro-steps x 1024 keys); a macro-step is 16 keys per lane
// (four coalesced 1 KiB loads per wave).
constexpr uint32_t kMacroKeys = 1024;   // keys per wave macro-step
constexpr uint32_t kKeysPerBlockQuantum = 16 * kMacroKeys;  // 16384: whole macro-steps for 8 (count) and 16 (scatter) waves

__device__ __forceinline__ void load_macro(const uint32_t *__restrict__ res, uint64_t lo, uint64_t hi, uint64_t at,
                                           uint32_t (&v)[16], uint32_t &valid)
{
    // lane's keys: at + q*256 + lane*4 + {0..3}, q = 0..3; v[j] index j = q*4 + e
    const int lane = threadIdx.x & 63;
    valid = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint64_t p = at + (uint64_t)q * 256 + (uint64_t)lane * 4;
        uint32_t e[4] = {0, 0, 0, 0};
        if (p + 4 <= hi) {
            // 4-byte aligned 16-byte load (the stream starts at an arbitrary key index)
            typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
            const u32x4_a4 q4 = *reinterpret_cast<const u32x4_a4 *>(res + lo + p);
            e[0] = q4.x;
            e[1] = q4.y;
            e[2] = q4.z;
            e[3] = q4.w;
            valid |= 0xFu << (12 - 4 * q);
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (p + t < hi) {
                    e[t] = res[lo + p + t];
                    valid |= 1u << (15 - (q * 4 + t));
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) v[q * 4 + t] = e[t];
    }
}

// cntmat2[c][b][g] (bucket-major inside each coarse bucket).
__global__ __launch_bounds__(kScatterThreads) void key_count_kernel(const uint32_t *__restrict__ res,
                                                                    const uint64_t *__restrict__ start1,
                                                                    uint32_t keys_per_block, uint32_t *__restrict__ cntmat2)
{
    __shared__ uint32_t cnt[kNumBuckets * kCountReplicas];
    const uint32_t c = blockIdx.y, G = gridDim.x;
    const uint64_t lo = start1[c], n = start1[c + 1] - lo;
    const uint64_t b0 = (uint64_t)blockIdx.x * keys_per_block;
    uint32_t *out = cntmat2 + ((uint64_t)c * kNumBuckets) * G + blockIdx.x;
    if (b0 >= n) {   // block-uniform: nothing of this stream falls into the block
        for (int b = threadIdx.x; b < kNumBuckets; b += blockDim.x) out[(uint64_t)b * G] = 0;
        return;
    }
    for (int i = threadIdx.x; i < kNumBuckets * kCountReplicas; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    const uint32_t rep = threadIdx.x & (kCountReplicas - 1);
    const uint64_t per_wave = keys_per_block / kScatterWaves;
    const uint64_t w0 = b0 + (uint64_t)(threadIdx.x >> 6) * per_wave;
    for (uint64_t at = w0; at < w0 + per_wave && at < n; at += kMacroKeys) {
        uint32_t v[16], valid;
        load_macro(res, lo, n, at, v, valid);
#pragma unroll
        for (int j = 0; j < 16; ++j) atomicAdd(&cnt[(v[j] >> kResKeyBits) * kCountReplicas + rep], (valid >> (15 - j)) & 1u);
    }
    __syncthreads();
    flush_bucket_counts(cnt, out, G);
}

__global__ __launch_bounds__(kLineThreads) void key_scatter_kernel(const uint32_t *__restrict__ res,
                                                                   const uint64_t *__restrict__ start1,
                                                                   uint32_t keys_per_block,
                                                                   const uint32_t *__restrict__ offs2,
                                                                   const uint64_t *__restrict__ bucket_start2,
                                                                   uint16_t *__restrict__ keys_out)
{
    __shared__ __attribute__((aligned(16))) unsigned char rows[kLineRowsBytes + 16];
    __shared__ uint32_t pos[kNumBuckets];
    __shared__ uint64_t gline[kNumBuckets];
    const uint32_t c = blockIdx.y, G = gridDim.x;
    const uint64_t lo = start1[c], n = start1[c + 1] - lo;
    const uint64_t b0 = (uint64_t)blockIdx.x * keys_per_block;
    if (b0 >= n) return;  // block-uniform
    if (threadIdx.x < kNumBuckets) {
        const uint64_t g0 = bucket_start2[(uint64_t)c * (kNumBuckets + 1) + threadIdx.x] +
                            offs2[((uint64_t)c * kNumBuckets + threadIdx.x) * G + blockIdx.x];
        gline[threadIdx.x] = g0;
        pos[threadIdx.x] = (uint32_t)(g0 & 63);
    }
    __syncthreads();
    const uint64_t per_wave = keys_per_block / kLineWaves;
    const uint64_t w0 = b0 + (uint64_t)(threadIdx.x >> 6) * per_wave;
    uint32_t v[16], valid, vn[16], validn;
    load_macro(res, lo, n, w0, v, valid);
    for (uint64_t t = 0; t < per_wave; t += kMacroKeys) {
        if (b0 + t >= n) break;  // block-uniform: wave 0 owns the lowest keys
        const bool more = t + kMacroKeys < per_wave;
        if (more) load_macro(res, lo, n, w0 + t + kMacroKeys, vn, validn);   // next tile's keys land during placement
        place16_lines<kResKeyBits>(rows, pos, gline, keys_out, v, valid);
        lds_barrier();
        flush_lines(rows, pos, gline, keys_out, false);
        if (more) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = vn[j];
            valid = validn;
        }
        lds_barrier();
    }
    flush_lines(rows, pos, gline, keys_out, true);
}

}  // namespace kpal
