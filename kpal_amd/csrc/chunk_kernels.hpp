// chunk_kernels.hpp -- one-pass radix partition into CHUNKED bucket lists (k = 8..12, gfx950).
//
// The exact-offset pipeline of partition_kernels.hpp reads the input twice: a counting pass (A1)
// exists only to give every (bucket, workgroup) run its final position.  Here the scatter kernel
// needs no positions: a bucket's keys go into 8 KiB chunks.  Workgroup g owns the chunk ids
// [g*R, (g+1)*R), R = steps per workgroup / 4 + 1024 + margin -- enough in EVERY case: at most
// steps/4 full chunks (a step is 1024 k-mers, a chunk 4096 keys), one partly filled and one
// pre-assigned next chunk per bucket -- so allocation is an LDS counter, never a global atomic.  A retired chunk is recorded
// as one word (chunk id | fill-1 << 20) in the table row of its (bucket, workgroup); rows hold four
// entries, further ones (skewed input only) go to an overflow list that is grouped by bucket.
// The histogram kernel walks its bucket's table rows and overflow entries.  Still exact, still
// deterministic in its result (integer adds commute), no capacity estimates, and skew needs no
// special layout: a bucket that receives everything simply owns more chunks.
//
//   C3  chunk_scatter_kernel<K>   ASCII -> 16-bit keys in chunks (LDS staging as in A3)
//   C3' chunk_key_lines_kernel    level 2 of k = 13..16: 24-bit residuals -> keys in chunks, aligned 128-byte lines
//       chunk_key_scatter_kernel  the same with per-tile runs (KPAL_LEVEL2=1)
//   C4a chunk_plan_kernel         scan of the overflow counts per bucket, slice plan for C5
//   C4b chunk_list_kernel         overflow entries grouped by bucket (empty for unskewed input)
//   C5  chunk_hist_kernel<KB>     LDS histogram over a bucket's chunks, merge into the table
#pragma once
#include "partition_kernels.hpp"

namespace kpal {

constexpr uint32_t kChunkShift = 12;
constexpr uint32_t kChunkKeys = 1u << kChunkShift;   // 4096 keys = 8 KiB
constexpr uint32_t kChunkRow = 4;                    // table entries per (bucket, workgroup)
constexpr uint32_t kChunkEmpty = 0xFFFFFFFFu;
constexpr uint32_t kChunkDeferCap = 256;             // runs of consecutive abandoned tiles a workgroup can remember

// Abandoned tiles are remembered as runs (first tile, count): skew comes in stretches.
struct DeferRun {
    uint32_t first, count;
};
__device__ __forceinline__ void defer_tile(DeferRun *runs, uint32_t &n_runs, uint32_t tile, uint32_t *error)
{
    if (n_runs && runs[n_runs - 1].first + runs[n_runs - 1].count == tile) {
        ++runs[n_runs - 1].count;
    } else if (n_runs < kChunkDeferCap) {
        runs[n_runs].first = tile;
        runs[n_runs].count = 1;
        ++n_runs;
    } else {
        *error = 1u;
    }
}
constexpr uint32_t kChunkIdBits = 20;                // chunk ids < 2^20: key indices fit 32 bits; entry = id | (fill-1) << 20

// All arrays carry a leading coarse-bucket dimension Y (1 for the one-level path; blockIdx.y
// selects it), so chunk ids -- relative to the coarse bucket's part of the pool -- stay below 2^20.
// Bucket scrambling.  A value v = (bucket : 9 bits | key : KB bits) is scattered into bucket
// b' = bucket ^ g(top six bits of key), g(t) = t << 3 | t: the three bases after the bucket prefix decide which of 64
// different buckets a prefix maps to, so compositional skew (AT-rich genomes: bucket prefixes differ
// threefold in frequency) is averaged over 64 buckets, while the bins of a scrambled bucket still
// come in runs of 2^(KB-6) consecutive table entries (for a fixed key top the map is a bijection of
// the buckets, so every workgroup of the histogram stage still owns its bins exclusively).
#if defined(KPAL_NO_SCRAMBLE)   // A/B builds only
__device__ __forceinline__ uint32_t chunk_bucket_mask(uint32_t) { return 0u; }
#elif defined(KPAL_SCRAMBLE_MUL)
__device__ __forceinline__ uint32_t chunk_bucket_mask(uint32_t key_top6) { return (key_top6 * 73u) & 511u; }
#else
__device__ __forceinline__ uint32_t chunk_bucket_mask(uint32_t key_top6) { return (key_top6 << 3) | key_top6; }   // 64 distinct 9-bit masks, one v_lshl_or
#endif

template <int KB>
__device__ __forceinline__ uint32_t chunk_scramble(uint32_t v)
{
    static_assert(KB >= 6, "needs six key bits");
    return v ^ (chunk_bucket_mask((v >> (KB - 6)) & 63u) << KB);
}

struct ChunkPool {
    uint16_t *keys;        // [Y][G * per_block chunks][kChunkKeys keys]
    uint32_t per_block;    // R: chunk ids of workgroup g are [g*R, (g+1)*R)
    uint32_t groups;       // G: scatter workgroups per coarse bucket
    uint32_t *table;       // [Y][512][G][kChunkRow] entries (chunk id | (fill-1) << 20), kChunkEmpty = none
    uint32_t *nlist;       // [Y][512] retired chunks per bucket (slice plan)
    uint32_t *ovf_n;       // [Y][512] overflow entries per bucket
    uint32_t *ovf_count;   // [Y] overflow entries
    uint2 *ovf;            // [Y][G * per_block] overflow entries (bucket, entry)
    uint32_t *error;       // set if a workgroup ran out of chunks (cannot happen: R is a worst-case bound)
};
__device__ __forceinline__ uint64_t chunk_pool_chunks(const ChunkPool &p) { return (uint64_t)p.groups * p.per_block; }
// The scatter kernel gets the pool by pointer (device memory): its rarely used fields must not sit
// in scalar registers for the whole kernel.

__device__ __forceinline__ uint32_t chunk_entry(uint32_t cid, uint32_t fill) { return cid | ((fill - 1u) << kChunkIdBits); }

// Workgroup-local allocation of n contiguous chunks.
__device__ __forceinline__ uint32_t chunk_alloc(const ChunkPool *p, uint32_t per_block, uint32_t *alloc_next, uint32_t n)
{
    uint32_t id = atomicAdd(alloc_next, n);
    const uint32_t end = (blockIdx.x + 1u) * per_block;
    if (id + n > end) {   // keep every write inside the workgroup's range, report
        *p->error = 1u;
        id = end - n;
    }
    return id;
}

// Record a finished chunk of bucket b; `nret` counts this (bucket, workgroup)'s chunks so far.
__device__ __forceinline__ uint32_t *chunk_table_row(const ChunkPool *p, uint32_t b)
{
    return p->table + (((uint64_t)blockIdx.y * kNumBuckets + b) * gridDim.x + blockIdx.x) * kChunkRow;
}

__device__ __forceinline__ void chunk_retire(const ChunkPool *p, uint32_t cid, uint32_t b, uint32_t fill, uint32_t &nret)
{
    const uint32_t e = chunk_entry(cid, fill);
    if (nret < kChunkRow) {
        chunk_table_row(p, b)[nret] = e;
    } else {   // skewed input only
        const uint32_t at = atomicAdd(&p->ovf_count[blockIdx.y], 1u);
        p->ovf[(uint64_t)blockIdx.y * chunk_pool_chunks(*p) + at] = make_uint2(b, e);
        atomicAdd(&p->ovf_n[blockIdx.y * kNumBuckets + b], 1u);
    }
    ++nret;
    atomicAdd(&p->nlist[blockIdx.y * kNumBuckets + b], 1u);
}

// Placement of 16 values into the per-tile rows (layout of place16).  A counted value whose row is
// full (slot >= 64: a few per tile even for uniform input) is stored directly at its final place:
// slot s of bucket b lies at gcur[b] + s while that is inside the bucket's current chunk, else in
// its pre-assigned next chunk.  Slots >= kChunkKeys cannot be placed (the bucket would need a third
// chunk in this tile): the caller abandons such a tile.  Returns the largest counted slot.
template <int KB>
__device__ __forceinline__ uint32_t place16_chunked(unsigned char *rows, uint32_t *pos, const uint32_t *gcur,
                                                    const uint32_t *nextc, uint16_t *__restrict__ keys,
                                                    const uint32_t (&v)[16], uint32_t valid)
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
        const uint32_t counted = (valid >> (15 - j)) & 1u;
        const uint32_t x = slot[j] | ((counted ^ 1u) << 16);                 // >= 64: not counted, or row full
        const uint32_t at = ((2u * slot[j] + b4) & 126u) | (b4 << 5);        // byte offset of the rotated slot
        *(uint16_t *)(rows + (x < (uint32_t)kSlotCap ? at : kRowsBytes)) = (uint16_t)(v[j] & kKeyMask);
        smax = max(smax, counted ? slot[j] : 0u);
    }
    if (smax >= (uint32_t)kSlotCap) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (((valid >> (15 - j)) & 1u) && slot[j] >= (uint32_t)kSlotCap && slot[j] < kChunkKeys) {
                const uint32_t b = v[j] >> KB;
                const uint32_t g = gcur[b];
                const uint32_t room = kChunkKeys - (g & (kChunkKeys - 1));
                const uint32_t at = slot[j] < room ? g + slot[j] : (nextc[b] << kChunkShift) + (slot[j] - room);
                keys[at] = (uint16_t)(v[j] & kKeyMask);
            }
        }
    }
    return smax;
}

// Copy-out of the staged rows of one tile.  Thread t owns bucket t
// (wave w: buckets [64w, 64w+64)).  n keys of the bucket go to key index g, g+1, ... of its current
// chunk; those beyond the chunk's end (`room`) go to key index second, second+1, ...  Only the
// staged part [0, min(n, 64)) is written here, as one range-checked buffer_store_short per bucket
// plus a second one for the rare bucket whose staged run crosses the chunk end.
__device__ __forceinline__ void chunk_store_rows(const unsigned char *rows, uint16_t *__restrict__ keys, uint32_t n,
                                                 uint32_t g, uint32_t second)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int first = wave * kBucketsPerWave;
    uint32_t r0 = 2u * lane + 4u * first;
    // The 32 rotated row offsets below are the same for every tile; hoisted out of the tile loop
    // they would be spilled and re-loaded from scratch per tile (and the s_waitcnt vmcnt of those
    // re-loads would also wait for the caller's prefetched chunks).  Two VALU ops each instead.
    asm volatile("" : "+v"(r0));
    const unsigned char *wrows = rows + (uint32_t)first * 128u;
    unsigned long long cross;
    {
        const uint32_t staged = min(n, (uint32_t)kSlotCap);
        const uint32_t room = kChunkKeys - (g & (kChunkKeys - 1));
        const uint64_t addr = (uint64_t)keys + 2ULL * g;
        const uint32_t my_lo = (uint32_t)addr, my_hi = (uint32_t)(addr >> 32);
        const uint32_t my_bytes = 2u * min(staged, room);
        cross = __builtin_amdgcn_ballot_w64(staged > room);
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
    }
    // staged runs that cross the end of their chunk: slots [room, staged) continue at `second`
    while (cross) {   // wave-uniform; everything below is scalar
        const int i = __ffsll((long long)cross) - 1;
        cross &= cross - 1;
        const uint32_t gi = __builtin_amdgcn_readlane(g, i);
        const uint32_t rm = kChunkKeys - (gi & (kChunkKeys - 1));
        const uint32_t st = min(__builtin_amdgcn_readlane(n, i), (uint32_t)kSlotCap);
        // slot s lands at second + (s - room)
        const uint32_t sec = (uint32_t)__builtin_amdgcn_readlane(second, i);   // (the builtin returns int: no sign extension)
        const uint64_t addr2 = (uint64_t)keys + 2ULL * sec - 2ULL * rm;
        const uint16_t v = *(const uint16_t *)(wrows + i * 128 + ((r0 + 4u * i) & 126u));
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)addr2, (short)0, (int)(2u * st), 0x00020000);
        // lanes below `room` were written by the first store: send them out of range
        __builtin_amdgcn_raw_buffer_store_b16((short)v, rsrc, (uint32_t)lane >= rm ? 2 * lane : 0x7FFFFFF0, 0, 0);
    }
}

// Copy-out phase of a normal tile: thread t owns bucket t.  Stores the staged keys, advances the
// cursor, and when the current chunk is full retires it, continues in the pre-assigned next chunk
// and takes another one from the workgroup's range.
__device__ __forceinline__ void chunk_finish_tile(const unsigned char *rows, uint32_t *pos, uint32_t *gcur, uint32_t *nextc,
                                                  uint32_t *alloc_next, const ChunkPool *p, uint32_t per_block,
                                                  uint16_t *__restrict__ keys, uint32_t &nret)
{
    const uint32_t mine = threadIdx.x;
    const uint32_t n = pos[mine], g = gcur[mine], nx = nextc[mine];
    chunk_store_rows(rows, keys, n, g, nx << kChunkShift);
    const uint32_t room = kChunkKeys - (g & (kChunkKeys - 1));
    uint32_t g2 = g + n;
    if (n >= room) {
        chunk_retire(p, g >> kChunkShift, mine, kChunkKeys, nret);
        g2 = (nx << kChunkShift) + (n - room);
        nextc[mine] = chunk_alloc(p, per_block, alloc_next, 1);
    }
    gcur[mine] = g2;
    pos[mine] = 0;
}

// Abandoned tile (a bucket would have needed a third chunk: more than 4096 of the tile's 24576
// k-mers in one bucket, i.e. homopolymer-like input under several waves at once): its k-mers are
// counted straight into the table.  Per step the wave counts the occurrences of its first k-mer with
// ballots into a pending (k-mer, count) pair that is flushed with one global atomic only when the
// hot k-mer changes; everything else takes one global atomic per k-mer.
template <int K>
__device__ __forceinline__ void chunk_count_tile_direct(const Span &s, uint64_t first_step,
                                                        unsigned long long *__restrict__ table, uint32_t &pend_hot,
                                                        unsigned long long &pend_cnt)
{
    const int lane = threadIdx.x & 63;
    Chunk carry = load_chunk(s, (int64_t)(first_step * 64) - 1);
    for (int st = 0; st < kScatterSteps; ++st) {
        uint64_t window;
        uint32_t mask;
        part_step<K>(s, first_step + st, carry, window, mask);
        // hot k-mer: the first counted one of the lowest lane that has any
        const unsigned long long have = __builtin_amdgcn_ballot_w64(mask != 0);
        if (!have) continue;   // wave-uniform
        const int src = __ffsll((long long)have) - 1;
        const uint32_t m0 = __builtin_amdgcn_readlane(mask, src);
        const uint32_t w0_hi = (uint32_t)__builtin_amdgcn_readlane((uint32_t)(window >> 32), src);   // (returns int)
        const uint32_t w0_lo = (uint32_t)__builtin_amdgcn_readlane((uint32_t)window, src);
        const uint64_t w0 = ((uint64_t)w0_hi << 32) | w0_lo;
        const uint32_t hot = kmer_at<K>(w0, 15 - (31 - __clz(m0)));   // highest set bit of m0 = smallest j
        uint32_t same = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t v = kmer_at<K>(window, j);
            const bool counted = (mask >> (15 - j)) & 1u;
            const bool eq = counted && v == hot;
            same += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(eq));
            if (counted && !eq) atomicAdd(&table[v], 1ULL);
        }
        if (hot != pend_hot) {   // wave-uniform
            if (pend_cnt && lane == 0) atomicAdd(&table[pend_hot], pend_cnt);
            pend_hot = hot;
            pend_cnt = 0;
        }
        pend_cnt += same;
    }
}

template <int K>
__global__ __launch_bounds__(kScatterThreads, 4) void chunk_scatter_kernel(Span s, uint64_t steps_per_block,
                                                                           const ChunkPool *__restrict__ p,
                                                                           uint16_t *__restrict__ keys, uint32_t per_block,
                                                                           unsigned long long *__restrict__ table)
{
    constexpr int KB = PartCfg<K>::kKeyBits;
    __shared__ __attribute__((aligned(16))) unsigned char rows[kRowsBytes + 16];
    __shared__ uint32_t pos[kNumBuckets];
    __shared__ uint32_t gcur[kNumBuckets];    // key index (chunk id << 12 | offset) of every bucket's cursor
    __shared__ uint32_t nextc[kNumBuckets];   // every bucket's pre-assigned next chunk
    __shared__ uint32_t tile_over, alloc_next, defer_n;
    __shared__ DeferRun defer_t[kChunkDeferCap];   // abandoned tiles, counted directly after the main loop
    static_assert(kScatterThreads == kNumBuckets && kSlotCap == 64, "one lane per slot, one thread per bucket");
    const int wave = threadIdx.x >> 6;
    const uint32_t mine = threadIdx.x;        // the bucket this thread owns in the copy-out phase
    const uint32_t first_chunk = blockIdx.x * per_block;
    if (threadIdx.x == 0) {
        alloc_next = first_chunk + 2 * kNumBuckets;   // a current and a next chunk per bucket are pre-assigned
        tile_over = 0;
        defer_n = 0;
    }
    gcur[mine] = (first_chunk + mine) << kChunkShift;
    nextc[mine] = first_chunk + kNumBuckets + mine;
    uint32_t nret = 0;                        // chunks of (mine, this workgroup) retired so far
    pos[mine] = 0;
    __syncthreads();
    const uint64_t total_steps = (s.nchunks + 63) / 64;
    // Tile j of this workgroup is tile j * G + blockIdx.x of the input, wave w takes its steps 3w .. 3w+2: at any
    // moment the whole grid reads one sliding window of G * 24 KiB (12 MiB) instead of 4096 separate streams
    // spread over the whole piece -- a handful of pages instead of hundreds, which the scatter's own 2500 pages of
    // open chunks leave little translation-cache room for (piece sizes whose streams happened to collide there ran
    // up to 1.6x slower).  The price: the chunk left of a tile is fetched per tile (one uniform 16-byte load).
    const uint64_t tiles_per_block = steps_per_block / (kScatterWaves * kScatterSteps);
    auto tile_step = [&](uint64_t j) -> uint64_t {
        return ((j * gridDim.x + blockIdx.x) * kScatterWaves + (uint64_t)wave) * kScatterSteps;
    };
    const int lane = threadIdx.x & 63;
    uint4 raw[kScatterSteps];
    uint4 rawh;   // the 16 bytes left of the tile's first chunk (same for every lane)
    {
        const uint64_t f = tile_step(0);
#pragma unroll
        for (int st = 0; st < kScatterSteps; ++st) raw[st] = fetch_chunk(s, (int64_t)((f + st) * 64 + lane));
        rawh = fetch_chunk(s, (int64_t)(f * 64) - 1);
    }
    for (uint64_t j = 0; j < tiles_per_block; ++j) {
        const uint64_t first = tile_step(j);
        if ((j * gridDim.x + blockIdx.x) * (uint64_t)(kScatterWaves * kScatterSteps) >= total_steps) break;  // block-uniform
        Chunk carry = encode16(rawh);
        range_fix(s, (int64_t)(first * 64) - 1, carry);
        uint64_t window[kScatterSteps];
        uint32_t mask[kScatterSteps];
#pragma unroll
        for (int st = 0; st < kScatterSteps; ++st) encode_step<K>(s, first + st, raw[st], carry, window[st], mask[st]);
        uint32_t smax = 0;
#pragma unroll
        for (int st = 0; st < kScatterSteps; ++st) {
            uint32_t v[16];
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) v[jj] = chunk_scramble<KB>(kmer_at<K>(window[st], jj));
            smax = max(smax, place16_chunked<KB>(rows, pos, gcur, nextc, keys, v, mask[st]));
        }
        if (smax >= kChunkKeys) tile_over = 1;   // benign race: every writer stores 1
        __syncthreads();
        if (j + 1 < tiles_per_block) {           // the next tile's chunks fly during the copy-out
            const uint64_t f = tile_step(j + 1);
#pragma unroll
            for (int st = 0; st < kScatterSteps; ++st) raw[st] = fetch_chunk(s, (int64_t)((f + st) * 64 + lane));
            rawh = fetch_chunk(s, (int64_t)(f * 64) - 1);
        }
        if (tile_over) {   // block-uniform, pathological input only: forget the tile, count it after the loop
            pos[mine] = 0;
            __syncthreads();
            if (threadIdx.x == 0) {
                tile_over = 0;
                defer_tile(defer_t, defer_n, (uint32_t)j, p->error);
            }
            __syncthreads();
            continue;
        }
        chunk_finish_tile(rows, pos, gcur, nextc, &alloc_next, p, per_block, keys, nret);
        __syncthreads();
    }
    // the partly filled current chunks, then the unused row entries
    __syncthreads();
    const uint32_t g = gcur[mine];
    if (g & (kChunkKeys - 1)) chunk_retire(p, g >> kChunkShift, mine, g & (kChunkKeys - 1), nret);
    for (uint32_t e = nret; e < kChunkRow; ++e) chunk_table_row(p, mine)[e] = kChunkEmpty;
    uint32_t pend_hot = 0;
    unsigned long long pend_cnt = 0;
    for (uint32_t i = 0; i < defer_n; ++i)
        for (uint32_t q = 0; q < defer_t[i].count; ++q)
            chunk_count_tile_direct<K>(s, tile_step((uint64_t)defer_t[i].first + q), table, pend_hot, pend_cnt);
    if (pend_cnt && (threadIdx.x & 63) == 0) atomicAdd(&table[pend_hot], pend_cnt);
}

// Level 2 of the two-level path (k = 13..16) as a chunked scatter: blockIdx.y = coarse bucket c, whose
// 24-bit residuals (9-bit bucket | 15-bit key) are res[start1[c] .. start1[c+1]); workgroup g takes
// residuals [g*KPB, (g+1)*KPB) of that stream, a wave-step is 1024 residuals (load_macro).  Same
// staging, chunk logic and table rows as chunk_scatter_kernel -- no counting pass over the residuals.
template <int KB>
__device__ __forceinline__ void chunk_count_keys_direct(const uint32_t *__restrict__ res, uint64_t lo, uint64_t n, uint64_t at,
                                                        unsigned long long *__restrict__ table_c, int steps, uint32_t &pend_hot,
                                                        unsigned long long &pend_cnt)
{
    for (int st = 0; st < steps; ++st) {
        uint32_t v[16], valid;
        load_macro(res, lo, n, at + (uint64_t)st * kMacroKeys, v, valid);
        // Per position j the wave looks for a frequent residual among the lanes 0, 8, .., 56 (a
        // low-complexity stretch of >= 128 residuals covers eight consecutive lanes, so it is sampled)
        // and counts its occurrences with a ballot into the pending (residual, count) pair, which is
        // flushed with one global atomic when the residual changes; the other lanes add individually.
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const bool counted = (valid >> (15 - j)) & 1u;
            uint32_t hot = pend_hot;
            uint32_t best = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(counted && v[j] == pend_hot));
            if (best < 32)   // wave-uniform: the pending residual does not dominate this position
#pragma unroll
            for (int l = 0; l < 64; l += 8) {
                const uint32_t cand = (uint32_t)__builtin_amdgcn_readlane(v[j], l);
                const uint32_t cnt = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(counted && v[j] == cand));
                if (cnt > best) {   // wave-uniform
                    best = cnt;
                    hot = cand;
                }
            }
            const bool eq = counted && v[j] == hot;
            const uint32_t same = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(eq));
            if (counted && !eq) atomicAdd(&table_c[v[j]], 1ULL);
            if (same) {
                if (hot != pend_hot) {
                    if (pend_cnt && (threadIdx.x & 63) == 0) atomicAdd(&table_c[pend_hot], pend_cnt);
                    pend_hot = hot;
                    pend_cnt = 0;
                }
                pend_cnt += same;
            }
        }
    }
}

__global__ __launch_bounds__(kScatterThreads, 4) void chunk_key_scatter_kernel(const uint32_t *__restrict__ res,
                                                                               const uint64_t *__restrict__ start1,
                                                                               uint32_t keys_per_block,
                                                                               const ChunkPool *__restrict__ p,
                                                                               uint16_t *__restrict__ keys_base, uint32_t per_block,
                                                                               unsigned long long *__restrict__ table)
{
    constexpr int KB = kResKeyBits;
    __shared__ __attribute__((aligned(16))) unsigned char rows[kRowsBytes + 16];
    __shared__ uint32_t pos[kNumBuckets];
    __shared__ uint32_t gcur[kNumBuckets];
    __shared__ uint32_t nextc[kNumBuckets];
    __shared__ uint32_t tile_over, alloc_next, defer_n;
    __shared__ DeferRun defer_t[kChunkDeferCap];
    const int wave = threadIdx.x >> 6;
    const uint32_t mine = threadIdx.x;
    const uint32_t c = blockIdx.y;
    const uint64_t lo = start1[c], n = start1[c + 1] - lo;
    uint16_t *keys = keys_base + (((uint64_t)c * gridDim.x * per_block) << kChunkShift);
    const uint32_t first_chunk = blockIdx.x * per_block;
    if (threadIdx.x == 0) {
        alloc_next = first_chunk + 2 * kNumBuckets;
        tile_over = 0;
        defer_n = 0;
    }
    gcur[mine] = (first_chunk + mine) << kChunkShift;
    nextc[mine] = first_chunk + kNumBuckets + mine;
    uint32_t nret = 0;
    pos[mine] = 0;
    __syncthreads();
    const uint64_t b0 = (uint64_t)blockIdx.x * keys_per_block;
    const uint64_t per_wave = keys_per_block / kScatterWaves;
    const uint64_t w0 = b0 + (uint64_t)wave * per_wave;
    for (uint64_t t = 0; t < per_wave; t += (uint64_t)kScatterSteps * kMacroKeys) {
        if (b0 + t >= n) break;  // block-uniform: wave 0 owns the lowest residuals (also skips empty workgroups)
        uint32_t smax = 0;
#pragma unroll
        for (int st = 0; st < kScatterSteps; ++st) {
            uint32_t v[16], valid;
            load_macro(res, lo, n, w0 + t + (uint64_t)st * kMacroKeys, v, valid);
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = chunk_scramble<KB>(v[j]);
            smax = max(smax, place16_chunked<KB>(rows, pos, gcur, nextc, keys, v, valid));
        }
        if (smax >= kChunkKeys) tile_over = 1;
        __syncthreads();
        if (tile_over) {   // block-uniform, pathological input only
            pos[mine] = 0;
            __syncthreads();
            if (threadIdx.x == 0) {
                tile_over = 0;
                defer_tile(defer_t, defer_n, (uint32_t)(t / ((uint64_t)kScatterSteps * kMacroKeys)), p->error);
            }
            __syncthreads();
            continue;
        }
        chunk_finish_tile(rows, pos, gcur, nextc, &alloc_next, p, per_block, keys, nret);
        __syncthreads();
    }
    __syncthreads();
    const uint32_t g = gcur[mine];
    if (g & (kChunkKeys - 1)) chunk_retire(p, g >> kChunkShift, mine, g & (kChunkKeys - 1), nret);
    for (uint32_t e = nret; e < kChunkRow; ++e) chunk_table_row(p, mine)[e] = kChunkEmpty;
    uint32_t pend_hot = 0;
    unsigned long long pend_cnt = 0;
    unsigned long long *table_c = table + ((uint64_t)c << kResidualBits);
    for (uint32_t i = 0; i < defer_n; ++i)
        for (uint32_t q = 0; q < defer_t[i].count; ++q)
            chunk_count_keys_direct<KB>(res, lo, n, w0 + (uint64_t)(defer_t[i].first + q) * kScatterSteps * kMacroKeys, table_c,
                                        kScatterSteps, pend_hot, pend_cnt);
    if (pend_cnt && (threadIdx.x & 63) == 0) atomicAdd(&table_c[pend_hot], pend_cnt);
}

// ------------------------------------------------------------------------------------------
// Level 2 with aligned-line staging AND chunks (default for k = 13..16).  key_scatter_kernel's
// persistent 128-slot rows, from which only whole aligned 128-byte lines leave the CU, need exact
// positions only to know where a row's lines go; with chunks a row simply fills its bucket's
// current 8 KiB chunk line by line (a chunk is 64 lines, so a line never straddles two chunks)
// and moves to the pre-assigned next chunk when it is full -- the counting pass over the
// residuals (key_count_kernel, 54 GB of reads at k = 15) disappears, and so do the partial first
// lines of the exact layout.
//   gl[b]   : key index of row slot 0 (64-aligned) | lo (slots < lo of the current line are already
//             in memory: they left directly when the row overflowed)
//   keep[b] : keys left in the row after the last flush (restored if a tile is abandoned)
// ------------------------------------------------------------------------------------------
// key index of row slot `slot` of a row whose slot 0 sits at `base` (64-aligned) in its chunk
__device__ __forceinline__ uint32_t line_slot_index(uint32_t base, uint32_t next, uint32_t slot)
{
    const uint32_t room = kChunkKeys - (base & (kChunkKeys - 1));
    return slot < room ? base + slot : (next << kChunkShift) + (slot - room);
}

template <int KB>
__device__ __forceinline__ uint32_t place16_lines_chunked(unsigned char *rows, uint32_t *pos, const uint32_t *gl,
                                                          const uint32_t *nextc, uint16_t *__restrict__ keys,
                                                          const uint32_t (&v)[16], uint32_t valid)
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
        const uint32_t counted = (valid >> (15 - j)) & 1u;
        const uint32_t x = slot[j] | ((counted ^ 1u) << 16);                  // >= 128: not counted, or row full
        const uint32_t rot = (b4 << 2) & 0xF0u;                               // 16 * (bucket % 16): bank spread
        const uint32_t at = ((2u * slot[j] + rot) & 254u) | (b4 << 6);        // row base = bucket * 256
        *(uint16_t *)(rows + (x < kLineSlots ? at : kLineRowsBytes)) = (uint16_t)(v[j] & kKeyMask);
        smax = max(smax, counted ? slot[j] : 0u);
    }
    if (smax >= kLineSlots) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (((valid >> (15 - j)) & 1u) && slot[j] >= kLineSlots && slot[j] < kChunkKeys) {
                const uint32_t b = v[j] >> KB;
                keys[line_slot_index(gl[b] & ~63u, nextc[b], slot[j])] = (uint16_t)(v[j] & kKeyMask);
            }
        }
    }
    return smax;
}

// Flush phase of one tile; eight lanes share a row (lane = row_in_group << 3 | piece, a piece is 8
// slots = 16 bytes).  Common case: exactly one complete line and lo == 0 -> one aligned 128-byte
// store per row.  Everything else (rows that overflowed, the final partial line) takes the slow
// path under a wave-uniform test.
struct LineRowState {
    uint32_t *pos, *keep, *gl, *nextc, *nret, *alloc_next;
};

__device__ __forceinline__ void line_row_advance(const LineRowState &st, const ChunkPool *p, uint32_t per_block, uint32_t row,
                                                 uint32_t base, uint32_t advance, uint32_t new_n, uint32_t new_lo)
{
    // the row's slot 0 moves `advance` keys (a multiple of 64) forward; crossing the end of the chunk
    // retires it and continues in the next one
    const uint32_t room = kChunkKeys - (base & (kChunkKeys - 1));
    uint32_t nb = base + advance;
    if (advance >= room) {
        uint32_t nr = st.nret[row];
        chunk_retire(p, base >> kChunkShift, row, kChunkKeys, nr);
        st.nret[row] = nr;
        nb = (st.nextc[row] << kChunkShift) + (advance - room);
        st.nextc[row] = chunk_alloc(p, per_block, st.alloc_next, 1);
    }
    st.pos[row] = new_n;
    st.keep[row] = new_n;
    st.gl[row] = nb | new_lo;
}

__device__ __forceinline__ void flush_row_slow_chunked(unsigned char *rb, uint32_t rot, uint32_t piece, uint32_t row, uint32_t n,
                                                       uint32_t glw, const LineRowState &st, const ChunkPool *p,
                                                       uint32_t per_block, uint16_t *__restrict__ keys, bool final)
{
    const uint32_t l0 = glw & 63u, base = glw & ~63u, next = st.nextc[row];
    const uint32_t L = n >> 6, r = n & 63u;
    const uint4 line0 = *(const uint4 *)(rb + ((16u * piece + rot) & 255u));
    const uint4 line1 = *(const uint4 *)(rb + ((16u * (piece + 8) + rot) & 255u));
    if (L >= 1) {
        if (l0 == 0) {   // whole line 0 (it lies inside the current chunk)
            *(uint4 *)(keys + base + 8 * piece) = line0;
        } else {         // slots below lo left directly when the row overflowed
            const uint16_t *k = (const uint16_t *)&line0;
#pragma unroll
            for (uint32_t e = 0; e < 8; ++e)
                if (8 * piece + e >= l0) keys[base + 8 * piece + e] = k[e];
        }
        if (L >= 2) *(uint4 *)(keys + line_slot_index(base, next, 64 + 8 * piece)) = line1;
        if (L == 1 && 8 * piece < r) *(uint4 *)(rb + ((16u * piece + rot) & 255u)) = line1;   // leftover moves down
    }
    uint32_t new_n = n, new_lo = l0, adv = 0;
    if (L >= 1) {
        new_n = r;
        new_lo = L >= 2 ? r : 0u;      // with L >= 2 the remainder went out directly
        adv = 64u * L;
    }
    if (final && new_n > new_lo && L <= 1) {
        // unfinished last line of this workgroup's share: slots [new_lo, new_n)
        const uint4 cur = (L == 1) ? line1 : line0;
        const uint16_t *k = (const uint16_t *)&cur;
#pragma unroll
        for (uint32_t e = 0; e < 8; ++e) {
            const uint32_t sl = 8 * piece + e;
            if (sl >= new_lo && sl < new_n) keys[line_slot_index(base, next, adv + sl)] = k[e];
        }
    }
    if (piece == 0) {
        line_row_advance(st, p, per_block, row, base, adv, new_n, new_lo);
        if (final) {   // the partly filled chunk the row ends in
            const uint32_t g = (st.gl[row] & ~63u) + new_n;
            if (g & (kChunkKeys - 1)) {
                uint32_t nr = st.nret[row];
                chunk_retire(p, g >> kChunkShift, row, g & (kChunkKeys - 1), nr);
                st.nret[row] = nr;
            }
        }
    }
}

__device__ __forceinline__ void flush_lines_chunked(unsigned char *rows, const LineRowState &st, const ChunkPool *p,
                                                    uint32_t per_block, uint16_t *__restrict__ keys, bool final)
{
    const int wave = threadIdx.x >> 6;
    uint32_t lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));   // row addresses are recomputed per tile, not hoisted and spilled (see chunk_store_rows)
    const uint32_t piece = lane & 7;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint32_t row = wave * 32 + g * 8 + (lane >> 3);
        const uint32_t n = st.pos[row];
        const uint32_t glw = st.gl[row];
        const uint32_t rot = (row & 15u) << 4;
        unsigned char *rb = rows + row * kLineRowBytes;
        const bool fast = (n >> 6) == 1 && (glw & 63u) == 0;
        const bool slow = !fast && ((n >> 6) >= 1 || final);
        if (__builtin_expect(__any(slow), 0)) {          // wave-uniform
            if (slow || fast) flush_row_slow_chunked(rb, rot, piece, row, n, glw, st, p, per_block, keys, final);
            else if (piece == 0) st.keep[row] = n;
            continue;
        }
        const uint32_t r = n & 63u;
        if (fast) {   // only the lanes of flushing rows touch LDS (roughly half of them per tile)
            const uint4 line0 = *(const uint4 *)(rb + ((16u * piece + rot) & 255u));
            *(uint4 *)(keys + glw + 8 * piece) = line0;                         // one aligned 128-byte line per 8 lanes
            if (8 * piece < r) {                                                // leftover moves down
                const uint4 line1 = *(const uint4 *)(rb + ((16u * (piece + 8) + rot) & 255u));
                *(uint4 *)(rb + ((16u * piece + rot) & 255u)) = line1;
            }
            if (piece == 0) line_row_advance(st, p, per_block, row, glw, 64u, r, 0u);
        } else if (piece == 0) {
            st.keep[row] = n;
        }
    }
}

__global__ __launch_bounds__(kLineThreads) void chunk_key_lines_kernel(const uint32_t *__restrict__ res,
                                                                      const uint64_t *__restrict__ start1,
                                                                      uint32_t keys_per_block, const ChunkPool *__restrict__ p,
                                                                      uint16_t *__restrict__ keys_base, uint32_t per_block,
                                                                      unsigned long long *__restrict__ table)
{
    __shared__ __attribute__((aligned(16))) unsigned char rows[kLineRowsBytes + 16];
    __shared__ uint32_t pos[kNumBuckets], keep[kNumBuckets], gl[kNumBuckets], nextc[kNumBuckets], nret[kNumBuckets];
    __shared__ uint32_t tile_over, alloc_next, defer_n;
    __shared__ DeferRun defer_t[kChunkDeferCap];
    const uint32_t c = blockIdx.y;
    const uint64_t lo = start1[c], n = start1[c + 1] - lo;
    uint16_t *keys = keys_base + (((uint64_t)c * gridDim.x * per_block) << kChunkShift);
    const uint32_t first_chunk = blockIdx.x * per_block;
    if (threadIdx.x == 0) {
        alloc_next = first_chunk + 2 * kNumBuckets;
        tile_over = 0;
        defer_n = 0;
    }
    if (threadIdx.x < kNumBuckets) {
        gl[threadIdx.x] = (first_chunk + threadIdx.x) << kChunkShift;
        nextc[threadIdx.x] = first_chunk + kNumBuckets + threadIdx.x;
        pos[threadIdx.x] = 0;
        keep[threadIdx.x] = 0;
        nret[threadIdx.x] = 0;
    }
    __syncthreads();
    const LineRowState st = {pos, keep, gl, nextc, nret, &alloc_next};
    const uint64_t b0 = (uint64_t)blockIdx.x * keys_per_block;
    const uint64_t per_wave = keys_per_block / kLineWaves;
    const uint64_t w0 = b0 + (uint64_t)(threadIdx.x >> 6) * per_wave;
    for (uint64_t t = 0; t < per_wave; t += kMacroKeys) {
        if (b0 + t >= n) break;  // block-uniform: wave 0 owns the lowest residuals (also skips empty workgroups)
        // (fetching the next tile's residuals one tile ahead, under the placement or under the flush,
        // fits in registers but measured 2-6 % slower: the kernel moves 3.5 TB/s already)
        uint32_t v[16], valid;
        load_macro(res, lo, n, w0 + t, v, valid);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = chunk_scramble<kResKeyBits>(v[j]);
        if (place16_lines_chunked<kResKeyBits>(rows, pos, gl, nextc, keys, v, valid) >= kChunkKeys) tile_over = 1;
        lds_barrier();
        if (tile_over) {   // block-uniform, pathological input only: forget the tile, count it after the loop
            if (threadIdx.x < kNumBuckets) pos[threadIdx.x] = keep[threadIdx.x];
            lds_barrier();
            if (threadIdx.x == 0) {
                tile_over = 0;
                defer_tile(defer_t, defer_n, (uint32_t)(t / kMacroKeys), p->error);
            }
        } else {
            flush_lines_chunked(rows, st, p, per_block, keys, false);
        }
        lds_barrier();
    }
    flush_lines_chunked(rows, st, p, per_block, keys, true);
    __syncthreads();
    if (threadIdx.x < kNumBuckets)
        for (uint32_t e = nret[threadIdx.x]; e < kChunkRow; ++e) chunk_table_row(p, threadIdx.x)[e] = kChunkEmpty;
    uint32_t pend_hot = 0;
    unsigned long long pend_cnt = 0;
    unsigned long long *table_c = table + ((uint64_t)c << kResidualBits);
    for (uint32_t i = 0; i < defer_n; ++i)
        for (uint32_t q = 0; q < defer_t[i].count; ++q)
            chunk_count_keys_direct<kResKeyBits>(res, lo, n, w0 + (uint64_t)(defer_t[i].first + q) * kMacroKeys, table_c, 1,
                                                 pend_hot, pend_cnt);
    if (pend_cnt && (threadIdx.x & 63) == 0) atomicAdd(&table_c[pend_hot], pend_cnt);
}

// C4a: exclusive scan of the overflow counts -> ostart[0..512], cursors reset; slice plan of the
// histogram launch as in part_bucketscan_kernel, in units of chunks.
__global__ __launch_bounds__(kNumBuckets) void chunk_plan_kernel(const uint32_t *__restrict__ nlist,
                                                                 const uint32_t *__restrict__ ovf_n,
                                                                 uint32_t *__restrict__ ostart, uint32_t *__restrict__ ocur,
                                                                 uint32_t *__restrict__ slice_start)
{
    __shared__ uint32_t wsum[kNumBuckets / 64], ssum[kNumBuckets / 64], tsum[kNumBuckets / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    nlist += blockIdx.x * kNumBuckets;           // one workgroup per coarse bucket
    ovf_n += blockIdx.x * kNumBuckets;
    ostart += blockIdx.x * (kNumBuckets + 1);
    ocur += blockIdx.x * kNumBuckets;
    slice_start += blockIdx.x * (kNumBuckets + 1);
    const uint32_t v = ovf_n[threadIdx.x];
    const uint32_t c = nlist[threadIdx.x];
    uint32_t incl = v, cincl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d), oc = __shfl_up(cincl, d);
        if (lane >= d) {
            incl += o;
            cincl += oc;
        }
    }
    if (lane == 63) {
        wsum[wave] = incl;
        tsum[wave] = cincl;
    }
    __syncthreads();
    uint32_t off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kNumBuckets / 64; ++w) {
        off += (w < wave) ? wsum[w] : 0u;
        total += tsum[w];
    }
    ostart[threadIdx.x] = off + incl - v;
    if (threadIdx.x == kNumBuckets - 1) ostart[kNumBuckets] = off + incl;
    ocur[threadIdx.x] = 0;
    const uint32_t target = max(64u, 2u * ((total + kNumBuckets - 1) / kNumBuckets));
    const uint32_t slices = max(1u, (c + target - 1) / target);
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
    slice_start[threadIdx.x] = soff + sincl - slices;
    if (threadIdx.x == kNumBuckets - 1) slice_start[kNumBuckets] = soff + sincl;
}

// C4b: overflow entries grouped by bucket (order inside a bucket is irrelevant).
__global__ __launch_bounds__(256) void chunk_list_kernel(ChunkPool p, const uint32_t *__restrict__ ostart,
                                                         uint32_t *__restrict__ ocur, uint32_t *__restrict__ osorted)
{
    const uint32_t n = p.ovf_count[blockIdx.y];
    const uint64_t region = (uint64_t)blockIdx.y * chunk_pool_chunks(p);
    ostart += blockIdx.y * (kNumBuckets + 1);
    ocur += blockIdx.y * kNumBuckets;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint2 e = p.ovf[region + i];
        osorted[region + ostart[e.x] + atomicAdd(&ocur[e.x], 1u)] = e.y;
    }
}

// C5: LDS histogram of one slice of one bucket's chunks, merged into the table (see
// part_hist_kernel for the slice plan and the hot-key counting).  The bucket's entries are its
// table row (G * 4 words, mostly two or three chunks per workgroup) followed by its overflow
// entries; a slice takes an equal share of both.  The entries are dealt out to the 16 waves round
// robin; a wave works through its non-empty ones chunk by chunk, four 16-byte loads per lane in flight.
template <int KB>
__global__ __launch_bounds__(1024) void chunk_hist_kernel(ChunkPool p, const uint32_t *__restrict__ ostart,
                                                          const uint32_t *__restrict__ osorted,
                                                          const uint32_t *__restrict__ slice_start,
                                                          unsigned long long *__restrict__ table)
{
    constexpr int BINS = 1 << KB;
    __shared__ __attribute__((aligned(16))) uint32_t hist[BINS + 64];
    const uint32_t G = p.groups;
    const uint64_t region = (uint64_t)blockIdx.y * chunk_pool_chunks(p);   // this coarse bucket's part of pool and overflow list
    const uint16_t *keys = p.keys + (region << kChunkShift);
    ostart += blockIdx.y * (kNumBuckets + 1);
    slice_start += blockIdx.y * (kNumBuckets + 1);
    if (blockIdx.x >= slice_start[kNumBuckets]) return;
    uint32_t b = 0;
#pragma unroll
    for (int step = kNumBuckets / 2; step >= 1; step >>= 1)
        if (slice_start[b + step] <= blockIdx.x) b += step;
    const uint32_t sl = blockIdx.x - slice_start[b];
    const uint32_t slices = slice_start[b + 1] - slice_start[b];
    for (int i = threadIdx.x; i < BINS; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // FULL: all 8 keys exist; otherwise only the first `live` (tail vector of a partly filled chunk)
    auto add8 = [&](const uint4 q, uint32_t live, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const uint32_t k[8] = {q.x & 0xFFFFu, q.x >> 16, q.y & 0xFFFFu, q.y >> 16,
                               q.z & 0xFFFFu, q.z >> 16, q.w & 0xFFFFu, q.w >> 16};
        const unsigned long long active = __builtin_amdgcn_ballot_w64(true);
        uint32_t hot = __builtin_amdgcn_readfirstlane(k[0]);
        const unsigned long long agree = __builtin_amdgcn_ballot_w64(k[0] == hot);
        if (__popcll(agree) < 32 && (active & ~agree)) {
            const uint32_t other = __builtin_amdgcn_readlane(k[0], __ffsll((long long)(active & ~agree)) - 1);
            if (__popcll(__builtin_amdgcn_ballot_w64(k[0] == other)) > __popcll(agree)) hot = other;
        }
        uint32_t same = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool exists = FULL || (uint32_t)j < live;
            const bool eq = exists && k[j] == hot;
            same += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(eq));
            atomicAdd(&hist[(eq || !exists) ? (uint32_t)(BINS + lane) : (k[j] & (uint32_t)(BINS - 1))], 1u);
        }
        if (lane == __ffsll((long long)active) - 1) atomicAdd(&hist[hot & (uint32_t)(BINS - 1)], same);
    };
    // A unit is half a chunk: vectors [256h, 256h + 256) (a vector = 8 keys), four per lane.  The
    // loads of the next unit are issued before the current one is histogrammed.
    uint4 qa[4];
    uint32_t la[4];
    bool have = false, all_full = false;   // all_full (wave-uniform): every key of the pending unit exists
    auto consume = [&]() {
        if (all_full) {
#pragma unroll
            for (int u = 0; u < 4; ++u) add8(qa[u], 8u, std::true_type{});
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (la[u]) add8(qa[u], la[u], std::false_type{});
        }
    };
    auto feed = [&](uint32_t e, uint32_t half) {   // wave-uniform arguments
        const uint32_t fill = (e >> kChunkIdBits) + 1u;
        const uint32_t nvec = (fill + 7u) >> 3;
        const uint4 *kv = reinterpret_cast<const uint4 *>(keys + ((uint64_t)(e & ((1u << kChunkIdBits) - 1u)) << kChunkShift));
        uint4 qb[4];
        uint32_t lb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t v = 256u * half + 64u * u + lane;
            lb[u] = v < nvec ? min(8u, fill - 8u * v) : 0u;
            qb[u] = make_uint4(0, 0, 0, 0);
            if (lb[u]) qb[u] = kv[v];
        }
        if (have) consume();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            qa[u] = qb[u];
            la[u] = lb[u];
        }
        have = true;
        all_full = fill >= 8u * 256u * (half + 1u);
    };
    auto walk = [&](const uint32_t *list, uint32_t n_entries) {
        const uint32_t per = (n_entries + slices - 1) / slices;
        const uint32_t e0 = min(sl * per, n_entries), e1 = min((sl + 1) * per, n_entries);
        // entries are dealt out in quads (one table row = the chunks of one scatter workgroup): quad q
        // belongs to wave q % 16, so short lists still keep all waves busy and every wave gets the
        // same mix of full and partly filled chunks.  Lane l looks at entry l % 4 of the wave's
        // (16 * round + l / 4)-th quad.
        for (uint32_t base = e0 + 4u * wave; base < e1; base += 16 * 64) {
            const uint32_t idx = base + 64u * (lane >> 2) + (lane & 3u);
            const uint32_t e = idx < e1 ? list[idx] : kChunkEmpty;
            unsigned long long m = __builtin_amdgcn_ballot_w64(e != kChunkEmpty);
            while (m) {   // wave-uniform
                const int i = __ffsll((long long)m) - 1;
                m &= m - 1;
                const uint32_t ei = __builtin_amdgcn_readlane(e, i);
                feed(ei, 0);
                if ((ei >> kChunkIdBits) + 1u > 256u * 8u) feed(ei, 1);
            }
        }
    };
    walk(p.table + ((uint64_t)blockIdx.y * kNumBuckets + b) * G * kChunkRow, G * kChunkRow);
    walk(osorted + region + ostart[b], ostart[b + 1] - ostart[b]);
    if (have) consume();
    __syncthreads();
    // bin i of scrambled bucket b is table entry (b ^ g(i's top six bits)) << KB | i: runs of 2^(KB-6) bins
    unsigned long long *dst = table + ((uint64_t)blockIdx.y << (kPartBits + KB));
    if (slices == 1) {
        for (int i = threadIdx.x; i < BINS; i += blockDim.x) {
            const uint32_t c = hist[i];
            if (c) dst[((uint64_t)(b ^ chunk_bucket_mask((uint32_t)i >> (KB - 6))) << KB) + i] += (unsigned long long)c;
        }
    } else {
        for (int i = threadIdx.x; i < BINS; i += blockDim.x) {
            const uint32_t c = hist[i];
            if (c) atomicAdd(&dst[((uint64_t)(b ^ chunk_bucket_mask((uint32_t)i >> (KB - 6))) << KB) + i], (unsigned long long)c);
        }
    }
}

}  // namespace kpal
