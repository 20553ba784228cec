// hex_kernels.hpp -- the HEX pipeline, k = 12 (gfx950): radix partition of groups of SIX overlapping k-mers into aligned
// 64-byte records of three-byte items, histogrammed per bucket in six forms of 16-bit LDS bins.  hex_index.hpp holds the
// item format and all index arithmetic (checked on the CPU by tests/native/hex_index_check.cpp); the scatter machinery --
// rows of 20 three-byte slots (sixteen in the low bytes of a row's dwords, four riding in their top bytes), the spill list
// carried from round to round, the per-workgroup hot-item table, records flushed through registers and stored under the
// next tile's placement, wave priorities -- is that of quad_kernels.hpp, which this file reuses where it can.
//
//   H0  hex_sample_kernel     row loads of a 1/64 sample (the host picks the tile: 4, 3, 2 or 1 wave-steps of 3 KiB per wave)
//   H1  hex_scatter_kernel    ASCII -> pool[row / 2][workgroup][round][row % 2][16 dwords]   (one workgroup per CU)
//   H2  hex_hist_kernel       one workgroup per row: six forms x 8192 bins as u16 pairs (96 KiB of LDS), staged as 8-bit counts
//                             in TABLE order (six planes of 4^12 bytes)
//   H3  hex_finalize_kernel   table[i] += the six staged counts of entry i
// 1 B per base read, ~0.7 B per base written and read once more (quad pipeline: 1.0), one LDS slot allocation + one LDS write
// per SIX k-mers.  A lane owns 48 bytes per step (three 16-byte loads): the eight groups ending at its bytes 5, 11, .. 47.
//
// Integer adds commute; every k-mer is in exactly one item with its mask bit set (hex_index_check): bit-exact.
#pragma once
#include "hex_index.hpp"
#include "quad_kernels.hpp"

namespace kpal {

using HX = HexIndex;
constexpr int kHexRowBytes = 64;                 // 16 dwords: 16 items in their low three bytes + 4 riders in the top bytes
constexpr int kHexRowItems = 20;
constexpr int kHexWaves = 16;
constexpr uint32_t kHexStepChunks = 192;         // 16-byte chunks per wave-step: 64 lanes x 3
typedef uint8_t hex_stage_t;
constexpr uint32_t kHexStageLimit = 256u;        // staged counts are 8 bits: larger ones go to the table directly

// ---- the k-mers of an item, counted on the spot (hot items / what no list can hold / what is still carried at the end)
__device__ __forceinline__ void hex_item_to_table(const TableOnly &table, uint32_t row, uint32_t item, unsigned long long n)
{
    uint32_t p23, m6;
    HX::unpack(item, p23, m6);
#pragma unroll
    for (int i = 0; i < HX::kForms; ++i)
        if ((m6 >> (5 - i)) & 1u) sink_add(table, HX::kmer_of(row, p23, i), n);
}

__device__ __forceinline__ uint32_t hex_hot_hash(uint32_t row, uint32_t item) { return (item * 0x9E3779B1u + row * 0x85EBCA6Bu) >> 24; }

// (see quad_items_direct_body: the same hot-item table, keyed by (row, item))
__device__ __forceinline__ void hex_items_direct_body(bool active, uint32_t row, uint32_t item, const TableOnly table, QuadHot *hot)
{
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __builtin_amdgcn_ballot_w64(active);
    for (int round = 0; round < 8 && todo; ++round) {   // wave-uniform
        const int src = __ffsll((long long)todo) - 1;
        const uint32_t hot_row = (uint32_t)__builtin_amdgcn_readlane(row, src);
        const uint32_t hot_item = (uint32_t)__builtin_amdgcn_readlane(item, src);
        const unsigned long long same = __builtin_amdgcn_ballot_w64(active && row == hot_row && item == hot_item) & todo;
        const uint32_t n = (uint32_t)__popcll(same);
        const unsigned long long key = ((unsigned long long)hot_row << 32) | hot_item;
        const uint32_t slot = (hex_hot_hash(hot_row, hot_item) + (uint32_t)(lane & 3)) & (uint32_t)(kQuadHotEntries - 1);
        const unsigned long long seen = lane < 4 ? hot[slot].key : ~0ull;
        const unsigned long long hit = __builtin_amdgcn_ballot_w64(seen == key);
        if (hit) {
            if (lane == __ffsll((long long)hit) - 1) atomicAdd(&hot[slot].count, n);
        } else {
            const unsigned long long free_slots = __builtin_amdgcn_ballot_w64(seen == 0ull);
            bool placed = false;
            if (n >= 2u && free_slots) {
                const int who = __ffsll((long long)free_slots) - 1;
                const unsigned long long old = lane == who ? atomicCAS(&hot[slot].key, 0ull, key) : 1ull;
                placed = __builtin_amdgcn_ballot_w64(lane == who && (old == 0ull || old == key)) != 0ull;
                if (placed && lane == who) atomicAdd(&hot[slot].count, n);
            }
            if (!placed && lane == src) hex_item_to_table(table, hot_row, hot_item, n);
        }
        todo &= ~same;
    }
    if ((todo >> lane) & 1ull) {   // many different items in one wave: every lane for itself
        const unsigned long long key = ((unsigned long long)row << 32) | item;
        const uint32_t h = hex_hot_hash(row, item);
        bool done = false;
#pragma unroll
        for (int pr = 0; pr < 4 && !done; ++pr) {
            const uint32_t slot = (h + (uint32_t)pr) & (uint32_t)(kQuadHotEntries - 1);
            if (hot[slot].key == key) {
                atomicAdd(&hot[slot].count, 1u);
                done = true;
            }
        }
        if (!done) hex_item_to_table(table, row, item, 1ULL);
    }
}

__device__ __attribute__((noinline)) void hex_items_direct(bool active, uint32_t row, uint32_t item, const TableOnly table, QuadHot *hot)
{
#if !defined(KPAL_AB_HEX_NO_HOT)      // A/B timing / resource builds (wrong counts)
    hex_items_direct_body(active, row, item, table, hot);
#endif
}

// Items that did not get a slot (bit q of `over`) into the spill list -- one LDS atomic per call reserves the entries of all the
// wave's items, a lane finds its own with ballots and lane counts (quad_place) -- or, DIRECT / list full, counted on the spot.
template <bool DIRECT, int N>
__device__ __forceinline__ void hex_spill(uint32_t over, QuadSpill *spill, uint32_t *spill_n, uint32_t cap, const uint32_t (&row)[N],
                                          const uint32_t (&item)[N], const TableOnly &table, QuadHot *hot)
{
    uint32_t base = 0;
    bool list_full = true;
    unsigned long long bq[N];
#pragma unroll
    for (int q = 0; q < N; ++q) bq[q] = __builtin_amdgcn_ballot_w64((over >> q) & 1u);
    if constexpr (!DIRECT) {
        uint32_t total = 0;
#pragma unroll
        for (int q = 0; q < N; ++q) total += (uint32_t)__popcll(bq[q]);
        uint32_t got = 0;
        if ((threadIdx.x & 63u) == 0u) got = atomicAdd(spill_n, total);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
        list_full = base + total > cap;
    }
#pragma unroll
    for (int q = 0; q < N; ++q) {
        if (bq[q] == 0ull) continue;             // wave-uniform
        const bool ov = (over >> q) & 1u;
        bool listed = false;
        if constexpr (!DIRECT) {
            const unsigned long long b = bq[q];
            const uint32_t at = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
            listed = ov && at < cap;
            if (listed) spill[at] = QuadSpill{row[q], item[q]};
            base += (uint32_t)__popcll(b);
        }
        if (list_full && __any(ov && !listed)) hex_items_direct(ov && !listed, row[q], item[q], table, hot);
    }
}

// Returns the mask (bit q) of this lane's items that did not fit their row.  pos[row] counts the BYTES in use of the row's
// sixteen dword slots (0, 4, .. 60), then its four rider slots (64 .. 76): see quad_place, ITEM3.  An item is placed iff it is not 0.
template <bool DIRECT, int N>
__device__ __forceinline__ uint32_t hex_place(uint32_t *rows, uint32_t *pos, QuadSpill *spill, uint32_t *spill_n, uint32_t cap,
                                              const uint32_t (&row)[N], const uint32_t (&item)[N], const TableOnly &table, QuadHot *hot)
{
    constexpr uint32_t RB = kHexRowBytes;
    uint32_t riders = 0, over = 0;
    uint32_t off[N];
#pragma unroll
    for (int q = 0; q < N; ++q) off[q] = atomicAdd(&pos[row[q]], item[q] ? 4u : 0u);
#pragma unroll
    for (int q = 0; q < N; ++q) {
        const bool counted = item[q] != 0u;
        const bool normal = off[q] < RB;
        const bool rider = !normal && off[q] < RB + RB / 4;
        if (counted && normal) atomicOr(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(rows) + row[q] * RB + off[q]), item[q]);
        riders |= (counted && rider) ? (1u << q) : 0u;
        over |= (counted && !normal && !rider) ? (1u << q) : 0u;
    }
    if (__any(riders != 0u)) {   // wave-uniform
#pragma unroll
        for (int q = 0; q < N; ++q)
            if ((riders >> q) & 1u) {
                unsigned char *p = reinterpret_cast<unsigned char *>(rows) + row[q] * RB + (off[q] - RB) * 4u;
                p[3] = (unsigned char)item[q];
                p[7] = (unsigned char)(item[q] >> 8);
                p[11] = (unsigned char)(item[q] >> 16);
            }
    }
    if (__builtin_expect(__any(over != 0u), 0)) hex_spill<DIRECT, N>(over, spill, spill_n, cap, row, item, table, hot);
    return over;
}

template <int CARRY>
__device__ __forceinline__ void hex_place_carried(uint32_t *rows, uint32_t *pos, QuadSpill *spill, uint32_t *spill_n, uint32_t cap,
                                                  const uint32_t (&carry_row)[CARRY], uint32_t (&carry_item)[CARRY], const TableOnly &table, QuadHot *hot)
{
#pragma unroll
    for (int c = 0; c < CARRY; ++c) {
        const uint32_t r1[1] = {carry_row[c]};
        const uint32_t i1[1] = {carry_item[c]};
        if (__any(carry_item[c] != 0u)) {
            // an item that does not fit even now has been counted: it is no longer carried
            if (hex_place<true, 1>(rows, pos, spill, spill_n, cap, r1, i1, table, hot) & 1u) carry_item[c] = 0;
        }
    }
}

// ---- a lane's three chunks of one wave-step -> codes, flags, the left neighbour's, the 48-bit emit mask
struct HexLane {
    uint32_t pc, c0, c1, c2;
    uint64_t emit;
};

// raw: the lane's 48 bytes (chunks 192 u + 3 lane + 0..2 of the stream).  carry: lane 63's last chunk of the step before.
__device__ __forceinline__ HexLane hex_encode(const Span &s, uint64_t u, const uint4 (&raw)[3], Chunk &carry, bool edge)
{
    const int lane = threadIdx.x & 63;
    Chunk a = encode16(raw[0]), b = encode16(raw[1]), c = encode16(raw[2]);
    const int64_t c0 = (int64_t)(u * kHexStepChunks + (uint64_t)lane * 3u);
    if (edge) {   // wave-uniform
        range_fix(s, c0, a);
        range_fix(s, c0 + 1, b);
        range_fix(s, c0 + 2, c);
    }
    HexLane r;
    r.pc = from_left_lane(c.codes, carry.codes);
    const uint32_t pb = from_left_lane(c.bad, carry.bad);
    carry.codes = __builtin_amdgcn_readlane(c.codes, 63);
    carry.bad = __builtin_amdgcn_readlane(c.bad, 63);
    r.c0 = a.codes;
    r.c1 = b.codes;
    r.c2 = c.codes;
    r.emit = HX::emit48(pb, a.bad, b.bad, c.bad);
    if (edge) {
        const uint64_t from = ((uint64_t)emit_from_mask(s, c0) << 32) | ((uint64_t)emit_from_mask(s, c0 + 1) << 16) | emit_from_mask(s, c0 + 2);
        r.emit &= from;
    }
    return r;
}

// group q of the lane -> row and its item(s): a = first item (0: nothing counts), b = second (see HexIndex::split), branch-free
template <int Q>
__device__ __forceinline__ void hex_group(const HexLane &l, uint32_t &row, uint32_t &a, uint32_t &b)
{
    const uint64_t x = HX::group_x(l.pc, l.c0, l.c1, l.c2, Q);
    const uint32_t m6 = HX::group_mask(l.emit, Q);
    const uint32_t up = m6 >> 3, lo = m6 & 7u;
    const uint32_t hl = lo ? HX::half_item(x, 1, lo) : 0u;
    row = HX::row_of(x);
    a = m6 == 63u ? HX::full_item(x) : (up ? HX::half_item(x, 0, up) : hl);
    b = (m6 != 63u && up) ? hl : 0u;
}

// interior: the chunks of wave-steps [u0, u1) and the chunk left of them lie inside the fed range and right of emit_from
__device__ __forceinline__ bool hex_interior(const Span &s, uint64_t u0, uint64_t u1)
{
    return interior_range(s, u0 * kHexStepChunks, u1 * kHexStepChunks);
}

__device__ __forceinline__ void hex_fetch(const Span &s, uint64_t u, uint32_t lane48, bool inside, uint4 (&out)[3])
{
    if (inside) {   // wave-uniform: scalar base + the lane's constant byte offset
        const char *p = reinterpret_cast<const char *>(s.base + u * kHexStepChunks) + lane48;
        out[0] = *reinterpret_cast<const uint4 *>(p);
        out[1] = *reinterpret_cast<const uint4 *>(p + 16);
        out[2] = *reinterpret_cast<const uint4 *>(p + 32);
    } else {
        const int64_t c = (int64_t)(u * kHexStepChunks + lane48 / 16u);
        out[0] = fetch_chunk(s, c);
        out[1] = fetch_chunk(s, c + 1);
        out[2] = fetch_chunk(s, c + 2);
    }
}

// H0: row loads of a sample (see quad_sample_kernel).  Workgroup g encodes `steps` wave-steps per wave starting at its share.
__global__ __launch_bounds__(512) void hex_sample_kernel(Span s, uint64_t stride_steps, uint32_t steps, uint32_t *__restrict__ load)
{
    __shared__ uint32_t cnt[HX::kRows];
    for (int i = threadIdx.x; i < HX::kRows; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t total = (s.nchunks + kHexStepChunks - 1) / kHexStepChunks;
    const uint64_t first = (uint64_t)blockIdx.x * stride_steps + (uint64_t)wave * steps;
    if (first < total) {
        Chunk carry = load_chunk(s, (int64_t)(first * kHexStepChunks) - 1);
        for (uint32_t st = 0; st < steps && first + st < total; ++st) {
            uint4 raw[3];
            hex_fetch(s, first + st, (uint32_t)lane * 48u, false, raw);
            const HexLane l = hex_encode(s, first + st, raw, carry, true);
            uint32_t row[8], a[8], b[8];
            hex_group<0>(l, row[0], a[0], b[0]);
            hex_group<1>(l, row[1], a[1], b[1]);
            hex_group<2>(l, row[2], a[2], b[2]);
            hex_group<3>(l, row[3], a[3], b[3]);
            hex_group<4>(l, row[4], a[4], b[4]);
            hex_group<5>(l, row[5], a[5], b[5]);
            hex_group<6>(l, row[6], a[6], b[6]);
            hex_group<7>(l, row[7], a[7], b[7]);
#pragma unroll
            for (int q = 0; q < 8; ++q) atomicAdd(&cnt[row[q]], (a[q] ? 1u : 0u) + (b[q] ? 1u : 0u));
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HX::kRows; i += blockDim.x)
        if (cnt[i]) atomicAdd(&load[i], cnt[i]);
}

// H1: ASCII -> records (see quad_scatter_kernel for the structure: place carried items, then the tile's; barrier; every row
// read into registers as one null-padded record and zeroed behind the read; barrier; the records leave during the next tile's
// placement).  Tile j of workgroup g is tile j * G + g of the input; wave w takes wave-steps STEPS * w .. STEPS * w + STEPS - 1 of it.
// A group's SECOND item (a read end inside the group with counting k-mers on both sides of the half boundary: 2.6 % of the
// groups of 150-base reads) does not get a placement of its own: it goes into the spill list and is placed with the carried
// items of the next round.
template <int STEPS, int DEPTH>
__global__ __launch_bounds__(kHexWaves * 64) void hex_scatter_kernel(Span s, uint64_t tiles_per_block, uint32_t *__restrict__ pool, uint32_t rounds_cap,
                                                                     uint32_t *__restrict__ nrounds, uint32_t *__restrict__ error, TableOnly table)
{
    constexpr int NB = HX::kRows, S = kHexRowBytes / 4;
    constexpr int THREADS = kHexWaves * 64;
    constexpr int CARRY = kQuadSpillCap / THREADS;
    constexpr uint32_t CAP = CARRY * THREADS;
    static_assert(STEPS % DEPTH == 0 && NB * S == kQuadRowWords, "tile shape");
    __shared__ __attribute__((aligned(16))) uint32_t rows[kQuadRowWords + kQuadDummyWords];
    __shared__ __attribute__((aligned(16))) uint32_t pos[NB];
    __shared__ QuadSpill spill[kQuadSpillCap];
    __shared__ uint32_t spill_cnt[2];
    __shared__ QuadHot hot[kQuadHotEntries];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const uint32_t lane48 = (uint32_t)lane * 48u;
    for (int i = threadIdx.x; i < kQuadRowWords + kQuadDummyWords; i += THREADS) rows[i] = 0;
    for (int i = threadIdx.x; i < NB; i += THREADS) pos[i] = 0;
    for (int i = threadIdx.x; i < kQuadHotEntries; i += THREADS) hot[i] = QuadHot{0ull, 0u, 0u};
    if (threadIdx.x == 0) {
        spill_cnt[0] = 0;
        spill_cnt[1] = 0;
    }
    __syncthreads();
    const uint64_t total_steps = (s.nchunks + kHexStepChunks - 1) / kHexStepChunks;
    constexpr int kTileSteps = kHexWaves * STEPS;
    auto tile_step = [&](uint64_t j) -> uint64_t { return ((j * gridDim.x + blockIdx.x) * kHexWaves + (uint64_t)wave) * STEPS; };
    auto tile_exists = [&](uint64_t j) -> bool { return j < tiles_per_block && (j * gridDim.x + blockIdx.x) * (uint64_t)kTileSteps < total_steps; };
    auto inside = [&](uint64_t u0, uint64_t u1) -> bool { return u1 * kHexStepChunks <= s.nchunks; };   // plain loads may be used
    uint4 raw[DEPTH][3];
    uint4 rawh;
    {
        const uint64_t f = tile_step(0);
#pragma unroll
        for (int st = 0; st < DEPTH; ++st) hex_fetch(s, f + st, lane48, inside(f + st, f + st + 1), raw[st]);
        rawh = fetch_chunk(s, (int64_t)(f * kHexStepChunks) - 1);
    }
    uint32_t round = 0;
    uint32_t carry_row[CARRY], carry_item[CARRY];
#pragma unroll
    for (int c = 0; c < CARRY; ++c) carry_row[c] = carry_item[c] = 0;
    constexpr int LPR = S / 4;                           // 4 vectors per record
    constexpr int FI = kQuadRowWords / 4 / THREADS;      // 8 vectors per thread
    uint4 rec[FI];
    bool have_rec = false;
    // rows 2j and 2j+1 share a 128-byte line of the pool (QuadCfg<12>::kPairRows): a flush writes whole lines
    const uint64_t row_bytes = (uint64_t)gridDim.x * rounds_cap * (uint64_t)(S * 4 * 2);
    const uint32_t thread_off = (uint32_t)((uint64_t)(threadIdx.x / (LPR * 2)) * row_bytes) + (threadIdx.x % (LPR * 2)) * 16u;
    auto store_rec = [&](int i) {
        uint32_t o = thread_off;
        asm volatile("" : "+v"(o));
        const uint64_t sc = (uint64_t)(i * (THREADS / (LPR * 2))) * row_bytes + ((uint64_t)blockIdx.x * rounds_cap + (round - 1u)) * (uint64_t)(S * 4 * 2);
        *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(pool) + sc + o) = rec[i];
    };
    constexpr int PER_STEP = FI / STEPS;                 // records a step of the next tile stores
    for (uint64_t j = 0; tile_exists(j); ++j) {          // block-uniform
        const uint64_t first = tile_step(j);
        const bool more = tile_exists(j + 1);
        const uint64_t fnext = tile_step(j + 1);
        const bool edge = !hex_interior(s, first, first + STEPS);
        const bool next_inside = more && inside(fnext, fnext + STEPS);
        uint32_t *spill_n = &spill_cnt[j & 1];
        hex_place_carried<CARRY>(rows, pos, spill, spill_n, CAP, carry_row, carry_item, table, hot);
        Chunk carry = encode16(rawh);
        range_fix(s, (int64_t)(first * kHexStepChunks) - 1, carry);
        if (more) rawh = fetch_chunk(s, (int64_t)(fnext * kHexStepChunks) - 1);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            quad_tile_priority<STEPS>(st);
            const HexLane l = hex_encode(s, first + st, raw[st % DEPTH], carry, edge);
            if (have_rec) {
#pragma unroll
                for (int i = st * PER_STEP; i < (st + 1) * PER_STEP; ++i) store_rec(i);
            }
            {
                uint32_t l48 = lane48;
                asm volatile("" : "+v"(l48));
                if (st + DEPTH < STEPS) hex_fetch(s, first + st + DEPTH, l48, inside(first + st + DEPTH, first + st + DEPTH + 1), raw[st % DEPTH]);
                else if (more) hex_fetch(s, fnext + (st + DEPTH - STEPS), l48, next_inside, raw[st % DEPTH]);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {                // four groups at a time
                uint32_t row[4], a[4], b[4];
                if (h == 0) {
                    hex_group<0>(l, row[0], a[0], b[0]);
                    hex_group<1>(l, row[1], a[1], b[1]);
                    hex_group<2>(l, row[2], a[2], b[2]);
                    hex_group<3>(l, row[3], a[3], b[3]);
                } else {
                    hex_group<4>(l, row[0], a[0], b[0]);
                    hex_group<5>(l, row[1], a[1], b[1]);
                    hex_group<6>(l, row[2], a[2], b[2]);
                    hex_group<7>(l, row[3], a[3], b[3]);
                }
                hex_place<false, 4>(rows, pos, spill, spill_n, CAP, row, a, table, hot);
                const uint32_t second = (b[0] ? 1u : 0u) | (b[1] ? 2u : 0u) | (b[2] ? 4u : 0u) | (b[3] ? 8u : 0u);
#if !defined(KPAL_AB_HEX_NO_SECOND)   // A/B timing / resource builds (wrong counts)
                if (__any(second != 0u)) hex_spill<false, 4>(second, spill, spill_n, CAP, row, b, table, hot);
#endif
            }
        }
        have_rec = false;
        quad_tile_priority<1>(0);
        lds_barrier();
        const uint32_t spilled = min(*spill_n, CAP);
        quad_note_spill(error, *spill_n, CAP);
        if (threadIdx.x == 0) spill_cnt[(j & 1) ^ 1] = 0;
        quad_take_carried<CARRY, THREADS>(spill, spilled, carry_row, carry_item);
        {
            const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
            uint4 *rv = reinterpret_cast<uint4 *>(rows);
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                const uint4 v = rv[threadIdx.x + (uint32_t)i * THREADS];
                rec[i].x = v.x;                      // (component by component: see quad_scatter_kernel -- whole-struct copies kept rec[] in scratch memory)
                rec[i].y = v.y;
                rec[i].z = v.z;
                rec[i].w = v.w;
                rv[threadIdx.x + (uint32_t)i * THREADS] = zero4;
            }
            for (int i = threadIdx.x * 4; i < NB; i += THREADS * 4) *reinterpret_cast<uint4 *>(&pos[i]) = zero4;
        }
        if (round >= rounds_cap) {               // cannot happen: one round per tile, rounds_cap = tiles per workgroup
            if (threadIdx.x == 0) *error = 2u;
        } else {
            have_rec = true;
            ++round;
#pragma unroll
            for (int i = STEPS * PER_STEP; i < FI; ++i) store_rec(i);
        }
        lds_barrier();
    }
    if (have_rec) {
#pragma unroll
        for (int i = 0; i < STEPS * PER_STEP; ++i) store_rec(i);
    }
    if (threadIdx.x == 0) nrounds[blockIdx.x] = min(round, rounds_cap);
    // what is still carried over, then the table of hot items: into the count table
#pragma unroll
    for (int c = 0; c < CARRY; ++c) hex_items_direct_body(carry_item[c] != 0u, carry_row[c], carry_item[c], table, hot);
    __syncthreads();
    uint32_t used = 0;
    for (int i = threadIdx.x; i < kQuadHotEntries; i += THREADS) {
        const QuadHot h = hot[i];
        if (h.key && h.count) {
            ++used;
            hex_item_to_table(table, (uint32_t)(h.key >> 32), (uint32_t)h.key, (unsigned long long)h.count);
        }
    }
    if (used) atomicAdd(error + 1, used);
}

// H2: histogram of one row's records in six forms.  hist[form][8192 bins] as u16 PAIRS: bin `local` of form i lives in the
// (local >> 12) half of word  i * 4096 + (local & 4095)  -- 96 KiB.  A half must never pass 65535: between two barriers every wave
// counts at most 1280 items (four 16-byte vectors of five items per lane), 20480 per workgroup, and the scan behind every
// such batch drains every half that has reached 32768 into the table -- so a half is below 32768 + 20480 at all times.  (u32
// bins would need 192 KiB; two workgroups per row reading its records twice cost more than the u16 bookkeeping.)
template <bool STAGED>
__global__ __launch_bounds__(1024) void hex_hist_kernel(const uint32_t *__restrict__ pool, const uint32_t *__restrict__ nrounds, uint32_t G,
                                                        uint32_t rounds_cap, TableOnly table, hex_stage_t *__restrict__ stage)
{
    constexpr int PLANE = 4096, S = kHexRowBytes / 4;
    __shared__ __attribute__((aligned(16))) uint32_t hist[HX::kForms * PLANE];
    __shared__ uint32_t wave_iters[16];
    // rows 2j and 2j+1 share every 128-byte line of their records: workgroups b and b + 8 (same XCD, dispatched together) take such a pair
    const uint32_t row = ((blockIdx.x >> 4) << 4) | ((blockIdx.x & 7u) << 1) | ((blockIdx.x >> 3) & 1u);
    for (int i = threadIdx.x; i < HX::kForms * PLANE; i += blockDim.x) hist[i] = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // a wave takes the runs g = wave, wave + 16, ... (run g = the records scatter workgroup g wrote for this row); every wave makes
    // the same number of batch iterations (the barriers of the drain scan): the longest wave's
    uint32_t my_iters = 0;
    for (uint32_t g = (uint32_t)wave; g < G; g += 16) my_iters += (nrounds[g] * (uint32_t)(S / 4) + 255u) / 256u;
    if (lane == 0) wave_iters[wave] = my_iters;
    __syncthreads();
    uint32_t iters = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) iters = max(iters, wave_iters[w]);
    auto add_item = [&](uint32_t it) {
        // (the bit-fields read here lie below bit 24: a rider's byte above an item does not matter)
        const bool full = (it >> 23) & 1u;
        const uint32_t m3 = (it >> 19) & 7u, p17 = it & 0x1FFFFu;
        const bool lower = (it >> 22) & 1u;
        const uint32_t up23 = ((p17 >> 7) << 13) | ((p17 & 127u) << 6);
        const uint32_t p23 = full ? (it & 0x7FFFFFu) : (lower ? p17 : up23);
        const uint32_t m6 = full ? 63u : (lower ? m3 : (m3 << 3));
#pragma unroll
        for (int i = 0; i < HX::kForms; ++i) {
            const uint32_t counted = (m6 >> (5 - i)) & 1u;
            const uint32_t word = __builtin_amdgcn_ubfe(p23, 10 - 2 * i, 12);         // local & 4095
            const uint32_t half = __builtin_amdgcn_ubfe(p23, 22 - 2 * i, 1);          // local >> 12
            atomicAdd(&hist[i * PLANE + word], counted << (half * 16u));
        }
    };
    auto add5 = [&](const uint4 q) {
        const uint32_t i4 = __builtin_amdgcn_perm(q.z, __builtin_amdgcn_perm(q.y, q.x, 0x0c0c0703u), 0x0c070100u);   // the top bytes of x, y, z
        if (q.x & 0xFFFFFFu) add_item(q.x);
        if (q.y & 0xFFFFFFu) add_item(q.y);
        if (q.z & 0xFFFFFFu) add_item(q.z);
        if (q.w & 0xFFFFFFu) add_item(q.w);
        if (i4 & 0xFFFFFFu) add_item(i4);
    };
    // drain scan: every half that has reached 32768 goes to the table and starts again at zero
    auto drain = [&]() {
        const uint4 *hv = reinterpret_cast<const uint4 *>(hist);
        uint32_t any = 0;
#pragma unroll
        for (int j = 0; j < HX::kForms * PLANE / 4 / 1024; ++j) {
            const uint4 v = hv[threadIdx.x + 1024u * (uint32_t)j];
            any |= v.x | v.y | v.z | v.w;
        }
        if (__builtin_expect((any & 0x80008000u) != 0u, 0)) {
#pragma unroll 1
            for (int j = 0; j < HX::kForms * PLANE / 4 / 1024; ++j) {
                const uint32_t vec = threadIdx.x + 1024u * (uint32_t)j;
#pragma unroll 1
                for (uint32_t e = 0; e < 4; ++e) {
                    const uint32_t w = vec * 4u + e;
                    uint32_t v = hist[w];
                    if ((v & 0x80008000u) == 0u) continue;
                    const uint32_t form = w / PLANE, word = w % PLANE;
                    if (v & 0x8000u) {
                        sink_add(table, HX::bin_entry(row, (int)form, word), (unsigned long long)(v & 0xFFFFu));
                        v &= 0xFFFF0000u;
                    }
                    if (v & 0x80000000u) {
                        sink_add(table, HX::bin_entry(row, (int)form, word + 4096u), (unsigned long long)(v >> 16));
                        v &= 0x0000FFFFu;
                    }
                    hist[w] = v;
                }
            }
        }
    };
    // the wave's position in its runs
    uint32_t g = (uint32_t)wave;                  // current run
    uint32_t v = 0, nvec = 0;                     // next vector of the run / its length (0: not opened yet)
    const uint4 *src = nullptr;
    bool open = false;
    auto open_run = [&]() {
        while (g < G) {
            nvec = nrounds[g] * (uint32_t)(S / 4);
            if (nvec) break;
            g += 16;
        }
        if (g < G) {
            src = reinterpret_cast<const uint4 *>(pool + ((uint64_t)((row >> 1) * G + g) * rounds_cap) * (2 * S) + (row & 1u) * S);
            v = 0;
            open = true;
        }
    };
    auto fetch = [&](uint32_t at) -> uint4 {
        const uint32_t a = min(at, nvec - 1u);
        uint4 r = src[(a >> 2) * 8u + (a & 3u)];                     // the row's half of every 128-byte line
        const uint32_t keep = at < nvec ? 0xFFFFFFFFu : 0u;
        r.x &= keep;
        r.y &= keep;
        r.z &= keep;
        r.w &= keep;
        return r;
    };
    open_run();
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0;
    if (open) {
        q0 = fetch(v + lane), q1 = fetch(v + 64u + lane), q2 = fetch(v + 128u + lane), q3 = fetch(v + 192u + lane);
    }
    for (uint32_t it = 0; it < iters; ++it) {     // block-uniform
        if (it < my_iters) {                      // wave-uniform
            const uint4 c0 = q0, c1 = q1, c2 = q2, c3 = q3;
            // request the next batch before this one is counted
            v += 256u;
            if (v >= nvec) {
                g += 16;
                open = false;
                if (it + 1 < my_iters) open_run();
            }
            if (open) {
                q0 = fetch(v + lane), q1 = fetch(v + 64u + lane), q2 = fetch(v + 128u + lane), q3 = fetch(v + 192u + lane);
            }
            add5(c0);
            add5(c1);
            add5(c2);
            add5(c3);
        }
        __syncthreads();
        drain();
        __syncthreads();
    }
    // ---- out: 8-bit staged counts in table order (plane i = form i), or straight into the table
    for (int i = 0; i < HX::kForms; ++i) {
        for (uint32_t w4 = threadIdx.x; w4 < (uint32_t)HX::kFormBins / 4u; w4 += blockDim.x) {
            // four consecutive bins of the form (consecutive table entries for forms >= 1; form 0: lopart has three bits, four
            // consecutive locals share hipart and the t's upper bit -- entries e, e+1, e+2, e+3 of one bucket only when t's low two
            // bits do not touch the bucket: they do (smask), so form 0 is written entry by entry)
            uint32_t c[4];
#pragma unroll
            for (uint32_t e = 0; e < 4; ++e) {
                const uint32_t local = w4 * 4u + e;
                c[e] = (hist[i * PLANE + (local & 4095u)] >> ((local >> 12) * 16u)) & 0xFFFFu;
            }
            if constexpr (STAGED) {
#pragma unroll
                for (uint32_t e = 0; e < 4; ++e)
                    if (__builtin_expect(c[e] >= kHexStageLimit, 0)) {
                        sink_add(table, HX::bin_entry(row, i, w4 * 4u + e), (unsigned long long)c[e]);
                        c[e] = 0;
                    }
                hex_stage_t *plane = stage + ((size_t)i << (2 * HX::K));
                if (i == 0) {
#pragma unroll
                    for (uint32_t e = 0; e < 4; ++e) plane[HX::bin_entry(row, 0, w4 * 4u + e)] = (hex_stage_t)c[e];
                } else {
                    *reinterpret_cast<uint32_t *>(plane + HX::bin_entry(row, i, w4 * 4u)) = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
                }
            } else {
#pragma unroll
                for (uint32_t e = 0; e < 4; ++e)
                    if (c[e]) sink_add(table, HX::bin_entry(row, i, w4 * 4u + e), (unsigned long long)c[e]);
            }
        }
    }
}

// H3: table[i] += the six staged counts of entry i (sixteen entries per thread and step).
__global__ __launch_bounds__(256) void hex_finalize_kernel(const hex_stage_t *__restrict__ stage, unsigned long long *__restrict__ table)
{
    constexpr uint64_t BINS = 1ull << (2 * HX::K);
    for (uint64_t e0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16u; e0 < BINS; e0 += (uint64_t)gridDim.x * blockDim.x * 16u) {
        uint4 f[HX::kForms];
#pragma unroll
        for (int i = 0; i < HX::kForms; ++i) f[i] = *reinterpret_cast<const uint4 *>(stage + (size_t)i * BINS + e0);
        uint32_t sum[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) sum[e] = 0;
#pragma unroll
        for (int i = 0; i < HX::kForms; ++i) {
            const uint32_t w[4] = {f[i].x, f[i].y, f[i].z, f[i].w};
#pragma unroll
            for (int e = 0; e < 16; ++e) sum[e] += (w[e >> 2] >> (8 * (e & 3))) & 0xFFu;
        }
        ulonglong2 *t = reinterpret_cast<ulonglong2 *>(table + e0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if ((sum[2 * e] | sum[2 * e + 1]) == 0u) continue;
            ulonglong2 v = t[e];
            v.x += sum[2 * e];
            v.y += sum[2 * e + 1];
            t[e] = v;
        }
    }
}

}  // namespace kpal
