// quad_kernels.hpp -- radix partition of QUADS of overlapping k-mers into aligned records (k = 8..16, gfx950).
//
// What bounded the chunked scatter of chunk_kernels.hpp (profiles/r2/): one range-checked
// buffer_store_short per (bucket, tile) run of ~48 two-byte keys costs ~30 clk of the CU's store path
// (SQ_INSTS_VMEM_WR: one store per 32 clk per CU over the whole kernel, LDS only 39 % busy), and the
// path moves 8-10 B/clk/CU no matter how the bytes are cut -- provided they leave as whole ALIGNED
// pieces of >= 64 B (tools/store_probe2.hip: 64 B 8 clk, 128 B 16 clk, 1 KiB 103 clk; 32 B 12.5 clk;
// 64 lanes x 2 B 30 clk per 128 B).  So this pipeline cuts bytes per k-mer and stores only such pieces:
//
//   * ITEM = four overlapping k-mers.  The (K+3)-mer x ending at byte 4q+3 of a lane's 16-byte chunk
//     holds the k-mers ending at bytes 4q .. 4q+3: k-mer i = x[2K+5-2i : 6-2i].  All four share the
//     bits x[2K-1 : 6]; the top B of those are the BUCKET.  An item keeps the rest: the six bits above
//     the shared field, the L = 2K-B bits below the bucket, and a 4-bit mask of which of the four
//     k-mers count (read ends, N, lower case handled by the mask exactly like emit_mask): one slot
//     allocation + one LDS write per FOUR k-mers, ~1 B per k-mer instead of 2.  4 bytes; at k = 12
//     (23 bits) 3 bytes, five to a 16-byte vector.
//   * RECORD = one bucket's items of one flush round, padded with null items (mask 0) to a fixed size
//     (256 B at k <= 11: 512 buckets x 64 items; 64 B at k = 12: 2048 buckets x 20 items, two buckets to a
//     128-byte line), written with 16-byte stores to its own aligned place  pool[bucket][workgroup][round]
//     -- no cursors, no chunk allocation, no tables: the histogram stage reads pool[bucket] as one stream.
//   * A row that overflows (Poisson tail, 1-3 % of the items) spills into a small LDS list whose
//     entries are placed again at the start of the next round; the host sizes the tile so that the
//     steady-state backlog of that list stays small (kpal_hip.hip: quad_expected_backlog).  Items that do
//     not fit even then, or that the list cannot hold (a row that is over-full every round: homopolymers,
//     satellite repeats), are counted on the spot: ballot-aggregated per wave into a 256-entry (row, item)
//     hash table in LDS that is added to the count table once per workgroup.  No round is ever abandoned,
//     no input is read twice.
//   * HISTOGRAM: one workgroup per bucket, four forms (one per k-mer position) of 2^L bins in LDS
//     (128 KiB at k = 11, 12); the bins of form i are table entries (hi << (B+s)) | (bucket << s) | lo
//     with s = L-6+2i; merged with global atomics (forms of different buckets interleave in the table).
//   * k = 13..16: two levels of the same scatter (coarse, then fine bucket), the forms staged as 16-bit
//     counts and combined per table entry (end of this file).
//
// Integer adds commute, every k-mer is in exactly one item with its mask bit set: bit-exact.
#pragma once
#include <type_traits>

#include "kpal_device.hpp"
#include "quad2_index.hpp"

namespace kpal {

constexpr int kQuadSpillCap = 2048;           // spilled items a round may carry over (16 KiB of LDS)
constexpr int kQuadRowWords = 32768;          // 128 KiB of rows
constexpr int kQuadDummyWords = 4;            // words behind the rows that lanes without a slot write to

template <int K>
struct QuadCfg {
    static_assert(K >= 8 && K <= 16, "quads: k = 8..16");
    // k = 13..16 (two levels, see the end of this file): these are the rows of LEVEL 1 -- the coarse bucket (the top
    // 2K-22 shared bits, scrambled) x a replica, 256 rows of 128 slots (1024 of 32 at k = 16); the 4-byte item keeps
    // the 9-bit fine bucket of level 2 as well:  hi6 << 26 | fine9 << 17 | low13 << 4 | mask4.
    static constexpr bool kTwoLevel = K >= 13;
    // the histogram stage can STAGE its forms for quad2_finalize_kernel instead of adding them to the table with atomics
    // (quad2_index.hpp): the two-level path always does, the one-level path at k = 12 when the host says so
    static constexpr bool kStaged = K >= 12;
    static constexpr int kCoarseBits = kTwoLevel ? 2 * K - 22 : 0;           // 4, 6, 8, 10
    static constexpr int kCoarse = 1 << kCoarseBits;
    static constexpr int kRep = kTwoLevel ? (kCoarse >= 256 ? 1 : 256 / kCoarse) : 1;   // 16, 4, 1, 1
    static constexpr int kBucketBits = kTwoLevel ? 9 : (K == 12 ? 11 : 9);  // (fine) bucket bits of the histogram stage
    static constexpr int kHistBuckets = 1 << kBucketBits;
    static constexpr int kBuckets = kTwoLevel ? kCoarse * kRep : (1 << kBucketBits);   // ROWS of the scatter
    static constexpr int kLowBits = kTwoLevel ? 13 : 2 * K - kBucketBits;   // L: 13, 13, 11, 9, 7 (k = 12 .. 8); 13
    static constexpr int kFormBins = 1 << kLowBits;
    static constexpr int kSlots = kQuadRowWords / kBuckets;         // items per row / record: 16, 64; 128, 32
    static constexpr int kRecordBytes = kSlots * 4;
    // k = 12: an item needs 23 bits (hi6 | low13 | mask4), so it is stored in THREE bytes, five to a 16-byte vector:
    // four in the low three bytes of the vector's dwords, the fifth in the top bytes of the first three dwords -- a
    // 64-byte record holds 20 items instead of 16, a tile of 128 wave-steps fills it to 74 %, and the record pool
    // shrinks from 1.33 to 1.0 bytes per input byte (written once, read once).
#if defined(KPAL_QUAD_ITEM4)   // A/B builds
    static constexpr bool kItem3 = false;
#else
    static constexpr bool kItem3 = !kTwoLevel && K == 12;
#endif
    static constexpr int kItems = kItem3 ? (kSlots / 4) * 5 : kSlots;   // items a row / record holds
    // 64-byte records (k = 12): rows 2j and 2j+1 share a 128-byte line of the pool -- pool[row / 2][workgroup][round][row % 2] --
    // so that a flush writes whole lines (the two records of a line leave in the same store instruction)
    // (Tried instead: 1024 rows of 32 slots with the dropped bucket bit kept in the item -- fuller records, 128-byte
    // records by construction; the scatter gained nothing and the histogram, parsing every item twice, went from 4.0 to
    // 6.0 ms.)
#if defined(KPAL_QUAD_NO_PAIR)   // A/B builds
    static constexpr bool kPairRows = false;
#else
    static constexpr bool kPairRows = !kTwoLevel && kRecordBytes == 64;
#endif
    static constexpr int kScrBits = (kLowBits - 6) < 4 ? (kLowBits - 6) : 4;   // bits of the low field that scramble the bucket
    static constexpr uint64_t kXMask = (1ull << (2 * K + 6)) - 1ull;            // the (K+3)-mer: at most 38 bits
    static constexpr uint32_t kLowMask = (1u << kLowBits) - 1u;
    // bucket scrambling (see chunk_scramble): the top kScrBits of the low field pick one of 2^kScrBits masks
    __host__ __device__ static constexpr uint32_t smask(uint32_t t)
    {
        return kScrBits > 0 ? (((t << (kBucketBits - kScrBits)) | t) & (uint32_t)((1 << kBucketBits) - 1)) : 0u;
    }
    // the same for the coarse bucket of the two-level path
    // (XOR, not OR: at k = 14 the two copies of t overlap, and quad2_combine_kernel relies on smask1(a) ^ smask1(b) = smask1(a ^ b))
    __host__ __device__ static constexpr uint32_t smask1(uint32_t t) { return (kCoarseBits > 4 ? ((t << (kCoarseBits - 4)) ^ t) : t) & (uint32_t)(kCoarse - 1); }
};

static_assert(Quad2Index<13>::smask(11) == QuadCfg<13>::smask(11) && Quad2Index<16>::smask(5) == QuadCfg<16>::smask(5) &&
                  Quad2Index<13>::smask1(9) == QuadCfg<13>::smask1(9) && Quad2Index<14>::smask1(13) == QuadCfg<14>::smask1(13) &&
                  Quad2Index<15>::smask1(7) == QuadCfg<15>::smask1(7) && Quad2Index<16>::smask1(15) == QuadCfg<16>::smask1(15) &&
                  Quad2Index<15>::CB == QuadCfg<15>::kCoarseBits &&
                  ((Quad2Index<12>::smask1(13) << 9) | Quad2Index<12>::smask(13)) == QuadCfg<12>::smask(13) &&
                  ((Quad2Index<12>::smask1(6) << 9) | Quad2Index<12>::smask(6)) == QuadCfg<12>::smask(6),
              "quad2_index.hpp restates the scramble masks of QuadCfg");

// item = hi6 << (L+4) | low << 4 | mask4 (mask bit 3 = oldest k-mer).  0 = null item.
template <int K>
__device__ __forceinline__ void quad_split(uint64_t x, uint32_t m4, uint32_t lane, uint32_t &row, uint32_t &item)
{
    using C = QuadCfg<K>;
    const uint32_t low = (uint32_t)x & C::kLowMask;
    const uint32_t hi6 = (uint32_t)(x >> (2 * K));
    if constexpr (C::kTwoLevel) {
        const uint32_t fine = ((uint32_t)x >> 13) & 511u;
        const uint32_t coarse = (uint32_t)(x >> 22) & (uint32_t)(C::kCoarse - 1);
        row = (coarse ^ C::smask1(low >> 9)) * C::kRep + (lane & (uint32_t)(C::kRep - 1));
        item = (hi6 << 26) | (fine << 17) | (low << 4) | m4;
    } else {
        const uint32_t b = ((uint32_t)x >> C::kLowBits) & (uint32_t)(C::kBuckets - 1);
        row = b ^ C::smask(low >> (C::kLowBits - C::kScrBits));
        item = (hi6 << (C::kLowBits + 4)) | (low << 4) | m4;
    }
}

// the k-mer at position i (0 = oldest) of an item of (scrambled) row `row`.  LEVEL 2: an item of the second
// level of the two-level path (hi6 << 17 | low13 << 4 | mask4 in fine row `row` of scrambled coarse bucket `coarse`).
template <int K, int LEVEL = 1>
__device__ __forceinline__ uint32_t quad_kmer(uint32_t row, uint32_t item, int i, uint32_t coarse = 0)
{
    using C = QuadCfg<K>;
    uint64_t x;
    if constexpr (LEVEL == 2) {
        const uint32_t low = (item >> 4) & 8191u, hi6 = item >> 17, t = low >> 9;
        x = ((uint64_t)hi6 << (2 * K)) | ((uint64_t)(coarse ^ C::smask1(t)) << 22) | ((uint64_t)(row ^ C::smask(t)) << 13) | low;
    } else if constexpr (C::kTwoLevel) {
        const uint32_t low = (item >> 4) & 8191u, fine = (item >> 17) & 511u, hi6 = item >> 26, t = low >> 9;
        x = ((uint64_t)hi6 << (2 * K)) | ((uint64_t)((row / C::kRep) ^ C::smask1(t)) << 22) | ((uint64_t)fine << 13) | low;
    } else {
        const uint32_t low = (item >> 4) & C::kLowMask;
        const uint32_t hi6 = item >> (C::kLowBits + 4);
        const uint32_t b = row ^ C::smask(low >> (C::kLowBits - C::kScrBits));
        x = ((uint64_t)hi6 << (2 * K)) | ((uint64_t)b << C::kLowBits) | low;
    }
    return (uint32_t)((x >> (6 - 2 * i)) & ((1ull << (2 * K)) - 1ull));
}

// Items of at most 23 bits (level 2 of the two-level path) leave the CU PACKED: the four items of a 16-byte LDS vector as three
// dwords, a 64-item record as 192 bytes instead of 256 -- done on the way out, in registers, so the rows in LDS stay plain
// dword slots (the k = 12 one-level path packs in LDS instead, with riders: its rows are short).  Rows 2j and 2j+1 share a
// 384-byte piece (three whole lines) of the pool.
struct QuadPacked {
    uint32_t a, b, c;
};
__device__ __forceinline__ QuadPacked quad_pack3(const uint4 &v)
{
    QuadPacked r;
    r.a = v.x | (v.y << 23);
    r.b = (v.y >> 9) | (v.z << 14);
    r.c = (v.z >> 18) | (v.w << 5);
    return r;
}
__device__ __forceinline__ uint4 quad_unpack3(uint32_t a, uint32_t b, uint32_t c)
{
    uint4 v;
    v.x = a & 0x7FFFFFu;
    v.y = __builtin_amdgcn_alignbit(b, a, 23) & 0x7FFFFFu;
    v.z = __builtin_amdgcn_alignbit(c, b, 14) & 0x7FFFFFu;
    v.w = (c >> 5) & 0x7FFFFFu;
    return v;
}
constexpr int kQuadPackedRecordBytes = 192;   // 64 items

struct QuadSpill {
    uint32_t row, item;
};

// Items that can neither stay in their row nor ride in the spill list (a row that is over-full round after
// round: poly-A, satellite repeats, adapter / primer prefixes shared by many reads, a strongly skewed stretch) are
// counted here.  Such items are mostly IDENTICAL across lanes and rounds, and a global atomic per item on one
// table entry serialises chip-wide (measured: 10-40x slower on 2 % low-complexity reads or a shared 40-base
// prefix).  So they are counted in LDS: a 256-entry hash table (row, item) -> count per workgroup that is added
// to the table once, at the end of the kernel.
//   * up to eight rounds of "the first remaining lane's item, its occurrences in the wave counted with a ballot":
//     four lanes probe the item's four hash slots; a hit adds the count; a miss claims a free slot (64-bit
//     compare-and-swap of the key) if the item occurred at least twice in the wave (singletons -- the partly
//     masked items at the ends of a hot stretch -- would only fill the table), else takes global atomics;
//   * lanes left after eight rounds probe for themselves.
// Call wave-converged; `active` selects the lanes that hold an item.
struct QuadHot {
    unsigned long long key;   // row << 32 | item (an item always has mask bits set); 1 << 63 | table index: a single k-mer; 0 = free
    uint32_t count, pad;      // pad: k-mer entries -- the count when the entry was last looked at (ageing, quad_scatter_kernel<.., REPEAT>)
};
constexpr int kQuadHotEntries = 256;

// Where the counts that bypass the records go (hot items, items the spill list cannot hold, what is still carried when a
// scatter kernel ends, staged counts beyond 16 bits).  Classic: atomic adds into the count table.  FRESH (two-level pipeline,
// first piece of a count, kpal_quads2.hip): the table has NOT been zeroed and the finalisation will WRITE it without
// reading it -- the bypassing counts are appended to a list instead ((index << 32) | count entries: per scatter workgroup a
// segment with its counter in LDS, one shared segment with a global counter for the histogram stage) and added to the
// finished table by quad2_apply_list_kernel.  A full segment raises *overflow: the host then zeroes the table and runs the
// piece again the classic way.
struct TableSink {
    unsigned long long *table;      // classic target (nullptr in list mode)
    unsigned long long *list;       // list mode: this workgroup's segment
    uint32_t *count;                // entries appended so far (LDS or global)
    uint32_t cap;                   // entries a segment holds
    uint32_t *overflow;             // set when an append did not fit
};

// One-level pipeline (k <= 12): the table is always zeroed; nothing but its pointer travels through the kernels.
struct TableOnly {
    unsigned long long *table;
};

__device__ __forceinline__ void sink_add(const TableOnly &sink, uint64_t index, unsigned long long n) { atomicAdd(&sink.table[index], n); }

// Inside the scatter kernels the TableSink lies in LDS and the helpers carry a pointer to it: eight dwords by value through the
// placement code (and into the out-of-line quad_items_direct) cost the k >= 13 scatter 18 spilled registers.
struct TableSinkRef {
    const TableSink *p;
};

__device__ __forceinline__ void sink_add(const TableSink &sink, uint64_t index, unsigned long long n);
__device__ __forceinline__ void sink_add(const TableSinkRef &ref, uint64_t index, unsigned long long n) { sink_add(*ref.p, index, n); }

__device__ __forceinline__ void sink_add(const TableSink &sink, uint64_t index, unsigned long long n)
{
    if (sink.table) {               // (uniform)
        atomicAdd(&sink.table[index], n);
    } else {
        const uint32_t at = atomicAdd(sink.count, 1u);
        if (at < sink.cap) sink.list[at] = ((unsigned long long)index << 32) | (n & 0xFFFFFFFFull);
        else *sink.overflow = 1u;
    }
}

__device__ __forceinline__ uint32_t quad_hot_hash(uint32_t row, uint32_t item)
{
    return ((item >> 4) * 0x9E3779B1u + row * 0x85EBCA6Bu) >> 24;   // 8 bits
}

// Ageing of the k-mer entries of the hot-item table (quad_scatter_kernel<.., REPEAT>): thread i looks at entry i.  Out of line: it
// runs once per four tiles and must not cost the kernel registers.
template <typename SINK>
__device__ __attribute__((noinline)) void quad_hot_age(const SINK table, QuadHot *hot)
{
    const QuadHot h = hot[threadIdx.x];
    if (h.key >> 63) {
        if (h.count == h.pad) {
            sink_add(table, (uint64_t)(h.key & 0x7FFFFFFFFFFFFFFFull), (unsigned long long)h.count);
            hot[threadIdx.x] = QuadHot{0ull, 0u, 0u};
        } else {
            hot[threadIdx.x].pad = h.count;
        }
    }
}

// One k-mer of the direct path into the hot-item table as an entry keyed by its table index (see quad_items_direct_body), or, when
// its four probe slots hold other keys, into the count table.  (Inline in quad_items_direct_body, which is itself only reached through
// the out-of-line quad_items_direct*: a second level of calls would need a stack frame -- scratch memory in every kernel.)
template <typename SINK>
__device__ __forceinline__ void quad_kmer_add(const SINK table, QuadHot *hot, uint64_t index, uint32_t n)
{
#if defined(KPAL_AB_DIRECT_NO_TABLE)   // A/B timing builds (wrong counts): what the global atomics of the direct path cost
    return;
#endif
    const unsigned long long key = (1ull << 63) | (unsigned long long)index;
    const uint32_t h = ((uint32_t)index * 0x9E3779B1u + (uint32_t)(index >> 32)) >> 24;
    bool done = false;
#pragma unroll 1
    for (uint32_t pr = 0; pr < 4u && !done; ++pr) {
        const uint32_t slot = (h + pr) & (uint32_t)(kQuadHotEntries - 1);
        unsigned long long cur = hot[slot].key;
        if (cur == 0ull) {
            cur = atomicCAS(&hot[slot].key, 0ull, key);
            if (cur == 0ull) cur = key;
        }
        if (cur == key) {
            atomicAdd(&hot[slot].count, n);
            done = true;
        }
    }
    if (!done) sink_add(table, index, (unsigned long long)n);
}

template <int K, int LEVEL = 1, typename SINK = TableOnly, bool COUNTED = false, bool KMERS = false>
__device__ __forceinline__ void quad_items_direct_body(bool active, uint32_t row, uint32_t item, const SINK table, QuadHot *hot, uint32_t coarse = 0,
                                                       uint32_t cnt = 1u)
{
    // COUNTED: lane l stands for cnt (0 .. 4) occurrences of its item (the repeat lanes of quad_scatter_kernel)
    const uint32_t mult = COUNTED ? cnt : 1u;
    const int lane = threadIdx.x & 63;
    // An item the hot-item table does not hold (a singleton in its wave, no free slot) is counted k-mer by k-mer -- NOT straight into
    // the count table: the items that end here are those of persistently over-full rows, and their k-mers repeat over the whole
    // input even when the items do not (the flanks of poly-A stretches inside reads: A^11 X, A^10 XY ... -- a few dozen table entries
    // for the whole chip, on which global atomics serialise: a timing build without them ran every skewed input of tools/skewdiag.py
    // within 25 % of uniform reads, the committed kernel up to 6 x slower).  So the k-mers go into the SAME LDS table, as entries
    // keyed by the table index (bit 63 marks them), added to the count table once per workgroup; only when their four probe slots are
    // taken by other keys does a k-mer leave as a global atomic.
    // (KMERS = false: the inlined copies in the kernels' epilogues -- once per kernel, what is still carried at the end -- add to the
    // count table directly: the table code there cost the kernels spilled registers)
    auto kmer_add = [&](uint64_t index, uint32_t n) {
        if constexpr (KMERS) quad_kmer_add<SINK>(table, hot, index, n);
        else sink_add(table, index, (unsigned long long)n);
    };
    // (the rare tail below -- more than eight different items in one wave -- keeps the plain adds: a second copy of the probe loop
    // cost the callers of this function registers)
    auto to_table = [&](uint32_t r, uint32_t it, uint32_t n) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if ((it >> (3 - i)) & 1u) sink_add(table, quad_kmer<K, LEVEL>(r, it, i, coarse), (unsigned long long)n);
    };
    unsigned long long todo = __builtin_amdgcn_ballot_w64(active);
    for (int round = 0; round < 8 && todo; ++round) {   // wave-uniform
        const int src = __ffsll((long long)todo) - 1;
        const uint32_t hot_row = (uint32_t)__builtin_amdgcn_readlane(row, src);
        const uint32_t hot_item = (uint32_t)__builtin_amdgcn_readlane(item, src);
        const unsigned long long same = __builtin_amdgcn_ballot_w64(active && row == hot_row && item == hot_item) & todo;
        uint32_t n = (uint32_t)__popcll(same);
        if constexpr (COUNTED) {   // the sum of the lanes' counts: one ballot per value
            n = 0;
#pragma unroll
            for (uint32_t v = 1; v <= 4u; ++v) n += v * (uint32_t)__popcll(same & __builtin_amdgcn_ballot_w64(cnt == v));
        }
        const unsigned long long key = ((unsigned long long)hot_row << 32) | hot_item;
        const uint32_t slot = (quad_hot_hash(hot_row, hot_item) + (uint32_t)(lane & 3)) & (uint32_t)(kQuadHotEntries - 1);
        const unsigned long long seen = lane < 4 ? hot[slot].key : ~0ull;
        const unsigned long long hit = __builtin_amdgcn_ballot_w64(seen == key);
        if (hit) {
            if (lane == __ffsll((long long)hit) - 1) atomicAdd(&hot[slot].count, n);
        } else {
            const unsigned long long free_slots = __builtin_amdgcn_ballot_w64(seen == 0ull);
            bool placed = false;
            if (n >= 2u && free_slots) {
                const int who = __ffsll((long long)free_slots) - 1;
                // (another wave may claim the slot first, or insert the same key elsewhere: both harmless)
                const unsigned long long old = lane == who ? atomicCAS(&hot[slot].key, 0ull, key) : 1ull;
                placed = __builtin_amdgcn_ballot_w64(lane == who && (old == 0ull || old == key)) != 0ull;
                if (placed && lane == who) atomicAdd(&hot[slot].count, n);
            }
            if constexpr (KMERS) {
                if (!placed && lane < 4 && ((hot_item >> (3 - lane)) & 1u)) kmer_add(quad_kmer<K, LEVEL>(hot_row, hot_item, lane, coarse), n);   // lane i: k-mer i
            } else {
                if (!placed && lane == src) to_table(hot_row, hot_item, n);
            }
        }
        todo &= ~same;
    }
    if ((todo >> lane) & 1ull) {   // many different items in one wave: every lane for itself
        const unsigned long long key = ((unsigned long long)row << 32) | item;
        const uint32_t h = quad_hot_hash(row, item);
        bool done = false;
#pragma unroll
        for (int pr = 0; pr < 4 && !done; ++pr) {
            const uint32_t slot = (h + (uint32_t)pr) & (uint32_t)(kQuadHotEntries - 1);
            if (hot[slot].key == key) {
                atomicAdd(&hot[slot].count, mult);
                done = true;
            }
        }
        if (!done) to_table(row, item, mult);
    }
}

// (rare path, called from the unrolled placement loop: kept out of line there; the scatter kernels' epilogues inline the body -- an
// out-of-line call next to everything that is live there cost spilled registers, and kernels that use scratch memory at all ran
// 8 % slower in same-box comparisons, wherever the spill sat)
// KM: with the k-mer entries of the hot-item table (quad_kmer_add).  Only the REPEAT instantiations of quad_scatter_kernel ask for it --
// the input whose sample showed hot rows; the larger function costs its callers a few spilled registers (16-32 bytes of scratch),
// which uniform reads do not pay.
template <int K, int LEVEL = 1, typename SINK = TableOnly, bool KM = false>
__device__ __attribute__((noinline)) void quad_items_direct(bool active, uint32_t row, uint32_t item, const SINK table, QuadHot *hot, uint32_t coarse = 0)
{
    quad_items_direct_body<K, LEVEL, SINK, false, KM>(active, row, item, table, hot, coarse);
}

// (the repeat lanes: lane l holds its item cnt times)
template <int K, typename SINK>
__device__ __attribute__((noinline)) void quad_items_direct_counted(uint32_t cnt, uint32_t row, uint32_t item, const SINK table, QuadHot *hot)
{
    quad_items_direct_body<K, 1, SINK, true, true>(cnt != 0u, row, item, table, hot, 0u, cnt);
}

// Returns the mask (bit q) of this lane's items that did not fit their row.  pos[row] counts the BYTES in use of
// the row (the atomic returns the item's byte offset: one shift-add gives its LDS address).
template <int K, bool DIRECT = false, int LEVEL = 1, int N = 4, typename SINK = TableOnly, bool KM = false>
__device__ __forceinline__ uint32_t quad_place(uint32_t *rows, uint32_t *pos, QuadSpill *spill, uint32_t *spill_n, uint32_t cap,
                                               const uint32_t (&row)[N], const uint32_t (&item)[N], const SINK &table,
                                               QuadHot *hot, uint32_t coarse = 0)
{
    using C = typename std::conditional<LEVEL == 2, QuadCfg<11>, QuadCfg<K>>::type;   // level 2: 512 rows x 64 slots
    constexpr uint32_t RB = (uint32_t)C::kSlots * 4u;                                  // bytes per row
    constexpr bool ITEM3 = LEVEL == 1 && C::kItem3;                                    // three-byte items: 20 to a 64-byte row
    uint32_t riders = 0;
    uint32_t off[N];
#pragma unroll
    for (int q = 0; q < N; ++q) off[q] = atomicAdd(&pos[row[q]], (item[q] & 15u) ? 4u : 0u);
    uint32_t over = 0;
#pragma unroll
    for (int q = 0; q < N; ++q) {
        const bool counted = (item[q] & 15u) != 0u;
        if constexpr (ITEM3) {
            // slots 0..15: the low three bytes of dword s of the row; slots 16..19 ride in the top bytes (see below)
            const bool normal = off[q] < RB;
            const bool rider = !normal && off[q] < RB + RB / 4;
            if (counted && normal) atomicOr(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(rows) + row[q] * RB + off[q]), item[q]);
            riders |= (counted && rider) ? (1u << q) : 0u;
            over |= (counted && !normal && !rider) ? (1u << q) : 0u;
        } else {
            const bool fits = off[q] < RB;
            const uint32_t at = row[q] * RB + off[q];
            *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(rows) + ((counted && fits) ? at : (uint32_t)kQuadRowWords * 4u)) = item[q];   // not counted / full row: dummy word
            over |= (counted && !fits) ? (1u << q) : 0u;
        }
    }
#if defined(KPAL_AB_SCATTER_NO_RIDER)   // A/B timing builds (wrong counts): what the rider path costs
    riders = 0;
#endif
#if defined(KPAL_AB_SCATTER_NO_SPILL)   // A/B timing builds (wrong counts): what the overflow path costs
    over = 0;
#endif
    if constexpr (ITEM3) {
        // Item 16 + r of a row rides in the top bytes of dwords 4r, 4r+1, 4r+2 (one byte each; a 23-bit item leaves the
        // top byte of its own dword free).  Everything is OR-ed into rows that the flush left zeroed, so the order in which
        // the lanes of different waves reach a dword does not matter.  ~15 % of the items at the usual fill: a few lanes
        // per instruction, where three byte stores for EVERY item kept the LDS busy 63 % of the time on bank conflicts.
        if (__any(riders != 0u)) {   // wave-uniform
#pragma unroll
            for (int q = 0; q < N; ++q)
                if ((riders >> q) & 1u) {
                    uint32_t *p = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(rows) + row[q] * RB + (off[q] - RB) * 4u);
#if defined(KPAL_AB_RIDER_OR)    // A/B: three read-modify-writes (the round-2 form; same-box 7.45 against 7.28 ms for the scatter)
                    atomicOr(p, item[q] << 24);
                    atomicOr(p + 1, (item[q] << 16) & 0xFF000000u);
                    atomicOr(p + 2, (item[q] << 8) & 0xFF000000u);
#else
                    // byte stores: nobody else writes these bytes, and an LDS read-modify-write of a dword's low bytes (the OR of
                    // its own item, above) is indivisible against them
                    reinterpret_cast<unsigned char *>(p)[3] = (unsigned char)item[q];
                    reinterpret_cast<unsigned char *>(p)[7] = (unsigned char)(item[q] >> 8);
                    reinterpret_cast<unsigned char *>(p)[11] = (unsigned char)(item[q] >> 16);
#endif
                }
        }
    }
    if (__builtin_expect(__any(over != 0u), 0)) {   // wave-uniform
#if defined(KPAL_QUAD_NO_CARRY)   // bisecting builds only: spilled items straight into the table
        constexpr bool direct = true;
#else
        constexpr bool direct = DIRECT;
#endif
#if defined(KPAL_AB_SPILL_LANE)   // A/B: every overflowing lane reserves its own list entries (a returning LDS atomic per item, few lanes)
        if constexpr (!direct) {
            uint32_t at[N];
#pragma unroll
            for (int q = 0; q < N; ++q) at[q] = cap;
            if (over) {
#pragma unroll
                for (int q = 0; q < N; ++q)
                    if ((over >> q) & 1u) at[q] = atomicAdd(spill_n, 1u);
            }
            uint32_t unlisted = 0;
#pragma unroll
            for (int q = 0; q < N; ++q) {
                const bool ov = (over >> q) & 1u;
                if (ov && at[q] < cap) spill[at[q]] = QuadSpill{row[q], item[q]};
                unlisted |= (ov && at[q] >= cap) ? (1u << q) : 0u;
            }
            if (__builtin_expect(__any(unlisted != 0u), 0)) {
#pragma unroll
                for (int q = 0; q < N; ++q)
                    if (__any((unlisted >> q) & 1u)) quad_items_direct<K, LEVEL, SINK, KM>((unlisted >> q) & 1u, row[q], item[q], table, hot, coarse);
            }
            return over;
        }
#endif
        // one LDS atomic per call reserves the list entries of all the wave's overflowed items (a lane finds its own
        // with ballots and lane counts)
        uint32_t base = 0;
        bool list_full = true;                       // (wave-uniform) something may be left to count directly
        unsigned long long bq[N];                    // (scalar registers: who overflowed at position q)
#pragma unroll
        for (int q = 0; q < N; ++q) bq[q] = __builtin_amdgcn_ballot_w64((over >> q) & 1u);
        if constexpr (!direct) {
            uint32_t total = 0;
#pragma unroll
            for (int q = 0; q < N; ++q) total += (uint32_t)__popcll(bq[q]);
            uint32_t got = 0;
            if ((threadIdx.x & 63u) == 0u) got = atomicAdd(spill_n, total);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
            list_full = base + total > cap;          // (one scalar test instead of a ballot per position below)
        }
#pragma unroll
        for (int q = 0; q < N; ++q) {
            if (bq[q] == 0ull) continue;             // wave-uniform: at the usual fill a position overflows in one wave-step of three
            const bool ov = (over >> q) & 1u;
            bool listed = false;
            if constexpr (!direct) {
                const unsigned long long b = bq[q];
                const uint32_t at = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                listed = ov && at < cap;
                if (listed) spill[at] = QuadSpill{row[q], item[q]};
                base += (uint32_t)__popcll(b);
            }
            // carried items that still do not fit, and whatever the list cannot hold: counted now
            if (list_full && __any(ov && !listed)) quad_items_direct<K, LEVEL, SINK, KM>(ov && !listed, row[q], item[q], table, hot, coarse);
        }
    }
    return over;
}

// The items a thread carries from one round to the next (entries threadIdx.x + c * THREADS of the spill list).
// Plain functions on array references: as [&] lambdas the arrays were kept in scratch memory.
template <int K, int CARRY, int LEVEL = 1, typename SINK = TableOnly, bool KM = false>
__device__ __forceinline__ void quad_place_carried(uint32_t *rows, uint32_t *pos, QuadSpill *spill, uint32_t *spill_n, uint32_t cap,
                                                   const uint32_t (&carry_row)[CARRY], uint32_t (&carry_item)[CARRY],
                                                   const SINK &table, QuadHot *hot, uint32_t coarse = 0)
{
#pragma unroll
    for (int c = 0; c < CARRY; ++c) {
        const uint32_t r1[1] = {carry_row[c]};
        const uint32_t i1[1] = {carry_item[c]};
        if (__any(carry_item[c] != 0u)) {
            // an item that does not fit even now has been counted: it is no longer carried
            if (quad_place<K, true, LEVEL, 1, SINK, KM>(rows, pos, spill, spill_n, cap, r1, i1, table, hot, coarse) & 1u) carry_item[c] = 0;
        }
    }
}

template <int CARRY, int THREADS>
__device__ __forceinline__ void quad_take_carried(const QuadSpill *spill, uint32_t n, uint32_t (&carry_row)[CARRY],
                                                  uint32_t (&carry_item)[CARRY])
{
#pragma unroll
    for (int c = 0; c < CARRY; ++c) {
        const uint32_t e = threadIdx.x + (uint32_t)c * THREADS;
        carry_item[c] = 0;
        if (e < n) {
            carry_row[c] = spill[e].row;
            carry_item[c] = spill[e].item;
        }
    }
}

// Wave priority inside a tile of the scatter kernels.  The sixteen waves of a workgroup -- four per SIMD -- have the same STEPS
// wave-steps to do between two barriers, but a SIMD's arbiter prefers its oldest wave: the four finished one after the other, and
// per-phase cycle counters (round 4) showed a wave waiting at the barrier behind the placement for 27 % of the kernel while the
// last wave of its SIMD ran alone, every LDS round trip exposed.  So a wave's priority FALLS as it advances through the tile
// (s_setprio 3, 2, 1, 0 over the four quarters): whoever is behind is served first and the four arrive together.  k = 12, 100 M
// reads: 6.87 -> 6.03 ms; level 1 of k = 15: 6.24 -> 5.61 ms (same-box A/B against -DKPAL_QUAD_NO_PRIO).
template <int STEPS>
__device__ __forceinline__ void quad_tile_priority(int st)
{
#if !defined(KPAL_QUAD_NO_PRIO)   // A/B builds
    const int level = st * 4 / STEPS;             // (the unrolled step loop folds this: s_setprio takes an immediate)
    if (st == 0 || level != (st - 1) * 4 / STEPS) {
        switch (level) {
        case 0: __builtin_amdgcn_s_setprio(3); break;
        case 1: __builtin_amdgcn_s_setprio(2); break;
        case 2: __builtin_amdgcn_s_setprio(1); break;
        default: __builtin_amdgcn_s_setprio(0); break;
        }
    }
#endif
}

// Statistics of the slow paths (kpal_count_stats; tests assert that skewed inputs really took them): error[2] += items that rode in
// the spill list this round, error[3] += items the list could not hold (counted on the spot: hot-item table or the table itself).
// One fire-and-forget atomic per workgroup and round that spilled at all.
__device__ __forceinline__ void quad_note_spill(uint32_t *__restrict__ error, uint32_t appended, uint32_t cap)
{
    if (threadIdx.x == 0 && appended) {
        atomicAdd(error + 2, min(appended, cap));
        if (appended > cap) atomicAdd(error + 3, appended - cap);
    }
}

// Q0: row loads of a sample of the input.  Workgroup g encodes `steps` wave-steps starting at its share of the
// stream and adds every item to load[row] (global atomics: a few hundred thousand).  The host picks the tile size
// from these loads: rows that would be over-full every round (compositional skew, e.g. an AT-rich genome) call
// for smaller tiles, a handful of very hot rows (poly-A, satellites) are left to the spill list.
template <int K>
__global__ __launch_bounds__(512) void quad_sample_kernel(Span s, uint64_t stride_steps, uint32_t steps, uint32_t *__restrict__ load)
{
    using C = QuadCfg<K>;
    // two-level path: load[kBuckets ..] also receives the loads of the 512 FINE rows of level 2 (over all coarse buckets);
    // the last word counts the items of REPEAT LANES (quad_scatter_kernel: four identical items -- they never reach a row there,
    // so they do not count towards a row's load here either; their number tells the host to launch the REPEAT instantiation)
    constexpr int NFINE = C::kTwoLevel ? 512 : 0;
    constexpr int NCNT = C::kBuckets + NFINE + 1;
    __shared__ uint32_t cnt[NCNT];
    for (int i = threadIdx.x; i < NCNT; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const uint64_t total_steps = (s.nchunks + 63) / 64;
    const uint64_t first = (uint64_t)blockIdx.x * stride_steps + (uint64_t)wave * steps;
    if (first < total_steps) {
        Chunk carry = load_chunk(s, (int64_t)(first * 64) - 1);
        for (uint32_t st = 0; st < steps && first + st < total_steps; ++st) {
            uint64_t window;
            uint32_t mask;
            part_step<K>(s, first + st, carry, window, mask);
            uint32_t row[4], item[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint64_t x = (window >> (24 - 8 * q)) & C::kXMask;
                quad_split<K>(x, (mask >> (12 - 4 * q)) & 15u, threadIdx.x & 63u, row[q], item[q]);
            }
            {   // the wave's repeat item, as quad_scatter_kernel<.., REPEAT> finds it: its occurrences never reach a row
                const bool c01 = item[1] == item[0] && row[1] == row[0] && (item[1] & 15u) == 15u;
                const bool c23 = item[3] == item[2] && row[3] == row[2] && (item[3] & 15u) == 15u;
                const unsigned long long cand = __builtin_amdgcn_ballot_w64(c01 || c23);
                if (cand) {
                    const int src = __ffsll((long long)cand) - 1;
                    const uint32_t hot_item = (uint32_t)__builtin_amdgcn_readlane(c01 ? item[1] : item[3], src);
                    const uint32_t hot_row = (uint32_t)__builtin_amdgcn_readlane(c01 ? row[1] : row[3], src);
                    uint32_t n = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool m = item[q] == hot_item && row[q] == hot_row;
                        n += m ? 1u : 0u;
                        item[q] = m ? 0u : item[q];
                    }
                    if (n) atomicAdd(&cnt[NCNT - 1], n);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                atomicAdd(&cnt[row[q]], (item[q] & 15u) ? 1u : 0u);
                if constexpr (C::kTwoLevel) {   // the row quad2_scatter_kernel gives the item
                    const uint32_t fine_row = ((item[q] >> 17) & 511u) ^ QuadCfg<11>::smask(((item[q] >> 4) & 8191u) >> 9);
                    atomicAdd(&cnt[C::kBuckets + fine_row], (item[q] & 15u) ? 1u : 0u);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NCNT; i += blockDim.x)
        if (cnt[i]) atomicAdd(&load[i], cnt[i]);
}

// One wave-step of input: 16 bytes per lane of step `step` (wave-uniform).  Interior steps are one load from a
// scalar base address + the lane's constant byte offset; a step that crosses the end of the buffer checks per lane.
__device__ __forceinline__ uint4 fetch_wave_step(const Span &s, uint64_t step, uint32_t lane16)
{
    if ((step + 1) * 64 <= s.nchunks)   // wave-uniform
        return *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(s.base + step * 64) + lane16);
    return fetch_chunk(s, (int64_t)(step * 64 + (lane16 >> 4)));
}

// The flush of both scatter kernels: the 128 KiB of rows leave LDS as 8192 16-byte vectors, FI per thread, and
// are zeroed behind the read -- the rows of the next round start empty, so whatever lies beyond a row's last
// item is the null padding of its record (no per-item compare against the row's fill); pos is cleared likewise.
// (Written out in both kernels: as a function taking rec[] by reference the array was kept in scratch memory.)

// Q1: ASCII -> records.  Tile j of workgroup g is tile j * G + g of the input (the grid reads one sliding
// window); wave w takes its steps STEPS*w .. STEPS*w+STEPS-1.  A tile of WAVES x STEPS KiB brings
// 0.119 x WAVES x STEPS items per 16-slot row at k = 12 (96 steps: 11.4, records 71 % full, 1.5 % of the items
// spill; 112: 13.3, 83 %, 4 %; 120: 14.3, 89 %, 6 %).  Per tile: place (the items
// carried over from the previous round first), barrier, read every row as one padded record into registers,
// barrier.  The records are STORED during the placement of the next tile, a store instruction or two per
// step, and each step's chunk of the next tile is requested as soon as this tile's has been encoded: the
// CU's memory pipe (8-10 B/clk for loads + stores together, tools/store_probe2.hip) stays busy under the
// LDS / VALU work instead of alternating with it.  WAVES = 8: two waves per SIMD with ~200 registers each
// (16 record vectors + a tile of prefetched chunks live).
//   The kernel issues ~2 VALU instructions per input byte and lane, so everything wave-uniform is kept scalar:
// the wave index is read with readfirstlane (tile / step numbers, the edge tests of encode_step and the load
// addresses become SALU work; a load is saddr + lane * 16), and a record store is a scalar base (row group, workgroup,
// round) + one per-thread 32-bit offset fixed for the whole kernel.
// pool word address of record (row, g, round): ((row * G + g) * rounds_cap + round) * kSlots
// (k = 12, kPairRows: (((row / 2) * G + g) * rounds_cap + round) * 32 + (row % 2) * 16).
template <int K, int WAVES, int STEPS, int DEPTH, typename SINK = TableOnly, bool REPEAT = false>
__global__ __launch_bounds__(WAVES * 64) void quad_scatter_kernel(Span s, uint64_t tiles_per_block, uint32_t *__restrict__ pool,
                                                                  uint32_t rounds_cap, uint32_t *__restrict__ nrounds,
                                                                  uint32_t *__restrict__ error, SINK sink_arg)
{
    using C = QuadCfg<K>;
    constexpr bool kLists = std::is_same<SINK, TableSink>::value;
    constexpr int S = C::kSlots, NB = C::kBuckets;
    constexpr int THREADS = WAVES * 64;
    constexpr int kQuadTileSteps = WAVES * STEPS;
    constexpr int CARRY = kQuadSpillCap / THREADS;      // carried items per thread
    constexpr uint32_t CAP = CARRY * THREADS;           // spill list entries in use
    static_assert((WAVES == 8 || WAVES == 16) && STEPS % DEPTH == 0, "tile shape");
    __shared__ __attribute__((aligned(16))) uint32_t rows[kQuadRowWords + kQuadDummyWords];
    __shared__ __attribute__((aligned(16))) uint32_t pos[NB];   // BYTES in use per row
    __shared__ QuadSpill spill[kQuadSpillCap];
    __shared__ uint32_t spill_cnt[2];           // appended-entries counter of even / odd tiles: the one of tile j is read by every
                                                // thread after the placement barrier, so it may only be reset a barrier later --
                                                // thread 0 resets the OTHER one (for tile j+1) during the flush of tile j
    __shared__ QuadHot hot[kQuadHotEntries];    // items of persistently over-full rows, counted here instead of in the table
    __shared__ uint32_t direct_n;               // list mode (TableSink): entries this workgroup appended to its segment
    __shared__ TableSink sink_lds;              // (k >= 13) the sink as this workgroup uses it
    using SINK2 = typename std::conditional<kLists, TableSinkRef, SINK>::type;
    SINK2 table;
    if constexpr (kLists) {
        if (threadIdx.x == 0) {
            TableSink t = sink_arg;
            if (t.list) {                       // this workgroup's segment, counted in LDS
                t.list += (size_t)blockIdx.x * t.cap;
                t.count = &direct_n;
            }
            sink_lds = t;
            direct_n = 0;
        }
        table = TableSinkRef{&sink_lds};
    } else {
        table = sink_arg;
    }
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const uint32_t lane16 = (uint32_t)lane * 16u;
    for (int i = threadIdx.x; i < kQuadRowWords + kQuadDummyWords; i += THREADS) rows[i] = 0;
    for (int i = threadIdx.x; i < NB; i += THREADS) pos[i] = 0;
    for (int i = threadIdx.x; i < kQuadHotEntries; i += THREADS) hot[i] = QuadHot{0ull, 0u, 0u};
    if (threadIdx.x == 0) {
        spill_cnt[0] = 0;
        spill_cnt[1] = 0;
    }
    __syncthreads();
    const uint64_t total_steps = (s.nchunks + 63) / 64;
    auto tile_step = [&](uint64_t j) -> uint64_t { return ((j * gridDim.x + blockIdx.x) * WAVES + (uint64_t)wave) * STEPS; };
    auto tile_exists = [&](uint64_t j) -> bool { return j < tiles_per_block && (j * gridDim.x + blockIdx.x) * (uint64_t)kQuadTileSteps < total_steps; };
    // input chunks are requested DEPTH steps ahead (DEPTH KiB per wave in flight) into a ring of DEPTH
    // register quadruples; DEPTH divides STEPS, so the ring position of a step does not depend on the tile
    uint4 raw[DEPTH];
    uint4 rawh;
    {
        const uint64_t f = tile_step(0);
#pragma unroll
        for (int st = 0; st < DEPTH; ++st) raw[st] = fetch_wave_step(s, f + st, lane16);
        rawh = fetch_chunk(s, (int64_t)(f * 64) - 1);
    }
    uint32_t round = 0;                         // records written per row so far
    uint32_t carry_row[CARRY], carry_item[CARRY];   // items carried over from the previous round (item 0 = none)
#pragma unroll
    for (int c = 0; c < CARRY; ++c) carry_row[c] = carry_item[c] = 0;
    // the flush: FI store instructions per wave, each writing 64 / (S/4) whole records with 16-byte stores;
    // thread t handles the 16-byte vectors t, t + THREADS, ... of the 8192 that make up the rows
    constexpr int LPR = S / 4;                  // lanes (vectors) per record
    constexpr int FI = kQuadRowWords / 4 / THREADS;   // 16
    static_assert(THREADS % LPR == 0, "a record is written by one wave");
    uint4 rec[FI];                              // the previous round's records, stored during this tile's placement
    bool have_rec = false;                      // block-uniform
    // vector t + i * THREADS is vector t % LPR of row t / LPR + i * (THREADS / LPR): the thread's part of the address
    // (32 bits: the host keeps a pool below 64 GiB) never changes, the rest is scalar
    // (kPairRows: a "row" of the address arithmetic is a PAIR of rows with 128-byte records)
    constexpr int PAIR = C::kPairRows ? 2 : 1;
    static_assert((THREADS / LPR) % PAIR == 0, "row pairs are written by one instruction");
    const uint64_t row_bytes = (uint64_t)gridDim.x * rounds_cap * (uint64_t)(S * 4 * PAIR);
    const uint32_t thread_off = (uint32_t)((uint64_t)(threadIdx.x / (LPR * PAIR)) * row_bytes) + (threadIdx.x % (LPR * PAIR)) * 16u;
    auto store_rec = [&](int i) {
#if defined(KPAL_AB_POOL_FILL)   // A/B timing builds (wrong counts): what records filled to 0.86 instead of 0.74 would store -- six of seven vectors
        if (i % 7 == 6) return;
#endif
        uint32_t o = thread_off;
        asm volatile("" : "+v"(o));             // (no FI hoisted 64-bit addresses)
        const uint64_t sc = (uint64_t)(i * (THREADS / (LPR * PAIR))) * row_bytes + ((uint64_t)blockIdx.x * rounds_cap + (round - 1u)) * (uint64_t)(S * 4 * PAIR);
#if defined(KPAL_QUAD_NT)   // A/B builds: non-temporal record stores
        uint32_t *q = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(pool) + sc + o);
        __builtin_nontemporal_store(rec[i].x, q);
        __builtin_nontemporal_store(rec[i].y, q + 1);
        __builtin_nontemporal_store(rec[i].z, q + 2);
        __builtin_nontemporal_store(rec[i].w, q + 3);
#elif defined(KPAL_AB_SCATTER_NO_STORE)   // A/B timing builds (wrong counts): no record stores
        asm volatile("" ::"v"(rec[i].x ^ rec[i].y ^ rec[i].z ^ rec[i].w), "v"(o), "s"((uint32_t)sc));
#else
        *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(pool) + sc + o) = rec[i];
#endif
    };
    auto store_rec_at = [&](int i, const uint4 &v) {   // record vector i of round `round`, i not a compile-time constant
        const uint64_t sc = (uint64_t)i * (uint64_t)(THREADS / (LPR * PAIR)) * row_bytes + ((uint64_t)blockIdx.x * rounds_cap + (round - 1u)) * (uint64_t)(S * 4 * PAIR);
        *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(pool) + sc + thread_off) = v;
    };
    for (uint64_t j = 0; tile_exists(j); ++j) {   // block-uniform
        const uint64_t first = tile_step(j);
        const bool more = tile_exists(j + 1);
        const uint64_t fnext = tile_step(j + 1);
        // one range test per wave and tile instead of one per step (64-bit compares are VALU work)
        const bool edge = !interior_steps(s, first, first + STEPS);
        const bool next_inside = more && (fnext + STEPS) * 64 <= s.nchunks;
        uint32_t *spill_n = &spill_cnt[j & 1];
        // ---- place: carried items, then this tile's
        quad_place_carried<K, CARRY, 1, SINK2, REPEAT>(rows, pos, spill, spill_n, CAP, carry_row, carry_item, table, hot);
        Chunk carry = encode16(rawh);
        range_fix(s, (int64_t)(first * 64) - 1, carry);
        if (more) rawh = fetch_chunk(s, (int64_t)(fnext * 64) - 1);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            uint64_t window;
            uint32_t mask;
            quad_tile_priority<STEPS>(st);
            encode_step<K>(s, first + st, raw[st % DEPTH], carry, window, mask, edge);
            // the previous round's records leave AFTER this step's chunk has been consumed: the compiler guards the first use of
            // raw[] in a tile with s_waitcnt vmcnt(0) (the loads were issued a tile ago, under conditions it cannot count) --
            // with a record store issued just before, every wave of the workgroup waited out that store's round trip at the
            // top of every tile, right after the barrier, all at the same time
            if (have_rec) {   // FI / STEPS records per step; the FI % STEPS left over went out with the flush itself
#pragma unroll
                for (int i = st * (FI / STEPS); i < (st + 1) * (FI / STEPS); ++i) store_rec(i);
            }
            {
                uint32_t l16 = lane16;
                asm volatile("" : "+v"(l16));
                if (st + DEPTH < STEPS) raw[st % DEPTH] = fetch_wave_step(s, first + st + DEPTH, l16);
                else if (next_inside) raw[st % DEPTH] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(s.base + (fnext + (st + DEPTH - STEPS)) * 64) + l16);
                else if (more) raw[st % DEPTH] = fetch_chunk(s, (int64_t)((fnext + (st + DEPTH - STEPS)) * 64 + (l16 >> 4)));
            }
            uint32_t row[4], item[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint64_t x = (window >> (24 - 8 * q)) & C::kXMask;
                quad_split<K>(x, (mask >> (12 - 4 * q)) & 15u, (uint32_t)lane, row[q], item[q]);
            }
            // REPEAT LANES (REPEAT: the instantiation the host launches when the sample of the feed shows hot rows -- on uniform reads
            // the test below and the registers of its call cost the 8-step tile of k = 12 1.4 %, same-box A/B).  Inside a homopolymer
            // or a repeat of period 2 or 4 (poly-A tails, (AC)n, (ACGT)n ...) the four items of a lane are one and the same: all such
            // lanes of the reads of a tile want the same row, round after round -- that row overflows, its items ride in the spill
            // list, fail again as carried items and end in the hot-item table one by one (2 % of such reads cost the kernel 3.4 x).
            // They go there at once: the item of the first lane that holds two equal neighbours is the wave's repeat item, EVERY
            // occurrence of it in the wave -- also the one to three in the lanes at the ends of the read, which alone kept the row
            // over-full -- is counted by one update of the hot-item table, and none of them enters the rows.
            bool place = true;                   // wave-uniform
            if constexpr (REPEAT) {
                // (two compares per step on the fast path: equal neighbours are the necessary condition -- 2^-23 per lane by chance)
                if (__builtin_expect(__any(item[1] == item[0] || item[3] == item[2]), 0)) {   // wave-uniform
                    const bool c01 = item[1] == item[0] && row[1] == row[0] && (item[1] & 15u) == 15u;
                    const bool c23 = item[3] == item[2] && row[3] == row[2] && (item[3] & 15u) == 15u;
                    const unsigned long long cand = __builtin_amdgcn_ballot_w64(c01 || c23);
                    if (cand) {
                        // the wave's repeat item: that of the first such lane (a second low-complexity read in the same KiB takes the rows)
                        const int src = __ffsll((long long)cand) - 1;
                        const uint32_t hot_item = (uint32_t)__builtin_amdgcn_readlane(c01 ? item[1] : item[3], src);
                        const uint32_t hot_row = (uint32_t)__builtin_amdgcn_readlane(c01 ? row[1] : row[3], src);
                        uint32_t cnt = 0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const bool m = item[q] == hot_item && row[q] == hot_row;
                            cnt += m ? 1u : 0u;
                            item[q] = m ? 0u : item[q];
                        }
                        quad_items_direct_counted<K, SINK2>(cnt, hot_row, hot_item, table, hot);
                        // (a step that held nothing else -- a homopolymer, the inside of a long repeat -- has nothing to place: the slot
                        // atomics of null items add 0, but sixty-four lanes on ONE row counter still serialise in the LDS)
                        place = __any((item[0] | item[1] | item[2] | item[3]) != 0u);
                    }
                }
            }
#if defined(KPAL_AB_SCATTER_NO_PLACE)    // A/B timing builds (wrong counts): loads + encode + split + flush only
            asm volatile("" ::"v"(row[0] ^ row[1] ^ row[2] ^ row[3] ^ item[0] ^ item[1] ^ item[2] ^ item[3]));
#else
            if (place) quad_place<K, false, 1, 4, SINK2, REPEAT>(rows, pos, spill, spill_n, CAP, row, item, table, hot);
#endif
        }
        have_rec = false;
        quad_tile_priority<1>(0);                // (the flush and the carried items: everybody is needed at the next barrier)
        lds_barrier();                           // rows, pos and the spill list are complete (loads and stores stay in flight)
        const uint32_t spilled = min(*spill_n, CAP);         // (beyond CAP: counted directly by quad_place); reset only during the flush of the NEXT tile
        quad_note_spill(error, *spill_n, CAP);
        if (threadIdx.x == 0) spill_cnt[(j & 1) ^ 1] = 0;   // last read before the barrier that ended the previous tile
        if constexpr (REPEAT) {
            // The k-mer entries of the hot-item table AGE: every fourth tile an entry that has not been added to since the last look
            // (pad = its count then) goes to the count table and frees its slot -- the table admits every k-mer of the direct path,
            // hot or not, and without this it was full of cold ones after a few tiles (all 256 entries in use in every case of
            // tools/skewdiag.py).  Nobody touches the table between the two barriers of the flush.
            if ((j & 3u) == 3u && threadIdx.x < (uint32_t)kQuadHotEntries) quad_hot_age<SINK2>(table, hot);   // (wave-uniform: waves 0..3)
        }
        // ---- flush: every row becomes one record of S items (null padded), kept in registers
        quad_take_carried<CARRY, THREADS>(spill, spilled, carry_row, carry_item);
        {
            const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
            uint4 *rv = reinterpret_cast<uint4 *>(rows);
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                const uint4 v = rv[threadIdx.x + (uint32_t)i * THREADS];
                rec[i].x = v.x;
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
            // (records the next tile's steps do not take -- 16 records over 7, 6 or 3 steps -- leave now: steps that store one
            // record more than others cost those variants spilled registers)
#pragma unroll
            for (int i = STEPS * (FI / STEPS); i < FI; ++i) store_rec(i);
        }
        lds_barrier();                           // every wave has its records in registers: the rows are free
    }
    if (have_rec) {
#pragma unroll
        for (int i = 0; i < STEPS * (FI / STEPS); ++i) store_rec(i);
    }
    if constexpr (kLists) {
        // ---- tail round (two-level path): what the last tile left in the spill list goes through the rows once more and leaves as
        // one more round of (mostly empty) records -- 128 KiB per workgroup -- instead of entry by entry into the table or, FRESH,
        // into the list that quad2_apply_list_kernel works off with atomics (k = 15: 0.6 ms for ~5 M entries, nearly all from here)
        if (round > 0u && round < rounds_cap) {   // block-uniform (no vote on whether anything is carried: at steady state something always is)
            static_assert(!C::kItem3, "plain four-byte slots");
#pragma unroll
            for (int c = 0; c < CARRY; ++c)
                if (carry_item[c]) {             // (the rows are empty: an item that finds its row full all the same stays carried and is counted below)
                    const uint32_t off = atomicAdd(&pos[carry_row[c]], 4u);
                    if (off < (uint32_t)S * 4u) {
                        *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(rows) + carry_row[c] * (uint32_t)(S * 4) + off) = carry_item[c];
                        carry_item[c] = 0;
                    }
                }
            lds_barrier();
            const uint4 *rv = reinterpret_cast<const uint4 *>(rows);
            ++round;
#pragma unroll 1
            for (int h = 0; h < FI; h += 4) {    // (a few records at a time: nothing here may cost the main loop registers)
#pragma unroll
                for (int i = 0; i < 4; ++i) rec[i] = rv[threadIdx.x + (uint32_t)(h + i) * THREADS];
#pragma unroll
                for (int i = 0; i < 4; ++i) store_rec_at(h + i, rec[i]);
            }
        }
    }
    if (threadIdx.x == 0) nrounds[blockIdx.x] = min(round, rounds_cap);
    // what is still carried over, then the table of hot items: into the count table
#pragma unroll
    for (int c = 0; c < CARRY; ++c) quad_items_direct_body<K, 1, SINK2>(carry_item[c] != 0u, carry_row[c], carry_item[c], table, hot);
    __syncthreads();
    uint32_t used = 0;
    for (int i = threadIdx.x; i < kQuadHotEntries; i += THREADS) {
        const QuadHot h = hot[i];
        if (h.key && h.count) {
            ++used;
            if (h.key >> 63) {                   // a k-mer of the direct path: key = 1 << 63 | table index
                sink_add(table, (uint64_t)(h.key & 0x7FFFFFFFFFFFFFFFull), (unsigned long long)h.count);
                continue;
            }
            const uint32_t r = (uint32_t)(h.key >> 32), it = (uint32_t)h.key;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if ((it >> (3 - q)) & 1u) sink_add(table, quad_kmer<K>(r, it, q), (unsigned long long)h.count);
        }
    }
    if (used) atomicAdd(error + 1, used);   // statistics only (KPAL_QUAD_VERBOSE)
    if constexpr (kLists) {
        if (sink_arg.list) {                // list mode: how many entries of the segment are valid
            __syncthreads();
            if (threadIdx.x == 0) sink_arg.count[blockIdx.x] = min(direct_n, sink_arg.cap);
        }
    }
}

// ==========================================================================================
// Two-level path, k = 13..16.  The (K+3)-mer of an item has up to 38 bits; its shared field x[2K-1:6] is cut into
//   coarse (2K-22 bits: 16 / 64 / 256 / 1024 buckets) | fine (9 bits) | the upper 7 bits of low13,
// so that after the coarse bucket is fixed an item is again  hi6 | fine9 | low13 | mask4 = 32 bits, and after the
// fine bucket is fixed it is the 23-bit item of the one-level path with L = 13: level 2 and the histogram are the
// k = 11 machinery (512 rows of 64 slots, four forms of 8192 bins), the count table index only gains the coarse
// field:  hi << (B1+9+s) | coarse << (9+s) | fine << s | lo.
//   Q1  quad_scatter_kernel<K>   ASCII -> level-1 records pool1[coarse row][workgroup][round]  (the kernel above)
//   Q3  quad2_scatter_kernel<K>  level-1 records of one coarse bucket -> pool2[coarse][fine row][workgroup][round]
//   Q2  quad_hist_kernel<K>      grid (512, coarse)
// 1 B/base read, then 4 x ~1.4 B per k-mer (two record pools written and read once each) instead of the 4-byte
// residuals + 2-byte keys + two counting passes of the round-1 two-level pipeline.
// ==========================================================================================
// Q3.  Workgroup (g2, c) takes the units [g2 * upw, (g2+1) * upw) of coarse bucket c; unit u = (replica u / G1,
// level-1 workgroup u % G1) is the run of nrounds1[u % G1] records that workgroup wrote for that row.  The units
// are addressed as one stream of `unit_cap` bytes each (records never written read as null items): a wave-step is
// 1 KiB of it, four items per lane -- the shape of the ASCII path, so the tile loop is the same.
// KM: the instantiation whose direct path aggregates k-mers in the hot-item table (with ageing): launched when the sample of the feed
// showed hot rows, like the REPEAT instantiation of level 1.
template <int K, int WAVES, int STEPS, bool KM = false>
__global__ __launch_bounds__(WAVES * 64) void quad2_scatter_kernel(const uint32_t *__restrict__ pool1, const uint32_t *__restrict__ nrounds1,
                                                            uint32_t G1, uint32_t rounds_cap1, uint32_t upw, uint32_t tiles_per_block,
                                                            uint32_t *__restrict__ pool2, uint32_t rounds_cap2,
                                                            uint32_t *__restrict__ nrounds2, uint32_t *__restrict__ error,
                                                            TableSink sink_arg)
{
    using C1 = QuadCfg<K>;
    using C = QuadCfg<11>;                      // rows of level 2: 512 x 64 slots, items of 23 bits
    constexpr int S = C::kSlots, NB = C::kBuckets, S1 = C1::kSlots;
    constexpr int THREADS = WAVES * 64;
    static_assert(WAVES == 8 || WAVES == 16, "tile shape");
    constexpr int CARRY = kQuadSpillCap / THREADS;
    constexpr uint32_t CAP = CARRY * THREADS;
    __shared__ __attribute__((aligned(16))) uint32_t rows[kQuadRowWords + kQuadDummyWords];
    __shared__ __attribute__((aligned(16))) uint32_t pos[NB];   // BYTES in use per row
    __shared__ QuadSpill spill[kQuadSpillCap];
    __shared__ uint32_t spill_cnt[2];
    __shared__ QuadHot hot[kQuadHotEntries];
    __shared__ uint32_t nr1[256];               // rounds written by every level-1 workgroup
    __shared__ uint32_t direct_n;               // list mode (TableSink): entries this workgroup appended to its segment
    __shared__ TableSink sink_lds;
    const uint32_t wg_linear = blockIdx.y * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0) {
        TableSink t = sink_arg;
        if (t.list) {
            t.list += (size_t)wg_linear * t.cap;
            t.count = &direct_n;
        }
        sink_lds = t;
        direct_n = 0;
    }
    const TableSinkRef table = {&sink_lds};
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const uint32_t coarse = blockIdx.y;
    for (int i = threadIdx.x; i < kQuadRowWords + kQuadDummyWords; i += THREADS) rows[i] = 0;
    for (int i = threadIdx.x; i < NB; i += THREADS) pos[i] = 0;
    for (int i = threadIdx.x; i < kQuadHotEntries; i += THREADS) hot[i] = QuadHot{0ull, 0u, 0u};
    for (uint32_t i = threadIdx.x; i < 256u; i += THREADS) nr1[i] = i < G1 ? nrounds1[i] : 0u;
    if (threadIdx.x == 0) {
        spill_cnt[0] = 0;
        spill_cnt[1] = 0;
    }
    __syncthreads();
    const uint32_t units = (uint32_t)C1::kRep * G1;                      // units of this coarse bucket
    const uint32_t u0 = blockIdx.x * upw, u1 = min(u0 + upw, units);
    const uint32_t unit_cap = rounds_cap1 * (uint32_t)(S1 * 4);          // bytes; a multiple of 1024 (the host rounds rounds_cap1)
    const uint32_t stream = (u1 > u0 ? u1 - u0 : 0u) * unit_cap;        // < 2^32: the host sizes the units so
    // The wave-step at stream position p (a multiple of 1024) lies in ONE unit: its unit (level-1 workgroup g1, replica
    // rl) and offset are wave-uniform and tracked incrementally -- one division per wave and tile, scalar adds per step --
    // and a lane only compares its 16 bytes against the number of records that workgroup wrote (the rest of the unit was
    // never written: null items).
    const uint32_t lane16 = (uint32_t)lane * 16u;
    uint64_t nx_p = 0;                          // next position to request
    uint32_t nx_g1 = 0, nx_rl = 0, nx_off = 0;
    auto seek = [&](uint64_t j) {               // to the first step of this wave in tile j
        nx_p = ((j * WAVES + (uint64_t)wave) * STEPS) * 1024ull;
        if (nx_p < (uint64_t)stream) {
            const uint32_t ul = (uint32_t)nx_p / unit_cap;
            nx_off = (uint32_t)nx_p - ul * unit_cap;
            const uint32_t u = u0 + ul;
            nx_rl = u / G1;
            nx_g1 = u - nx_rl * G1;
        }
    };
    auto fetch_next = [&]() -> uint4 {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (nx_p < (uint64_t)stream) {          // wave-uniform
            const uint32_t written = nr1[nx_g1] * (uint32_t)(S1 * 4);
            if (nx_off + lane16 < written) {
                const char *src = reinterpret_cast<const char *>(pool1) +
                                  ((uint64_t)((coarse * C1::kRep + nx_rl) * G1 + nx_g1) * rounds_cap1) * (uint64_t)(S1 * 4) + nx_off;
                v = *reinterpret_cast<const uint4 *>(src + lane16);
            }
            nx_p += 1024;
            nx_off += 1024;
            if (nx_off >= unit_cap) {
                nx_off = 0;
                if (++nx_g1 == G1) {
                    nx_g1 = 0;
                    ++nx_rl;
                }
            }
        }
        return v;
    };
    uint4 raw[STEPS];
    seek(0);
#pragma unroll
    for (int st = 0; st < STEPS; ++st) raw[st] = fetch_next();
    uint32_t round = 0;
    uint32_t carry_row[CARRY], carry_item[CARRY];
#pragma unroll
    for (int c = 0; c < CARRY; ++c) carry_row[c] = carry_item[c] = 0;
    constexpr int LPR = S / 4;
    constexpr int NVEC = NB * LPR;
    constexpr int FI = NVEC / THREADS;          // 16
    constexpr int DEFER = FI / STEPS;           // records a step of the next tile stores
    static_assert(S == 64 && LPR == 16, "packed records of 64 items");
    QuadPacked rec[FI];                         // (packed: 48 instead of 64 registers)
    bool have_rec = false;
    const uint32_t wg = coarse * gridDim.x + blockIdx.x;
    // (see quad_scatter_kernel: per-thread 32-bit offset + scalar base.)  pool2[coarse][row / 2][workgroup][round][row % 2][48 dwords]:
    // thread t holds vector t % 16 of row t / 16 (+ 64 i), i.e. 12 bytes of the 384-byte piece of row pair t / 32 (+ 32 i)
    constexpr uint32_t PIECE = 2u * (uint32_t)kQuadPackedRecordBytes;
    const uint64_t row_bytes = (uint64_t)gridDim.x * rounds_cap2 * (uint64_t)PIECE;      // per row PAIR
    const uint32_t thread_off = (uint32_t)((uint64_t)(threadIdx.x / (2 * LPR)) * row_bytes) + (threadIdx.x % (2 * LPR)) * 12u;
    auto store_rec = [&](int i) {
        uint32_t o = thread_off;
        asm volatile("" : "+v"(o));
        const uint64_t sc = ((uint64_t)coarse * (NB / 2) + (uint64_t)(i * (THREADS / (2 * LPR)))) * row_bytes + ((uint64_t)blockIdx.x * rounds_cap2 + (round - 1u)) * (uint64_t)PIECE;
        *reinterpret_cast<QuadPacked *>(reinterpret_cast<char *>(pool2) + sc + o) = rec[i];
    };
    auto store_rec_at = [&](int i, const QuadPacked &v) {   // record vector i of round `round`, i not a compile-time constant
        const uint64_t sc = ((uint64_t)coarse * (NB / 2) + (uint64_t)i * (uint64_t)(THREADS / (2 * LPR))) * row_bytes + ((uint64_t)blockIdx.x * rounds_cap2 + (round - 1u)) * (uint64_t)PIECE;
        *reinterpret_cast<QuadPacked *>(reinterpret_cast<char *>(pool2) + sc + thread_off) = v;
    };
    const uint64_t tile_bytes = (uint64_t)WAVES * STEPS * 1024;
    for (uint64_t j = 0; j < tiles_per_block && j * tile_bytes < (uint64_t)stream; ++j) {   // block-uniform
        const bool more = j + 1 < tiles_per_block && (j + 1) * tile_bytes < (uint64_t)stream;
        uint32_t *spill_n = &spill_cnt[j & 1];
        quad_place_carried<K, CARRY, 2, TableSinkRef, KM>(rows, pos, spill, spill_n, CAP, carry_row, carry_item, table, hot, coarse);
        if (more) seek(j + 1);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            quad_tile_priority<STEPS>(st);
            const uint4 v = raw[st];
            const uint32_t it[4] = {v.x, v.y, v.z, v.w};
            uint32_t row[4], item[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t t = ((it[q] >> 4) & 8191u) >> 9;
                row[q] = ((it[q] >> 17) & 511u) ^ C::smask(t);
                item[q] = ((it[q] >> 26) << 17) | (it[q] & 0x1FFFFu);   // the fine bucket leaves the item
            }
            asm volatile("" ::"v"(row[0]), "v"(item[0]));   // (the step's records are in use before the stores below are issued: see quad_scatter_kernel)
            if (have_rec) {
                // step st stores FI / STEPS records; the FI % STEPS left over (7- and 6-step tiles) went out with the flush itself
#pragma unroll
                for (int i = st * DEFER; i < (st + 1) * DEFER; ++i) store_rec(i);
            }
            if (more) raw[st] = fetch_next();
            quad_place<K, false, 2, 4, TableSinkRef, KM>(rows, pos, spill, spill_n, CAP, row, item, table, hot, coarse);
        }
        have_rec = false;
        quad_tile_priority<1>(0);
        lds_barrier();
        const uint32_t spilled = min(*spill_n, CAP);
        quad_note_spill(error, *spill_n, CAP);
        if (threadIdx.x == 0) spill_cnt[(j & 1) ^ 1] = 0;
        if constexpr (KM) {   // (ageing of the k-mer entries: see quad_scatter_kernel)
            if ((j & 3u) == 3u && threadIdx.x < (uint32_t)kQuadHotEntries) quad_hot_age<TableSinkRef>(table, hot);
        }
        quad_take_carried<CARRY, THREADS>(spill, spilled, carry_row, carry_item);
        {
            const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
            uint4 *rv = reinterpret_cast<uint4 *>(rows);
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                const uint4 v = rv[threadIdx.x + (uint32_t)i * THREADS];
                rec[i] = quad_pack3(v);
                rv[threadIdx.x + (uint32_t)i * THREADS] = zero4;
            }
            for (int i = threadIdx.x * 4; i < NB; i += THREADS * 4) *reinterpret_cast<uint4 *>(&pos[i]) = zero4;
        }
        if (round >= rounds_cap2) {
            if (threadIdx.x == 0) *error = 3u;
        } else {
            have_rec = true;
            ++round;
            // the records that the next tile's steps do not take (uneven split: steps that store three records instead of two cost
            // the 7- and 6-step variants 18 spilled registers) leave now
#pragma unroll
            for (int i = STEPS * DEFER; i < FI; ++i) store_rec(i);
        }
        lds_barrier();
    }
    if (have_rec) {
#pragma unroll
        for (int i = 0; i < STEPS * DEFER; ++i) store_rec(i);
    }
    {
        // ---- tail round: see quad_scatter_kernel
        if (round > 0u && round < rounds_cap2) {   // block-uniform
#pragma unroll
            for (int c = 0; c < CARRY; ++c)
                if (carry_item[c]) {
                    const uint32_t off = atomicAdd(&pos[carry_row[c]], 4u);
                    if (off < (uint32_t)S * 4u) {
                        *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(rows) + carry_row[c] * (uint32_t)(S * 4) + off) = carry_item[c];
                        carry_item[c] = 0;
                    }
                }
            lds_barrier();
            const uint4 *rv = reinterpret_cast<const uint4 *>(rows);
            ++round;
#pragma unroll 1
            for (int h = 0; h < FI; h += 4) {    // (a few records at a time: nothing here may cost the main loop registers)
#pragma unroll
                for (int i = 0; i < 4; ++i) rec[i] = quad_pack3(rv[threadIdx.x + (uint32_t)(h + i) * THREADS]);
#pragma unroll
                for (int i = 0; i < 4; ++i) store_rec_at(h + i, rec[i]);
            }
        }
    }
    if (threadIdx.x == 0) nrounds2[wg] = min(round, rounds_cap2);
#pragma unroll
    for (int c = 0; c < CARRY; ++c) quad_items_direct<K, 2, TableSinkRef, KM>(carry_item[c] != 0u, carry_row[c], carry_item[c], table, hot, coarse);
    __syncthreads();
    for (int i = threadIdx.x; i < kQuadHotEntries; i += THREADS) {
        const QuadHot h = hot[i];
        if (h.key && h.count) {
            if (h.key >> 63) {                   // a k-mer of the direct path: key = 1 << 63 | table index
                sink_add(table, (uint64_t)(h.key & 0x7FFFFFFFFFFFFFFFull), (unsigned long long)h.count);
                continue;
            }
            const uint32_t r = (uint32_t)(h.key >> 32), itm = (uint32_t)h.key;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if ((itm >> (3 - q)) & 1u) sink_add(table, quad_kmer<K, 2>(r, itm, q, coarse), (unsigned long long)h.count);
        }
    }
    if (sink_arg.list) {
        __syncthreads();
        if (threadIdx.x == 0) sink_arg.count[wg_linear] = min(direct_n, sink_arg.cap);
    }
}

// Table entry of bin `local` of form i (the k-mer at position i of an item) in the histogram of scrambled row `row`
// (of scrambled coarse bucket `coarse` on the two-level path): the bins of a form are table entries
// hi << (B + s) | bucket << s | lo with s = L - 6 + 2i; the bucket is unscrambled with the top bits of lo.
template <int K>
__device__ __forceinline__ uint64_t quad_bin_index(uint32_t row, uint32_t coarse, int i, uint32_t local)
{
    using C = QuadCfg<K>;
    const int sh = C::kLowBits - 6 + 2 * i;
    const uint32_t lopart = local & ((1u << sh) - 1u);
    const uint32_t hipart = local >> sh;
    const uint32_t t = C::kScrBits > 0 ? (lopart >> (sh - C::kScrBits)) : 0u;
    const uint32_t b = row ^ C::smask(t);
    if constexpr (C::kTwoLevel)
        return ((uint64_t)hipart << (C::kCoarseBits + 9 + sh)) | ((uint64_t)(coarse ^ C::smask1(t)) << (9 + sh)) | ((uint64_t)b << sh) | lopart;
    else
        return ((uint64_t)hipart << (C::kBucketBits + sh)) | ((uint64_t)b << sh) | lopart;
}

// Q2: histogram of one bucket's records, merged into the table.  hist[i * 2^L + local]: k-mer position i.
// Same hot-key guard as part_hist_kernel: per form the wave counts the occurrences of its first lane's bin
// with a ballot, those lanes add to private dummy words instead (64 adds to one LDS address serialise).
// PACKED (two-level path, chosen by the host for k = 15, 16 when a workgroup's record stream holds fewer than 2^16 item slots -- no
// bin can then reach 2^16): forms i and i + 2 share a word (low / high half), the histogram is 64 KiB instead of 128 and TWO
// workgroups fit a CU -- at k = 15 a workgroup streams only ~120 KB of records between zeroing its bins and staging them, and with
// one workgroup per CU nothing ran during those phases (and during the latency of the first loads).
template <int K, typename SINK = TableOnly, bool PACKED = false>
__global__ __launch_bounds__(1024) void quad_hist_kernel(const uint32_t *__restrict__ pool, const uint32_t *__restrict__ nrounds,
                                                         uint32_t G, uint32_t rounds_cap, SINK table,
                                                         uint32_t *__restrict__ stage)
{
    using C = QuadCfg<K>;
    static_assert(!PACKED || C::kTwoLevel, "packed bins are staged, not merged");
    // k = 13..16: blockIdx.y is the (scrambled) coarse bucket, blockIdx.x the fine row of level 2 (512 rows of 64 slots)
    constexpr int L = C::kLowBits, BINS = C::kFormBins, S = C::kTwoLevel ? 64 : C::kSlots;
    constexpr int PLANES = PACKED ? 2 : 4;
    __shared__ __attribute__((aligned(16))) uint32_t hist[PLANES * BINS + 64];
    auto plane_of = [](int i) -> uint32_t { return (uint32_t)(PACKED ? (i & 1) : i) * (uint32_t)BINS; };   // word offset of form i's plane
    auto shift_of = [](int i) -> int { return PACKED ? 16 * (i >> 1) : 0; };                                // ... and its half of the word
    // kPairRows: rows 2j and 2j+1 share every 128-byte line of their records.  Workgroups b and b + 8 are dispatched to
    // the same XCD (round-robin over eight) at nearly the same time: they take such a pair, so the second reader of a
    // line finds it in that XCD's L2 (or, drifting apart, in the memory-side cache).
    // (the two-level path's packed records: rows 2j and 2j+1 share the middle line of every 384-byte piece -- the same pairing)
    constexpr bool PACK3 = C::kTwoLevel;
    constexpr bool SIBLINGS = C::kPairRows || PACK3;
    const uint32_t row = SIBLINGS ? (((blockIdx.x >> 4) << 4) | ((blockIdx.x & 7u) << 1) | ((blockIdx.x >> 3) & 1u)) : blockIdx.x;
    const uint32_t coarse = blockIdx.y;
    const uint32_t row_linear = coarse * (C::kTwoLevel ? 512u : 0u) + row;
    nrounds += (size_t)coarse * G;
    for (int i = threadIdx.x; i < PLANES * BINS + 64; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Fast path: one ds_add per k-mer.  GUARD = true (skewed input only, chosen per 16-byte load when the wave's
    // first items repeat): per form the occurrences of the first active lane's bin are counted with a ballot and
    // those lanes add to private dummy words -- 64 adds to one LDS address serialise.
    // Two-level path: the planes are staged t-major (below), i.e. read from LDS with the hipart running fastest -- rows of 128
    // (form 0) or 512 (form 1) words apart, all in the same banks.  So the bins of those two forms are kept swizzled in LDS:
    // low bits of the hipart XOR-ed into bank bits that the staging read holds fixed (8 consecutive bins stay 8 consecutive,
    // aligned words).  The adds hit random bins either way.
    auto stage_swizzle = [](int i, uint32_t local) -> uint32_t {
#if defined(KPAL_AB_NO_STAGE_SWIZZLE)   // A/B timing
        return local;
#endif
        if constexpr (C::kStaged) {
            if (i == 0) return local ^ (((local >> 8) & 7u) << 3);                                   // hipart[3:1] -> bits 5:3 (t's low bits); a
                                                                                                      // staging lane reads hipart 2j and 2j+1
            if (i == 1) return local ^ (((local >> 9) & 7u) << 3);                                     // hipart[2:0] -> bits 5:3 (two lanes share a
                                                                                                      // hipart: bits 4:3 also tell the lane's half and h)
        }
        return local;
    };
    // One LDS add per k-mer -- and the histogram is bound by VALU issue as much as by the adds (75 % against 69 % busy,
    // profiles/r6/pmc_lds_quad.json; a three-operand or bit-field instruction costs 4.25 cycles of a SIMD, a plain shift right /
    // and / or / v_bitop3 2.5: tools/valu_bench.hip), so the fast path is spelled in the cheap ones (!PACKED: 32-bit bins):
    //   * the BYTE offset of bin `local` is (item >> (8 - 2i)) & ((2^L - 1) << 2) -- no v_bfe_u32, no shift left by two;
    //   * the staging swizzle of forms 0 and 1 works on that byte offset (the same bits, two places up);
    //   * a k-mer adds item & (8 >> i) -- its mask bit where it stands: 8, 4, 2 or 1 (or 0) -- instead of the extracted bit:
    //     plane i counts in UNITS of 8 >> i, and whoever reads a plane (staging, merge) shifts right by 3 - i.  (A bin holds at
    //     most 20 items per tile and scatter workgroup -- a row's capacity --, 2.6 M for a 16 GiB piece: x 8 stays far below 2^32.)
    uint32_t plane_base[2] = {2u * (uint32_t)(BINS * 4), 3u * (uint32_t)(BINS * 4)};   // byte offsets of planes 2 and 3, kept in registers
    asm volatile("" : "+v"(plane_base[0]), "+v"(plane_base[1]));
    constexpr bool kUnits = !PACKED;                    // plane i holds counts << unit_shift(i)
    auto unit_shift = [](int i) -> int { return kUnits ? 3 - i : 0; };
    auto add_item = [&](uint32_t it, auto guard_tag) {
        constexpr bool GUARD = decltype(guard_tag)::value;
        // bin of k-mer i in its form: the low 6-2i bits of hi6 above the top L-6+2i bits of low -- with the item laid out
        // hi6 | low | mask4 that is ONE bit-field of the item: L bits from bit 10 - 2i
#if defined(KPAL_AB_HIST_TWO_ADDS)   // A/B timing builds (wrong counts): what a histogram of (k + 1)-mers would issue -- two adds per item
        if constexpr (!GUARD && !PACKED) {   // (1: the bins as they are; 2: 2^(L + 2) u16 bins per form, two forms: the half-word select as well)
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                const uint32_t counted = __builtin_amdgcn_ubfe(it, 3 - i, 1) & __builtin_amdgcn_ubfe(it, 2 - i, 1);
                if (KPAL_AB_HIST_TWO_ADDS == 2) {
                    const uint32_t local = __builtin_amdgcn_ubfe(it, 8 - 2 * i, L + 2);                  // 15 bits at k = 12
                    atomicAdd(&hist[(uint32_t)(i / 2) * (uint32_t)(2 * BINS) + (local >> 1)], counted << (16u * (local & 1u)));
                } else {
                    atomicAdd(&hist[plane_of(i) + stage_swizzle(i, __builtin_amdgcn_ubfe(it, 10 - 2 * i, L))], counted << unit_shift(i));
                }
            }
            return;
        }
#endif
#if !defined(KPAL_AB_HIST_OLD_SPELLING)   // (A/B: the bit-field spelling of rounds 2-5 below)
        if constexpr (!GUARD && !PACKED) {
            char *base = reinterpret_cast<char *>(hist);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t off = (it >> (8 - 2 * i)) & (((1u << L) - 1u) << 2);          // 4 * local
#if !defined(KPAL_AB_NO_STAGE_SWIZZLE)
                if constexpr (C::kStaged) {                                             // (stage_swizzle, two places up)
                    if (i == 0) off ^= (off >> 5) & 0xE0u;
                    if (i == 1) off ^= (off >> 6) & 0xE0u;
                }
#endif
                if (i >= 2 && BINS * 4 * 2 > 65535) {
                    // (planes 2 and 3 lie beyond the 16-bit offset field of a DS instruction: their base is OR-ed in -- by the same
                    // v_bitop3_b32 that masks, (x & mask) | plane, where the compiler writes a v_and_b32 and a v_or_b32)
                    const uint32_t shifted = it >> (8 - 2 * i);
                    uint32_t addr;
                    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xea" : "=v"(addr) : "v"(shifted), "s"(((1u << L) - 1u) << 2), "v"(plane_base[i - 2]));
                    atomicAdd(reinterpret_cast<uint32_t *>(base + addr), it & (8u >> i));
                } else {
                    atomicAdd(reinterpret_cast<uint32_t *>(base + ((uint32_t)i * (uint32_t)(BINS * 4) | off)), it & (8u >> i));
                }
            }
            return;
        }
#endif
#if !defined(KPAL_AB_HIST_OLD_SPELLING)
        if constexpr (!GUARD && PACKED) {                  // (16-bit halves: no room for units; the byte offsets as above)
            char *base = reinterpret_cast<char *>(hist);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t off = (it >> (8 - 2 * i)) & (((1u << L) - 1u) << 2);
#if !defined(KPAL_AB_NO_STAGE_SWIZZLE)
                if (i == 0) off ^= (off >> 5) & 0xE0u;
                if (i == 1) off ^= (off >> 6) & 0xE0u;
#endif
                // forms 0, 1: the mask bit as 0 / 1 in the low half; forms 2, 3: moved to bit 16 (the high half of the word of forms 0, 1)
                const uint32_t value = i < 2 ? __builtin_amdgcn_ubfe(it, 3 - i, 1) : ((it << (13 + i)) & 0x10000u);
                atomicAdd(reinterpret_cast<uint32_t *>(base + ((uint32_t)(i & 1) * (uint32_t)(BINS * 4) | off)), value);
            }
            return;
        }
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t counted = __builtin_amdgcn_ubfe(it, 3 - i, 1);
            const uint32_t local = stage_swizzle(i, __builtin_amdgcn_ubfe(it, 10 - 2 * i, L));
            if constexpr (GUARD) {
                const uint32_t hot = __builtin_amdgcn_readfirstlane(local);
                const bool eq = counted && local == hot;
                const uint32_t same = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(eq));
                atomicAdd(&hist[eq ? (uint32_t)(PLANES * BINS + lane) : plane_of(i) + local], counted << (shift_of(i) + unit_shift(i)));
                if (same && lane == (__ffsll((long long)__builtin_amdgcn_ballot_w64(true)) - 1)) atomicAdd(&hist[plane_of(i) + hot], same << (shift_of(i) + unit_shift(i)));
            } else {
                atomicAdd(&hist[plane_of(i) + local], counted << (shift_of(i) + unit_shift(i)));
            }
        }
    };
    // low-complexity sequence repeats whole items across the lanes of a load: tested on the FIRST of the four vectors a lane
    // counts per iteration (runs of a repeated item span many records; per vector the test cost the k = 12 histogram 3 %)
    auto repeats = [&](const uint4 q) -> bool {
#if defined(KPAL_AB_HIST_NO_SKEWCHECK)   // A/B timing: what the test for repeated items costs
        return false;
#endif
        const uint32_t it = C::kItem3 ? (q.x & 0xFFFFFFu) : q.x;
        const uint32_t first = __builtin_amdgcn_readfirstlane(it);
        return first != 0u && __popcll(__builtin_amdgcn_ballot_w64(it == first)) >= 8;   // wave-uniform
    };
    auto add4 = [&](const uint4 q, const bool skew) {
        if constexpr (C::kItem3) {   // four 3-byte items + a fifth in the top bytes
            // (the bit-fields add_item reads lie below bit 23: the rider's byte above an item does not matter; an item is
            // null iff its mask nibble is zero)
            const uint32_t i0 = q.x, i1 = q.y, i2 = q.z, i3 = q.w;
            const uint32_t i4 = __builtin_amdgcn_perm(q.z, __builtin_amdgcn_perm(q.y, q.x, 0x0c0c0703u), 0x0c070100u);   // the top bytes of x, y, z
            if (__builtin_expect(skew, 0)) {
                if (i0 & 15u) add_item(i0, std::true_type{});
                if (i1 & 15u) add_item(i1, std::true_type{});
                if (i2 & 15u) add_item(i2, std::true_type{});
                if (i3 & 15u) add_item(i3, std::true_type{});
                if (i4 & 15u) add_item(i4, std::true_type{});
            } else {
                if (i0 & 15u) add_item(i0, std::false_type{});
                if (i1 & 15u) add_item(i1, std::false_type{});
                if (i2 & 15u) add_item(i2, std::false_type{});
                if (i3 & 15u) add_item(i3, std::false_type{});
#if !defined(KPAL_AB_HIST_NO_RIDER)   // A/B timing builds (wrong counts): what the fifth item group costs
                if (i4 & 15u) add_item(i4, std::false_type{});
#endif
            }
            return;
        }
        if (__builtin_expect(skew, 0)) {
            if (q.x) add_item(q.x, std::true_type{});
            if (q.y) add_item(q.y, std::true_type{});
            if (q.z) add_item(q.z, std::true_type{});
            if (q.w) add_item(q.w, std::true_type{});
        } else {
            if (q.x) add_item(q.x, std::false_type{});
            if (q.y) add_item(q.y, std::false_type{});
            if (q.z) add_item(q.z, std::false_type{});
            if (q.w) add_item(q.w, std::false_type{});
        }
    };
    // run g of the bucket = the records workgroup g of the scatter wrote for it.  With at least 16 runs a wave takes
    // whole runs; with fewer (the two-level path at k = 15, 16: 4 runs, 1 run) 16 / G waves share a run, 256 vectors
    // at a time each -- otherwise only G of the 16 waves would have loads in flight.
    const uint32_t parts = G >= 16u ? 1u : 16u / G;            // waves per run
    const uint32_t g_first = G >= 16u ? (uint32_t)wave : (uint32_t)wave % G;
    const uint32_t part = G >= 16u ? 0u : (uint32_t)wave / G;
    for (uint32_t g = g_first; g < G && part < parts; g += 16) {   // wave-uniform
        // PACK3: the run of (row, g) is one 192-byte record (16 vectors of 12 bytes) in every 384-byte piece of its row pair
        const uint32_t *src3 = pool + ((uint64_t)((row_linear >> 1) * G + g) * rounds_cap) * (uint64_t)(2 * kQuadPackedRecordBytes / 4) +
                               (row_linear & 1u) * (uint32_t)(kQuadPackedRecordBytes / 4);
        const uint4 *src = C::kPairRows
                               ? reinterpret_cast<const uint4 *>(pool + ((uint64_t)((row_linear >> 1) * G + g) * rounds_cap) * (2 * S) + (row_linear & 1u) * S)
                               : reinterpret_cast<const uint4 *>(pool + ((uint64_t)(row_linear * G + g) * rounds_cap) * S);
#if !defined(KPAL_QUAD_NO_PRIO)   // A/B builds
        // The wave that is behind in its runs goes first (quad_tile_priority: the arbiter's oldest-first order let the waves of a SIMD
        // finish one after the other, and the LDS idled under the last ones): k = 12 3.79 -> 3.58 ms, k = 11 3.35 -> 3.19, k = 13
        // 3.85 -> 3.63.  Not when waves SHARE a run (G < 16; k = 15, 16): g then says nothing about progress, and the same switch
        // cost that histogram 8 %.
        if (parts == 1u) {
            switch (g * 4u / G) {
            case 0: __builtin_amdgcn_s_setprio(3); break;
            case 1: __builtin_amdgcn_s_setprio(2); break;
            case 2: __builtin_amdgcn_s_setprio(1); break;
            default: __builtin_amdgcn_s_setprio(0); break;
            }
        }
#endif
#if defined(KPAL_AB_POOL_FILL)   // (... and read: six sevenths of the vectors, i.e. fewer null items to test AND fewer real ones: an upper bound of the gain)
        const uint32_t nvec = nrounds[g] * (uint32_t)(S / 4) / 7u * 6u;
#else
        const uint32_t nvec = nrounds[g] * (uint32_t)(S / 4);
#endif
        if (nvec == 0u) continue;                  // wave-uniform: this scatter workgroup wrote nothing
        // four 16-byte loads per lane in flight; the next four are requested before these are counted.  The loads are
        // UNCONDITIONAL (index clamped, value zeroed afterwards): predicated ones sit in basic blocks of their own, the
        // compiler then waits with vmcnt(0) before the first use -- for the four just requested as well, so every
        // iteration paid a full memory latency and the LDS idled 43 % of the time.
        auto fetch = [&](uint32_t at) -> uint4 {
            const uint32_t a = min(at, nvec - 1u);
            uint4 r;
            if constexpr (PACK3) {
                const uint32_t *p3 = src3 + (a >> 4) * (uint32_t)(2 * kQuadPackedRecordBytes / 4) + (a & 15u) * 3u;
                const QuadPacked w = *reinterpret_cast<const QuadPacked *>(p3);
                r = quad_unpack3(w.a, w.b, w.c);
            } else {
                r = src[C::kPairRows ? ((a >> 2) * 8u + (a & 3u)) : a];   // (pairs: the row's half of every 128-byte line)
            }
            const uint32_t keep = at < nvec ? 0xFFFFFFFFu : 0u;
            r.x &= keep;
            r.y &= keep;
            r.z &= keep;
            r.w &= keep;
            return r;
        };
        uint32_t v = part * 256u + lane;
        uint4 q0 = fetch(v), q1 = fetch(v + 64u), q2 = fetch(v + 128u), q3 = fetch(v + 192u);
        while (v < nvec) {
            v += parts * 256u;
            const uint4 n0 = fetch(v), n1 = fetch(v + 64u), n2 = fetch(v + 128u), n3 = fetch(v + 192u);
#if defined(KPAL_AB_HIST_NO_ADD)     // A/B timing builds (wrong counts): the record stream alone
            asm volatile("" ::"v"(q0.x ^ q0.y ^ q0.z ^ q0.w ^ q1.x ^ q1.y ^ q1.z ^ q1.w ^ q2.x ^ q2.y ^ q2.z ^ q2.w ^ q3.x ^ q3.y ^ q3.z ^ q3.w));
#else
            const bool skew = repeats(q0);
            add4(q0, skew);
            add4(q1, skew);
            add4(q2, skew);
            add4(q3, skew);
#endif
            q0 = n0;
            q1 = n1;
            q2 = n2;
            q3 = n3;
        }
    }
#if !defined(KPAL_QUAD_NO_PRIO)
    __builtin_amdgcn_s_setprio(3);               // (staging / merging: every wave is needed)
#endif
    __syncthreads();
    if (stage) {
        // two-level path: every table entry would receive four atomic adds (one per form, from four different
        // workgroups: 4 x 4^k atomics, 28 ms at k = 15).  The forms are staged instead, as 8-bit counts (4.3 GB at k = 15; as
        // 16-bit counts this kernel wrote and the finalisation read twice that), and quad2_finalize_kernel gathers the four of
        // every entry.  Layout (quad2_index.hpp): the workgroup's four planes as one contiguous 32 KiB block, each plane
        // t-major -- the bins of one t (the top four bits of the low part, which select the scramble mask) are the bins of one
        // true bucket, so the finalisation reads them as one piece of 512 B.  A count that does not fit (a k-mer seen 256 times
        // in one batch within ONE of its four positions) goes to the table / the FRESH list directly and is staged as zero.
        if constexpr (C::kStaged) {
            // (k = 12, one level: `row` is the scrambled 11-bit bucket = scrambled coarse * 512 + scrambled fine of quad2_index.hpp, `coarse` 0)
            using Q = Quad2Index<K>;
            quad2_stage_t *dst = reinterpret_cast<quad2_stage_t *>(stage);
            for (int j = threadIdx.x; j < 4 * BINS / 16; j += blockDim.x) {
                const int i = j / (BINS / 16);                             // plane (form)
                const uint32_t o = ((uint32_t)j % (uint32_t)(BINS / 16)) * 16u;   // first of sixteen words of the plane
                uint32_t c[16];
                uint32_t any = 0;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t local = Q::bin_of_word(i, o + 8u * (uint32_t)h);   // eight consecutive words are eight consecutive bins
                    const uint32_t phys = stage_swizzle(i, local) & ~7u;   // (aligned blocks of eight stay together; the mask only tells the
                                                                            // compiler so: without it the two reads become four ds_read2_b32)
                    const uint4 *src = reinterpret_cast<const uint4 *>(__builtin_assume_aligned(&hist[plane_of(i) + phys], 16));
                    const uint4 a = src[0];
                    const uint4 b = src[1];
                    c[8 * h + 0] = a.x, c[8 * h + 1] = a.y, c[8 * h + 2] = a.z, c[8 * h + 3] = a.w;
                    c[8 * h + 4] = b.x, c[8 * h + 5] = b.y, c[8 * h + 6] = b.z, c[8 * h + 7] = b.w;
                }
                if constexpr (PACKED) {
                    const int sh = shift_of(i);                        // (wave-uniform: a plane is 512 vectors)
#pragma unroll
                    for (int e = 0; e < 16; ++e) c[e] = (c[e] >> sh) & 0xFFFFu;
                } else {
                    const int sh = unit_shift(i);                      // (the plane's unit; wave-uniform)
#pragma unroll
                    for (int e = 0; e < 16; ++e) c[e] >>= sh;
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) any |= c[e];
                if (__builtin_expect(any >= kQuad2StageLimit, 0)) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (c[e] >= kQuad2StageLimit) {
                            sink_add(table, quad_bin_index<K>(row, coarse, i, Q::bin_of_word(i, o + (uint32_t)(e & 8)) + (uint32_t)(e & 7)), (unsigned long long)c[e]);
                            c[e] = 0;
                        }
                }
                auto pack = [&](int w) { return c[4 * w] | (c[4 * w + 1] << 8) | (c[4 * w + 2] << 16) | (c[4 * w + 3] << 24); };
                *reinterpret_cast<uint4 *>(dst + Q::word_pos(i, coarse, row, o)) = make_uint4(pack(0), pack(1), pack(2), pack(3));
            }
        }
        return;
    }
#if defined(KPAL_AB_HIST_NO_MERGE)   // A/B timing builds (wrong counts): what the merge into the table costs
    return;
#endif
    if constexpr (!PACKED) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            for (int local = threadIdx.x; local < BINS; local += blockDim.x) {
                const uint32_t c = hist[i * BINS + stage_swizzle(i, (uint32_t)local)] >> unit_shift(i);   // (k = 12 keeps forms 0 and 1 swizzled for the staging read)
                if (c) atomicAdd(&table.table[quad_bin_index<K>(row, coarse, i, (uint32_t)local)], (unsigned long long)c);
            }
        }
    }
}

// Q4 (k = 13..16, and the one-level pipeline at k = 12): the finalisation.  table[idx] += the four staged form counts of entry idx -- and, BALANCE, Profile.balance
// (klib.py:285-298) in the same pass: table[idx] = v[idx] + v[rc(idx)], v = table + forms.  Balancing afterwards costs a
// second read and write of the whole table (k = 15: 8 GiB each way, 5.8 of 31 ms); here the table is read once and written
// once either way.
//   Workgroup R owns set R of quad2_index.hpp: 2^14 entries = every value of the low 7 bits x every value of their
// reverse-complement image (the top 6 bits and bit 2K-8).  rc maps set R onto set R' = rc(R), so the pair is closed:
// the workgroup of the smaller base takes both, the other returns at once (a set that is its own partner -- even k
// only -- is handled alone).  The ten sources (table + four forms, of R and of R') are read as THEY lie -- 16 bytes per
// lane, consecutive lanes consecutive addresses, pieces of 256 B .. 16 KiB (stream_entry) -- and every value is added
// into ONE LDS array indexed by the position in set R: a value of R' goes to the position of its reverse complement.
// The array then holds out[p] = v[p] + v'[rc(p)] for every p of R, and out is symmetric: row-wise it is written to R's
// table entries, column-wise to those of R' (rows padded to 129: both directions free of bank conflicts).  64-bit
// LDS adds: exact for any counts.
//   FRESH: the table holds nothing yet (first piece of a count; it was neither zeroed nor does it hold direct adds -- those wait in
// the TableSink lists): it is not read, only written -- 8.6 GB less to move at k = 15, on top of the 8.6 GB the skipped memset saves.
template <int K, bool BALANCE, bool FRESH>
__global__ __launch_bounds__(1024) void quad2_finalize_kernel(const quad2_stage_t *__restrict__ stage, unsigned long long *__restrict__ table)
{
    using Q = Quad2Index<K>;
    constexpr int RS = Q::kRowStride;
    // FRESH: an entry is the sum of at most eight staged counts (four forms of the entry and of its reverse complement), so 32-bit
    // accumulators do: half the LDS -- TWO workgroups per CU, one streaming while the other zeroes, waits at its barrier or writes out
    using acc_t = typename std::conditional<FRESH, uint32_t, unsigned long long>::type;
    static_assert(!FRESH || 8u * (kQuad2StageLimit - 1u) < 0xFFFFFFFFu, "accumulator width");
    __shared__ acc_t acc[128 * RS];
    // Which set a workgroup takes.  Plain: blockIdx (neighbours in the dispatch order touch adjacent 1 KiB runs).  Balancing: a set's
    // 128 table runs lie 128 MiB apart and its partner's runs somewhere else entirely -- with R's low bits running fastest every
    // workgroup in flight has partner runs in pages of its own (the partner's page number is the reverse complement of R's LOW
    // bits), and the kernel ran at 2.9 TB/s against 4.9 TB/s for the plain form: address translation, not DRAM.  So the
    // bits of R that are neither page bits of its own runs (index bits >= 18: 2 MiB pages of 8-byte entries) nor of the
    // partner's (index bits <= 2K-19) run fastest: workgroups dispatched together share the pages on both sides.
    uint32_t set_id = blockIdx.x;
    if constexpr (BALANCE) {
        constexpr int SB = Q::kSetBits;
        constexpr int LO = 2 * K - 25 > 0 ? 2 * K - 25 : 0;          // R bit j is index bit 7 + j (j <= 2K-16)
        constexpr int HI = 10 < SB - 2 ? 10 : SB - 2;
        constexpr int NN = HI - LO + 1;
        if constexpr (NN > 0 && LO > 0) {
            const uint32_t b = blockIdx.x;
            set_id = ((b & ((1u << NN) - 1u)) << LO) | ((b >> NN) & ((1u << LO) - 1u)) | ((b >> (NN + LO)) << (NN + LO));
        }
    }
    const uint64_t base = Q::set_base(set_id);
    const uint64_t pbase = BALANCE ? Q::partner_base(base) : base;
    if (BALANCE && base > pbase) return;                        // block-uniform
    const bool self = BALANCE && base == pbase;
    const bool pair = BALANCE && base != pbase;
    for (int i = threadIdx.x; i < 128 * RS; i += 1024) acc[i] = 0;
    __syncthreads();
    // Entry (lo7, hi7) of set R lives at acc[hi7][lo7 ^ swizzle]: a lane that unpacks eight consecutive 16-bit counts adds to
    // eight consecutive entries while its neighbours add 8 entries further on -- unswizzled, the lanes of a 32-lane group
    // would share four 8-byte banks (8-way conflicts: measured 55-70 % of the LDS cycles at LDS 55 % busy); XOR-ing the low three
    // bits with the next three spreads them over all banks, and keeps even / odd neighbours adjacent.
    auto at = [&](uint32_t hi7, uint32_t lo7) -> uint32_t { return hi7 * RS + (lo7 ^ ((lo7 >> 3) & 7u)); };
    // a value of set R at (lo7, hi7) -> acc[hi7][lo7]; of the partner set (or, self-paired, once more) -> the transposed place
    auto add_own = [&](uint32_t lo7, uint32_t hi7, unsigned long long v) {
        if (v) atomicAdd(&acc[at(hi7, lo7)], (acc_t)v);
    };
    auto add_partner = [&](uint32_t lo7, uint32_t hi7, unsigned long long v) {
        if (v) atomicAdd(&acc[at(Q::partner_hi7(lo7), Q::partner_lo7(hi7))], (acc_t)v);
    };
    // ---- the table itself: 8192 vectors of two entries per set (four loads in flight per thread at a time: with all eight
    // live next to the partner arithmetic the balancing form needed more than the 128 registers a 1024-thread workgroup has)
#pragma unroll 1
    for (int half = 0; half < (FRESH ? 0 : (pair ? 2 : 1)); ++half) {
        const uint64_t b = half ? pbase : base;
#pragma unroll 1
        for (int batch = 0; batch < 2; ++batch) {
            ulonglong2 t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t q = 2u * (threadIdx.x + 1024u * (uint32_t)(4 * batch + j));
                uint32_t lo7, hi7;
                Q::stream_entry(4, q, lo7, hi7);
                t[j] = *reinterpret_cast<const ulonglong2 *>(table + Q::entry(b, lo7, hi7));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t q = 2u * (threadIdx.x + 1024u * (uint32_t)(4 * batch + j));
                uint32_t lo7, hi7;
                Q::stream_entry(4, q, lo7, hi7);
                if (half == 0) {
                    add_own(lo7, hi7, t[j].x);
                    add_own(lo7 + 1u, hi7, t[j].y);
                }
                if (half == 1 || self) {
                    add_partner(lo7, hi7, t[j].x);
                    add_partner(lo7 + 1u, hi7, t[j].y);
                }
            }
        }
    }
    // ---- the four forms: 1024 vectors of sixteen 8-bit counts per form and set -- one load per form and thread; a vector is two
    // groups of eight consecutive entries (stream positions q and q + 8)
    static_assert(sizeof(quad2_stage_t) == 1, "sixteen counts per 16-byte vector");
#pragma unroll 1
    for (int half = 0; half < (pair ? 2 : 1); ++half) {
        const uint64_t b = half ? pbase : base;
        const uint32_t q = 16u * threadIdx.x;
        uint4 f[4];
#pragma unroll
        for (int form = 0; form < 4; ++form) {
            uint32_t lo7, hi7;
            Q::stream_entry(form, q, lo7, hi7);
            f[form] = *reinterpret_cast<const uint4 *>(stage + Q::stage_pos(form, Q::entry(b, lo7, hi7)));
        }
#pragma unroll
        for (int form = 0; form < 4; ++form) {
            if ((f[form].x | f[form].y | f[form].z | f[form].w) == 0u) continue;
#pragma unroll 1
            for (int g = 0; g < 2; ++g) {        // (not unrolled: 64 adds with their partner arithmetic side by side cost the balancing form spilled registers)
                const uint32_t w[2] = {g ? f[form].z : f[form].x, g ? f[form].w : f[form].y};
                uint32_t lo7, hi7;
                Q::stream_entry(form, q + 8u * (uint32_t)g, lo7, hi7);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned long long v = (w[e >> 2] >> (8 * (e & 3))) & 0xFFu;
                    if (half == 0) add_own(lo7 + (uint32_t)e, hi7, v);
                    if (half == 1 || self) add_partner(lo7 + (uint32_t)e, hi7, v);
                }
            }
        }
    }
    __syncthreads();
    // ---- out: rows of R as they lie, rows of R' transposed
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const uint32_t v = threadIdx.x + 1024u * (uint32_t)it;
        const uint32_t hi7 = v >> 6, lo7 = (v & 63u) * 2u;
        ulonglong2 o;
        o.x = acc[at(hi7, lo7)];
        o.y = acc[at(hi7, lo7 + 1u)];
        *reinterpret_cast<ulonglong2 *>(table + Q::entry(base, lo7, hi7)) = o;
    }
    if (pair) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const uint32_t v = threadIdx.x + 1024u * (uint32_t)it;
            const uint32_t hi7 = v >> 6, lo7 = (v & 63u) * 2u;            // an entry pair of R'
            ulonglong2 o;
            o.x = acc[at(Q::partner_hi7(lo7), Q::partner_lo7(hi7))];
            o.y = acc[at(Q::partner_hi7(lo7 + 1u), Q::partner_lo7(hi7))];
            *reinterpret_cast<ulonglong2 *>(table + Q::entry(pbase, lo7, hi7)) = o;
        }
    }
}

// FRESH finalisation, second step: the counts that bypassed the records (TableSink lists) are added to the finished table --
// balanced: to the entry and to its reverse complement (a palindrome receives both adds: Profile.balance doubles it).
// Segment g < nseg holds count[g] entries of (index << 32) | count, `cap` apart; the LAST segment (the histogram stage's, shared by
// all its workgroups: `cap_last` entries) is worked off by kQuad2ListTailBlocks workgroups together.
constexpr uint32_t kQuad2ListTailBlocks = 256;
template <int K>
__global__ __launch_bounds__(256) void quad2_apply_list_kernel(const unsigned long long *__restrict__ list, const uint32_t *__restrict__ count,
                                                               uint32_t cap, uint32_t nseg, uint32_t cap_last, uint32_t balance,
                                                               unsigned long long *__restrict__ table)
{
    const bool last = blockIdx.x >= nseg;
    const uint32_t n = last ? min(count[nseg], cap_last) : min(count[blockIdx.x], cap);
    const unsigned long long *seg = list + (size_t)(last ? nseg : blockIdx.x) * cap;
    const uint32_t first = last ? (blockIdx.x - nseg) * blockDim.x + threadIdx.x : threadIdx.x;
    const uint32_t step = last ? kQuad2ListTailBlocks * blockDim.x : blockDim.x;
    for (uint32_t i = first; i < n; i += step) {
        const unsigned long long e = seg[i];
        const uint64_t idx = e >> 32;
        const unsigned long long c = e & 0xFFFFFFFFull;
        atomicAdd(&table[idx], c);
        if (balance) atomicAdd(&table[Quad2Index<K>::revcomp(idx)], c);
    }
}

}  // namespace kpal
