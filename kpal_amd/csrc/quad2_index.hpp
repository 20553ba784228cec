// quad2_index.hpp -- index arithmetic of the finalisation stage of the quad pipelines (two-level: k = 13..16; and the one-level
// pipeline at k = 12, whose 2048 histogram buckets are this file's (coarse: 2 bits | fine: 9 bits) read as one number):
// where the histogram workgroups stage their four forms, and which table entries one finalisation workgroup owns.
// Plain integer functions, shared by quad_kernels.hpp (device) and tests/quad2_index_check.cpp (host: the same
// functions drive a CPU emulation of the staging + finalisation that is compared with a direct count + balance).
//
// FORMS.  An item holds four overlapping k-mers; form i (0 = oldest) is the histogram of the k-mers at position i.
// Seen from a table entry idx (2K bits), form i cuts it with s = 7 + 2i as
//     hipart (13 - s bits) | coarse (CB = 2K - 22 bits) | fine (9 bits) | lopart (s bits),     t = top 4 bits of lopart,
// and the k-mer was histogrammed by the workgroup of the SCRAMBLED bucket (coarse ^ smask1(t), fine ^ smask(t)) in bin
// local = hipart << s | lopart (quad_bin_index).
//
// STAGING LAYOUT (8-bit counts, quad2_stage_t; a count >= kQuad2StageLimit bypasses the staging: TableSink).  A histogram workgroup
// stores its four planes of 8192 bins as ONE contiguous 32 KiB block (scattered pieces cost the histogram kernel 20 % at k = 15),
// each plane ordered t-major:
//        pos = ((scrambled coarse * 512 + scrambled fine) * 4 + i) * 8192 + t * 512 + hipart * 2^(s-4) + (lopart mod 2^(s-4)).
// The bins of one t are the bins of ONE true bucket, so a reader that lets the low bits of idx and its top bits run finds
// every form in pieces of 128 (form 3) to 512 counts: the scrambling (which exists to spread compositional skew over the
// scatter rows) only decides which block a piece lies in.  Sixteen aligned consecutive positions are two groups of eight
// consecutive table entries (stream_entry).
//
// SETS.  Finalisation workgroup R owns the 2^14 entries whose bits outside FREE equal R's, FREE = the low 7 bits
// (digits 0..2 and the low bit of digit 3) and their reverse-complement image (the top 6 bits and bit 2K-8).  The
// reverse complement maps set R onto set R' = rc(R): balancing (out[i] = v[i] + v[rc(i)]) needs the pair (R, R') and
// nothing else, the table is read and written in runs of 1 KiB, the forms in pieces of 128 .. 512 B.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define KPAL_HD __host__ __device__ __forceinline__
#else
#define KPAL_HD inline
#endif

namespace kpal {

typedef uint8_t quad2_stage_t;                      // one staged count (per table entry and form)
constexpr uint32_t kQuad2StageLimit = 256u;         // counts >= this go to the table / the FRESH list directly and are staged as zero

template <int K>
struct Quad2Index {
    static_assert(K >= 12 && K <= 16, "staged quad histograms: k = 12..16");
    static constexpr int CB = 2 * K - 22;                       // coarse bucket bits: 4, 6, 8, 10 (k = 12: 2, the top bits of its 11-bit bucket)
    static constexpr uint32_t kCoarseMask = (1u << CB) - 1u;
    static constexpr int kSetBits = 2 * K - 14;                 // sets: 2^12 .. 2^18
    static constexpr uint32_t kSets = 1u << kSetBits;
    static constexpr int kRowStride = 129;                      // LDS: 128 rows (hi7) x 129 (128 lo7 + 1 pad) u64

    // k = 12 scrambles its 11-bit bucket with (t << 7 | t) (QuadCfg<12>::smask): the low nine bits and the top two of that
    static constexpr KPAL_HD uint32_t smask(uint32_t t) { return K == 12 ? ((((t & 3u) << 7) | t) & 511u) : (((t << 5) | t) & 511u); }   // = QuadCfg<K>::smask
    static constexpr KPAL_HD uint32_t smask1(uint32_t t)                                                                     // = QuadCfg<K>::smask1
    {
        return K == 12 ? ((t >> 2) & 3u) : ((CB > 4 ? ((t << (CB - 4)) ^ t) : t) & kCoarseMask);
    }

    // ---- staging
    // position (in counts) of form i of table entry idx
    static KPAL_HD uint64_t stage_pos(int i, uint64_t idx)
    {
        const int s = 7 + 2 * i;
        const uint32_t lopart = (uint32_t)idx & ((1u << s) - 1u);
        const uint32_t fine = (uint32_t)(idx >> s) & 511u;
        const uint32_t coarse = (uint32_t)(idx >> (s + 9)) & kCoarseMask;
        const uint32_t hipart = (uint32_t)(idx >> (s + 9 + CB));
        const uint32_t t = lopart >> (s - 4), rest = lopart & ((1u << (s - 4)) - 1u);
        const uint32_t sc = coarse ^ smask1(t), sf = fine ^ smask(t);   // the block of the workgroup that histogrammed it
        return ((uint64_t)((sc * 512u + sf) * 4u + (uint32_t)i) << 13) + (t << 9) + (hipart << (s - 4)) + rest;
    }
    // the histogram side: word o (0 .. 8191, t-major) of plane i as written by the workgroup of scrambled bucket
    // (sc, sf): which of its LDS bins it is, and where it goes
    static KPAL_HD uint32_t bin_of_word(int i, uint32_t o)
    {
        const int s = 7 + 2 * i;
        const uint32_t t = o >> 9, rem = o & 511u;
        return ((rem >> (s - 4)) << s) | (t << (s - 4)) | (rem & ((1u << (s - 4)) - 1u));
    }
    static KPAL_HD uint64_t word_pos(int i, uint32_t sc, uint32_t sf, uint32_t o)
    {
        return ((uint64_t)((sc * 512u + sf) * 4u + (uint32_t)i) << 13) + o;
    }

    // ---- sets
    static constexpr uint64_t kFree = 0x7Full | (0x3Full << (2 * K - 6)) | (1ull << (2 * K - 8));
    static KPAL_HD uint64_t set_base(uint32_t R)               // R's bits dropped into the positions outside kFree
    {
        return ((uint64_t)(R & ((1u << (2 * K - 15)) - 1u)) << 7) | ((uint64_t)(R >> (2 * K - 15)) << (2 * K - 7));
    }
    static KPAL_HD uint64_t revcomp(uint64_t idx)               // reverse complement of a K-digit index (klib.py:394-412)
    {
        uint64_t x = ~idx;
        x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
        x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
        x = ((x >> 8) & 0x00FF00FF00FF00FFull) | ((x & 0x00FF00FF00FF00FFull) << 8);
        x = ((x >> 16) & 0x0000FFFF0000FFFFull) | ((x & 0x0000FFFF0000FFFFull) << 16);
        x = (x >> 32) | (x << 32);
        return x >> (64 - 2 * K);
    }
    static KPAL_HD uint64_t partner_base(uint64_t base) { return revcomp(base) & ~kFree; }
    // entry (lo7, hi7) of the set with base `base`; hi7 = top six bits << 1 | bit 2K-8
    static KPAL_HD uint64_t entry(uint64_t base, uint32_t lo7, uint32_t hi7)
    {
        return base | lo7 | ((uint64_t)(hi7 >> 1) << (2 * K - 6)) | ((uint64_t)(hi7 & 1u) << (2 * K - 8));
    }
    // reverse the three base-4 digits of a 6-bit value and complement them
    static KPAL_HD uint32_t rc3(uint32_t v)
    {
        v = ~v & 63u;
        return ((v & 3u) << 4) | (v & 12u) | (v >> 4);
    }
    // the partner rc(j) of entry j = (lo7, hi7) of one set is entry (partner_lo7(hi7), partner_hi7(lo7)) of the other
    static KPAL_HD uint32_t partner_lo7(uint32_t hi7) { return (((hi7 & 1u) ^ 1u) << 6) | rc3(hi7 >> 1); }
    static KPAL_HD uint32_t partner_hi7(uint32_t lo7) { return (rc3(lo7 & 63u) << 1) | ((lo7 >> 6) ^ 1u); }
    // Stream order of the entries of a set as source `src` (0..3: the forms, 4: the table itself) sees them: position
    // q (0 .. 16383) -> (lo7, hi7), such that consecutive q are consecutive addresses of that source as far as they go
    // (forms 0..2: 512 words -- one t of one true bucket --, form 3 / table: 128 entries).  q and lo7 agree in their low 3 bits.
    static KPAL_HD void stream_entry(int src, uint32_t q, uint32_t &lo7, uint32_t &hi7)
    {
        uint32_t top6;
        const uint32_t b = (q >> 13) & 1u;
        if (src == 0) {
            lo7 = (((q >> 9) & 15u) << 3) | (q & 7u);
            top6 = (q >> 3) & 63u;
        } else if (src == 1) {
            lo7 = (((q >> 9) & 3u) << 5) | (q & 31u);
            top6 = (((q >> 5) & 15u) << 2) | ((q >> 11) & 3u);
        } else if (src == 2) {
            lo7 = q & 127u;
            top6 = (((q >> 7) & 3u) << 4) | ((q >> 9) & 15u);
        } else {
            lo7 = q & 127u;
            top6 = (q >> 7) & 63u;
        }
        hi7 = (top6 << 1) | b;
    }
};

}  // namespace kpal
