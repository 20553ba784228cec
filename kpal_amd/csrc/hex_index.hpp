// hex_index.hpp -- index arithmetic of the HEX pipeline (k = 12): items of SIX overlapping k-mers in three bytes.
// Plain integer functions, shared by hex_kernels.hpp (device) and tests/native/hex_index_check.cpp (host: the same functions
// drive a CPU emulation of split -> items -> histogram forms -> table that is compared with a direct count).
//
// Why.  The quad pipeline (quad_kernels.hpp) writes and reads 0.75 bytes per k-mer at k = 12 (four k-mers in a 23-bit item
// with a 4-bit mask) and its scatter moves bytes at the rate the CUs' store paths take: fewer bytes per k-mer are
// what is left.  The mask is all-ones for nine items in ten, so it leaves the item:
//
//   GROUP  = the six k-mers ending at bytes 6g .. 6g+5 of the stream = the 17-mer x (34 bits, first base most significant)
//            ending at byte 6g+5; k-mer i (0 = oldest) is x[33-2i : 10-2i].  All six share x[23:10] (seven bases).
//   BUCKET = x[23:13] (11 bits, 2048 rows / histogram workgroups), XOR-scrambled with t = x[12:10], the three shared bits below it
//            (every form can recover t from its own bin index, see hex_bin_entry).
//   ITEM   = 24 bits.  A group whose six k-mers all count is ONE full item: 1 << 23 | x[33:24] << 13 | x[12:0].
//            A group with a read end / N inside (6.6 % of the groups of 150-base reads) becomes one or two HALF items of three
//            k-mers with a 3-bit mask: 0 << 23 | half << 22 | mask3 << 19 | payload17 -- upper half (k-mers 0..2, bases 0..13):
//            payload = x[33:24] << 7 | x[12:6]; lower half (k-mers 3..5, bases 3..16): payload = x[27:24] << 13 | x[12:0].
//            mask3 != 0, so 0 is the null item (record padding).  0.5 B per k-mer + ~10 % for the half items.
//   FORMS  = six histograms of 2^13 bins per bucket (k-mer i of every item): bin local_i = payload bits [22-2i : 10-2i] of
//            the item brought into full-item layout (hex_unpack); its table entry is hex_bin_entry().
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define KPAL_HEX_HD __host__ __device__ __forceinline__
#else
#define KPAL_HEX_HD inline
#endif

namespace kpal {

struct HexIndex {
    static constexpr int K = 12;
    static constexpr int kForms = 6;
    static constexpr int kBucketBits = 11;
    static constexpr int kRows = 1 << kBucketBits;              // scatter rows = histogram workgroups
    static constexpr int kLowBits = 13;                         // x[12:0]
    static constexpr int kFormBins = 1 << 13;
    static constexpr uint32_t kFull = 1u << 23;
    static constexpr uint64_t kXMask = (1ull << 34) - 1ull;

    // t in both ends of the bucket (as QuadCfg::smask does with four bits)
    static constexpr KPAL_HEX_HD uint32_t smask(uint32_t t) { return ((t << 8) | t) & 2047u; }

    static KPAL_HEX_HD uint32_t row_of(uint64_t x)
    {
        const uint32_t lo = (uint32_t)x;
        return ((lo >> 13) & 2047u) ^ smask((lo >> 10) & 7u);
    }
    static KPAL_HEX_HD uint32_t full_item(uint64_t x) { return kFull | ((uint32_t)(x >> 24) << 13) | ((uint32_t)x & 8191u); }
    // half = 0: k-mers 0..2, mask3 bit 2 = k-mer 0;  half = 1: k-mers 3..5, mask3 bit 2 = k-mer 3
    static KPAL_HEX_HD uint32_t half_item(uint64_t x, int half, uint32_t mask3)
    {
        const uint32_t hi10 = (uint32_t)(x >> 24) & 1023u, low13 = (uint32_t)x & 8191u;
        const uint32_t payload = half ? (((hi10 & 15u) << 13) | low13) : ((hi10 << 7) | (low13 >> 6));
        return ((uint32_t)half << 22) | (mask3 << 19) | payload;
    }
    // The items of one group: m6 bit 5 = k-mer 0 (oldest) counts ... bit 0 = k-mer 5.  a = the group's first item (0: none),
    // b = its second (only when both halves hold counting k-mers but not all six do).
    static KPAL_HEX_HD void split(uint64_t x, uint32_t m6, uint32_t &row, uint32_t &a, uint32_t &b)
    {
        row = row_of(x);
        const uint32_t up = m6 >> 3, lo = m6 & 7u;
        if (m6 == 63u) {
            a = full_item(x);
            b = 0;
        } else if (up) {
            a = half_item(x, 0, up);
            b = lo ? half_item(x, 1, lo) : 0u;
        } else {
            a = lo ? half_item(x, 1, lo) : 0u;
            b = 0;
        }
    }
    // item -> (payload in full-item layout: hi10 << 13 | low13, 23 bits; mask6).  Bits of bases a half item does not hold are zero.
    static KPAL_HEX_HD void unpack(uint32_t item, uint32_t &p23, uint32_t &m6)
    {
        if (item & kFull) {
            p23 = item & 0x7FFFFFu;
            m6 = 63u;
        } else {
            const uint32_t m3 = (item >> 19) & 7u, p17 = item & 0x1FFFFu;
            if (item & (1u << 22)) {      // lower half: hi10's low four bits | low13 -- already in place
                p23 = p17;
                m6 = m3;
            } else {                      // upper half: hi10 | low13's upper seven bits
                p23 = ((p17 >> 7) << 13) | ((p17 & 127u) << 6);
                m6 = m3 << 3;
            }
        }
    }
    // bin of k-mer i of an item in form i's histogram of its row
    static KPAL_HEX_HD uint32_t local_of(uint32_t p23, int i) { return (p23 >> (10 - 2 * i)) & 8191u; }
    // table entry (24-bit k-mer) of bin `local` of form i in the histogram of scrambled row `row`:
    //   k-mer i = hipart (10-2i bits) | bucket (11) | lopart (3+2i bits),  local = hipart << (3+2i) | lopart,  t = top three bits of lopart
    static KPAL_HEX_HD uint32_t bin_entry(uint32_t row, int i, uint32_t local)
    {
        const int s = 3 + 2 * i;
        const uint32_t lopart = local & ((1u << s) - 1u), hipart = local >> s;
        const uint32_t t = lopart >> (s - 3);
        return (hipart << (kBucketBits + s)) | ((row ^ smask(t)) << s) | lopart;
    }
    static KPAL_HEX_HD uint32_t kmer_of(uint32_t row, uint32_t p23, int i) { return bin_entry(row, i, local_of(p23, i)); }

    // ---- a lane's 48 bytes: three 16-byte chunks (codes c0..c2: 2 bits per base, first base most significant; bad flags b0..b2:
    // one bit per byte, first byte most significant) behind the left neighbour's last chunk (pc, pb) hold eight groups.
    // the 17-mer of group q (0..7): bases 6q-11 .. 6q+5 of the lane
    static KPAL_HEX_HD uint64_t group_x(uint32_t pc, uint32_t c0, uint32_t c1, uint32_t c2, int q)
    {
        const int o = 12 * q + 10;                       // first bit (from the top of pc) of the 34
        const int wi = o >> 5, sh = 30 - (o & 31);       // 34 bits from one pair of words: (o & 31) <= 30 for every q
        // (selects, not an array of the four words: indexed arrays stay in scratch memory on the device even when q is a constant)
        const uint32_t hi = wi == 0 ? pc : (wi == 1 ? c0 : c1), lo = wi == 0 ? c0 : (wi == 1 ? c1 : c2);
        const uint64_t pair = ((uint64_t)hi << 32) | lo;
        return (pair >> sh) & kXMask;
    }
    // bit (47 - j) set iff the k-mer ending at byte j of the lane counts: no flagged byte among bytes j-11 .. j
    static KPAL_HEX_HD uint64_t emit48(uint32_t pb, uint32_t b0, uint32_t b1, uint32_t b2)
    {
        uint64_t s = ((uint64_t)(pb & 0xFFFFu) << 48) | ((uint64_t)(b0 & 0xFFFFu) << 32) | ((uint64_t)(b1 & 0xFFFFu) << 16) | (b2 & 0xFFFFu);
        s |= s >> 1;
        s |= s >> 2;
        s |= s >> 4;      // 8
        s |= s >> 4;      // 12
        return ~s & 0xFFFFFFFFFFFFull;
    }
    static KPAL_HEX_HD uint32_t group_mask(uint64_t emit, int q) { return (uint32_t)(emit >> (42 - 6 * q)) & 63u; }
};

}  // namespace kpal
