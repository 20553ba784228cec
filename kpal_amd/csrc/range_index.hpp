// range_index.hpp -- index arithmetic of the BIN-RANGE merge of per-rank count tables (k >= 13 across GPUs, kpal_multi.hip):
// after ncclReduceScatter rank r of W = 2^w holds the merged bins [r * 4^k / W, (r + 1) * 4^k / W) -- the entries whose top w
// bits are r -- and Profile.balance (kpal/klib.py:285-298: c[i] += c[rc(i)]) needs, for every entry i of that range, the entry
// rc(i), which lies in the range named by the LOW digits of i (rc complements and reverses the k base-4 digits,
// klib.py:394-412): every rank holds 1/W of the mirror of every range.  So the ranks exchange: rank r sends rank q the entries
//     S(r -> q) = { j : top w bits of j = r,  top w bits of rc(j) = q },        4^k / W^2 entries,
// packed in the order of pos(j) below; the receiver looks the mirror of its entry i up at pos(rc(i)) of what owner(rc(i)) sent.
// Plain integer functions, shared by the pack / unpack kernels and tests/native/range_index_check.cpp (an emulation of W ranks
// on the CPU against the oracle-style balance) and restated in kpal_amd/dist.py for the torch.distributed variant.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define KPAL_RANGE_HD __host__ __device__ __forceinline__
#else
#define KPAL_RANGE_HD inline
#endif

namespace kpal {

struct RangeIndex {
    int k;            // digits
    int w;            // log2 of the number of ranks (0 .. 2k)
    KPAL_RANGE_HD uint64_t bins() const { return 1ull << (2 * k); }
    KPAL_RANGE_HD uint64_t range_bins() const { return bins() >> w; }                 // 4^k / W
    KPAL_RANGE_HD uint64_t pair_bins() const { return range_bins() >> w; }            // 4^k / W^2: what one rank sends another
    KPAL_RANGE_HD int low_bits() const { return 2 * ((w + 1) / 2); }                  // whole digits that hold the w bits deciding rc's range
    static KPAL_RANGE_HD uint64_t revcomp(uint64_t idx, int k)                        // klib.py:394-412
    {
        uint64_t x = ~idx;
        x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
        x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
        x = ((x >> 8) & 0x00FF00FF00FF00FFull) | ((x & 0x00FF00FF00FF00FFull) << 8);
        x = ((x >> 16) & 0x0000FFFF0000FFFFull) | ((x & 0x0000FFFF0000FFFFull) << 16);
        x = (x >> 32) | (x << 32);
        return x >> (64 - 2 * k);
    }
    KPAL_RANGE_HD uint32_t owner(uint64_t j) const { return w ? (uint32_t)(j >> (2 * k - w)) : 0u; }   // the rank whose range holds entry j
    // position of entry j inside S(owner(j) -> owner(rc(j))): the bits of j that neither name its own range (top w) nor decide the
    // destination (low_bits(); of those, the low_bits() - w bits of rc(j) below its top w stay free when w is odd) in one number
    KPAL_RANGE_HD uint64_t pos(uint64_t j) const
    {
        const int lb = low_bits(), spare = lb - w;                                   // 0 or 1
        const uint64_t mid = (j & (range_bins() - 1)) >> lb;                         // (for 2w > 2k - lb the fields overlap: callers keep 2 * low_bits() <= 2k)
        const uint64_t sub = spare ? ((revcomp(j, k) >> (2 * k - lb)) & 1ull) : 0ull;
        return (mid << spare) | sub;
    }
    KPAL_RANGE_HD bool valid() const { return k >= 1 && k <= 31 && w >= 0 && 2 * low_bits() <= 2 * k; }
};

}  // namespace kpal
