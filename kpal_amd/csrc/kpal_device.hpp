// kpal_device.hpp -- device-side building blocks shared by the counting kernels (gfx950).
//
// Stream model: the input is a flat byte stream.  It is cut into 16-byte CHUNKS on 16-byte
// aligned addresses; one lane owns one chunk per step, a wave owns 64 consecutive chunks
// (1 KiB, one coalesced global_load_dwordx4).  A lane turns its 16 ASCII bytes into a 32-bit
// word of 2-bit codes plus a 16-bit "bad" mask (byte outside AaCcGgTt or outside the fed
// range), fetches its left neighbour's word with one cross-lane move, and then owns the 16
// k-mers that END inside its chunk.  This is the reference's rolling window
// (kpal/klib.py:157-168) evaluated at all positions at once: a k-mer ending at byte i is
// counted iff bytes i-k+1..i are all in AaCcGgTt.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kpal {

// splitmix64 finaliser of the synthetic-read generator
__host__ __device__ inline uint64_t mix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// Profile.reverse_complement (kpal/klib.py:394-412) as bit operations: complement, reverse the
// k 2-bit digits.
__host__ __device__ inline uint64_t revcomp(uint64_t x, int k)
{
    uint64_t y = ~x;
#if defined(__HIP_DEVICE_COMPILE__)
    y = __brevll(y);
#else
    y = ((y >> 1) & 0x5555555555555555ULL) | ((y & 0x5555555555555555ULL) << 1);
    y = ((y >> 2) & 0x3333333333333333ULL) | ((y & 0x3333333333333333ULL) << 2);
    y = ((y >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((y & 0x0F0F0F0F0F0F0F0FULL) << 4);
    y = ((y >> 8) & 0x00FF00FF00FF00FFULL) | ((y & 0x00FF00FF00FF00FFULL) << 8);
    y = ((y >> 16) & 0x0000FFFF0000FFFFULL) | ((y & 0x0000FFFF0000FFFFULL) << 16);
    y = (y >> 32) | (y << 32);
#endif
    // full bit reversal also swapped the two bits inside each digit: swap them back
    y = ((y >> 1) & 0x5555555555555555ULL) | ((y & 0x5555555555555555ULL) << 1);
    return y >> (64 - 2 * k);
}

// 4 ASCII bytes -> 8 bits of codes (first byte most significant) + 4 bad flags (same order).
// code = ((c >> 1) ^ (c >> 2)) & 3 maps A,C,G,T (either case) to 0,1,2,3 (kpal/klib.py:43-48);
// a byte is in the alphabet iff looking its code up in "ACGT" (one v_perm_b32 for the four
// bytes) gives the byte back with the case bit 0x20 ignored.  ~17 VALU ops per dword.
__device__ __forceinline__ void encode4(uint32_t w, uint32_t &code8, uint32_t &bad4)
{
    const uint32_t t = ((w >> 1) ^ (w >> 2)) & 0x03030303u;
    const uint32_t expect = __builtin_amdgcn_perm(0u, 0x54474341u /* "ACGT" */, t);
    const uint32_t x = (w & 0xDFDFDFDFu) ^ expect;
    const uint32_t nz = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;  // 0x80 per non-matching byte
    // byte 0 (lowest address) is the oldest base -> most significant digit.  The 24-bit multiply
    // gathers the codes of bytes 0..2 into bits 21:16 (fields never overlap, so no carries).
    const uint32_t g = __umul24(t, 0x100401u);
    code8 = ((g >> 14) & 0xFCu) | (t >> 24);
    const uint32_t h = __umul24(nz >> 7, 0x040201u);  // flags of bytes 0..2 -> bits 18:16
    bad4 = ((h >> 15) & 0xEu) | (nz >> 31);
}

struct Chunk {
    uint32_t codes;  // 16 bases x 2 bits, base 0 (lowest address) in bits 31:30
    uint32_t bad;    // 16 flags, base 0 in bit 15
};

__device__ __forceinline__ Chunk encode16(uint4 v)
{
    uint32_t c0, c1, c2, c3, m0, m1, m2, m3;
    encode4(v.x, c0, m0);
    encode4(v.y, c1, m1);
    encode4(v.z, c2, m2);
    encode4(v.w, c3, m3);
    Chunk r;
    r.codes = (c0 << 24) | (c1 << 16) | (c2 << 8) | c3;
    r.bad = (m0 << 12) | (m1 << 8) | (m2 << 4) | m3;
    return r;
}

// Geometry of one fed buffer in chunk space.
struct Span {
    const uint4 *base;   // 16-byte aligned address at or below the first fed byte
    uint64_t lo;         // position (relative to base) of the first fed byte, 0..15
    uint64_t hi;         // position one past the last fed byte
    uint64_t emit_from;  // k-mers ending before this position are not counted (halo of a host staging piece)
    uint64_t nchunks;    // ceil(hi / 16)
};

// Raw 16 bytes of chunk c, zeros when c is outside the buffer (no memory access).
__device__ __forceinline__ uint4 fetch_chunk(const Span &s, int64_t c)
{
    if (c < 0 || (uint64_t)c >= s.nchunks) return make_uint4(0u, 0u, 0u, 0u);
    return s.base[c];
}

// Flag the bytes of chunk c that lie outside the fed range [lo, hi) as bad.
__device__ __forceinline__ void range_fix(const Span &s, int64_t c, Chunk &r)
{
    if (c < 0 || (uint64_t)c >= s.nchunks) {
        r.codes = 0;
        r.bad = 0xFFFFu;
        return;
    }
    const uint64_t p0 = (uint64_t)c * 16;
    if (p0 < s.lo || p0 + 16 > s.hi) {
        const uint32_t a = s.lo > p0 ? (uint32_t)(s.lo - p0) : 0u;                    // first in-range byte
        const uint32_t b = s.hi < p0 + 16 ? (uint32_t)(s.hi > p0 ? s.hi - p0 : 0) : 16u;  // one past last
        const uint32_t in_range = (a < b) ? (((1u << (16 - a)) - 1u) & ~((1u << (16 - b)) - 1u)) : 0u;
        r.bad |= ~in_range & 0xFFFFu;
    }
}

// Load and encode chunk c; bytes outside [lo, hi) are flagged bad; c >= nchunks or c < 0 gives
// an all-bad chunk without touching memory.
__device__ __forceinline__ Chunk load_chunk(const Span &s, int64_t c)
{
    Chunk r = encode16(fetch_chunk(s, c));
    range_fix(s, c, r);
    return r;
}

// Mask (bit 15-j for the k-mer ending at byte j of the chunk) of k-mers that may be counted:
// no bad byte among the k bytes ending there, and end position >= emit_from.
template <int K>
__device__ __forceinline__ uint32_t emit_mask(uint32_t prev_bad, uint32_t cur_bad)
{
    uint32_t b = (prev_bad << 16) | cur_bad;  // window of 32 bytes, oldest in bit 31
    // smear: bit t of s set iff any of bits t..t+K-1 of b is set
    uint32_t s = b;
    int have = 1;
#pragma unroll
    for (int step = 1; step < 32; step <<= 1) {
        if (have * 2 <= K) {
            s |= s >> have;
            have *= 2;
        }
    }
    if (have < K) s |= s >> (K - have);
    return ~s & 0xFFFFu;
}

__device__ __forceinline__ uint32_t emit_mask_rt(int k, uint32_t prev_bad, uint32_t cur_bad)
{
    uint32_t b = (prev_bad << 16) | cur_bad;
    uint32_t s = b;
    int have = 1;
    while (have * 2 <= k) {
        s |= s >> have;
        have *= 2;
    }
    if (have < k) s |= s >> (k - have);
    return ~s & 0xFFFFu;
}

__device__ __forceinline__ uint32_t emit_from_mask(const Span &s, int64_t c)
{
    const uint64_t p0 = (uint64_t)c * 16;
    if (s.emit_from <= p0) return 0xFFFFu;
    if (s.emit_from >= p0 + 16) return 0u;
    return (1u << (16 - (uint32_t)(s.emit_from - p0))) - 1u;
}

// lane l <- lane l-1 across the whole wave (DPP wave_shr:1); lane 0 receives `lane0_value`.
__device__ __forceinline__ uint32_t from_left_lane(uint32_t v, uint32_t lane0_value)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0_value, (int)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}

// One wave-step: returns this lane's 64-bit window (previous chunk's codes in the high word)
// and the emit mask.  `carry` holds lane 63's chunk of the previous step (or the chunk left of
// the wave's first chunk); it is updated for the next step.  EDGE = false is the interior fast
// path: the 64 chunks and the halo chunk lie wholly inside the fed range and right of
// emit_from, so no range logic is evaluated.
template <int K, bool EDGE>
__device__ __forceinline__ void wave_step(const Span &s, int64_t c, Chunk &carry, uint64_t &window,
                                          uint32_t &mask)
{
    Chunk cur;
    if constexpr (EDGE) cur = load_chunk(s, c);
    else cur = encode16(s.base[c]);
    const uint32_t pc = from_left_lane(cur.codes, carry.codes);
    const uint32_t pb = from_left_lane(cur.bad, carry.bad);
    carry.codes = __builtin_amdgcn_readlane(cur.codes, 63);
    carry.bad = __builtin_amdgcn_readlane(cur.bad, 63);
    window = ((uint64_t)pc << 32) | cur.codes;
    mask = emit_mask<K>(pb, cur.bad);
    if constexpr (EDGE) mask &= emit_from_mask(s, c);
}

// wave_step on data that was fetched earlier (software prefetch): `raw` = fetch_chunk(s, c).
template <int K>
__device__ __forceinline__ void encode_step(const Span &s, uint64_t step, const uint4 raw, Chunk &carry,
                                            uint64_t &window, uint32_t &mask)
{
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)(step * 64 + lane);
    Chunk cur = encode16(raw);
    const bool edge = !(step * 64 >= 1 && (step * 64 - 1) * 16 >= s.lo && (step * 64 + 64) * 16 <= s.hi && s.emit_from <= step * 64 * 16);
    if (edge) range_fix(s, c, cur);   // wave-uniform branch
    const uint32_t pc = from_left_lane(cur.codes, carry.codes);
    const uint32_t pb = from_left_lane(cur.bad, carry.bad);
    carry.codes = __builtin_amdgcn_readlane(cur.codes, 63);
    carry.bad = __builtin_amdgcn_readlane(cur.bad, 63);
    window = ((uint64_t)pc << 32) | cur.codes;
    mask = emit_mask<K>(pb, cur.bad);
    if (edge) mask &= emit_from_mask(s, c);
}

// true iff chunks [c0 - 1, c1) are wholly inside [lo, hi) and every k-mer ending in [c0, c1) is emitted
__device__ __forceinline__ bool interior_range(const Span &s, uint64_t c0, uint64_t c1)
{
    return c0 >= 1 && (c0 - 1) * 16 >= s.lo && c1 * 16 <= s.hi && s.emit_from <= c0 * 16;
}

// k-mer ending at byte j (0..15) of the lane's chunk.
template <int K>
__device__ __forceinline__ uint32_t kmer_at(uint64_t window, int j)
{
    constexpr uint64_t M = (K >= 16) ? 0xFFFFFFFFULL : ((1ULL << (2 * K)) - 1ULL);
    return (uint32_t)((window >> (2 * (15 - j))) & M);
}

}  // namespace kpal
