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

// 4 ASCII bytes -> the per-byte code field t (2 bits at the bottom of every byte) and the per-byte flag nz
// (0x80 where the byte is not in the alphabet).
// code = ((c >> 1) ^ (c >> 2)) & 3 maps A,C,G,T (either case) to 0,1,2,3 (kpal/klib.py:43-48);
// a byte is in the alphabet iff looking its code up in "ACGT" (one v_perm_b32 for the four
// bytes) gives the byte back with the case bit 0x20 ignored.
__device__ __forceinline__ void classify4(uint32_t w, uint32_t &t, uint32_t &nz)
{
    t = ((w >> 1) ^ (w >> 2)) & 0x03030303u;
    const uint32_t expect = __builtin_amdgcn_perm(0u, 0x54474341u /* "ACGT" */, t);
    const uint32_t x = (w & 0xDFDFDFDFu) ^ expect;
    nz = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;  // 0x80 per non-matching byte
}

// Byte 0 (lowest address) is the oldest base -> most significant digit.  Two shift-or steps bring the four
// 2-bit codes of t together in the TOP byte (b0 b1 b2 b3; the fields never meet, the byte comes out clean):
//   x = t << 10 | t   puts b0 next to b1 (bits 11:8) and b2 next to b3 (bits 27:24);  x << 20 | x  joins them.
// (v_lshl_or_b32 spelled out: written as shifts and ORs the compiler turns each gather into a multiplication by
// (1 + 2^10)(1 + 2^20), and a 32-bit v_mul_lo_u32 issues at a quarter of the rate)
template <int SH>
__device__ __forceinline__ uint32_t shl_or(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(SH), "v"(b));
    return r;
}

__device__ __forceinline__ uint32_t codes_to_top_byte(uint32_t t)
{
    const uint32_t x = shl_or<10>(t, t);
    return shl_or<20>(x, x);
}

// The same for the flags of m (one bit per byte at bit position 0 and / or 4 of the byte: the flags of TWO dwords
// can ride together): the flags at position 0 come together in the low nibble of the top byte, those at position 4
// in the high nibble.
__device__ __forceinline__ uint32_t flags_to_top_byte(uint32_t m)
{
    const uint32_t x = shl_or<9>(m, m);
    return shl_or<18>(x, x);
}

// (kept for callers that encode single dwords) 8 bits of codes + 4 bad flags, first byte most significant
__device__ __forceinline__ void encode4(uint32_t w, uint32_t &code8, uint32_t &bad4)
{
    uint32_t t, nz;
    classify4(w, t, nz);
    code8 = codes_to_top_byte(t) >> 24;
    bad4 = flags_to_top_byte(nz >> 7) >> 24;
}

struct Chunk {
    uint32_t codes;  // 16 bases x 2 bits, base 0 (lowest address) in bits 31:30
    uint32_t bad;    // 16 flags, base 0 in bit 15
};

// ~45 VALU: 8 per dword to classify; the four 2-bit codes of a dword are gathered into one byte by ONE v_dot4_u32_u8 (the
// byte-wise dot product with the weights 64, 16, 4, 1: the oldest base is the most significant digit) and the four bytes joined
// by three v_lshl_or_b32; the flags (0x80 per bad byte) of a PAIR of dwords by two chained dot products (weights 128 .. 16 and
// 8 .. 1: 0x80 x the pair's eight flags) and the two pairs by a shift and a v_lshl_or_b32.  (Round 3's shift-or gathers and
// byte permutes took 54; KPAL_ENCODE_SHIFTS builds them for A/B.)
__device__ __forceinline__ Chunk encode16(uint4 v)
{
    uint32_t t0, t1, t2, t3, z0, z1, z2, z3;
    classify4(v.x, t0, z0);
    classify4(v.y, t1, z1);
    classify4(v.z, t2, z2);
    classify4(v.w, t3, z3);
    Chunk r;
#if defined(KPAL_ENCODE_SHIFTS)
    const uint32_t c0 = codes_to_top_byte(t0), c1 = codes_to_top_byte(t1), c2 = codes_to_top_byte(t2), c3 = codes_to_top_byte(t3);
    const uint32_t c01 = __builtin_amdgcn_perm(c0, c1, 0x07030000u);      // bytes 3, 2 = top bytes of c0, c1
    const uint32_t c23 = __builtin_amdgcn_perm(c2, c3, 0x07030000u);
    const uint32_t f01 = flags_to_top_byte((z0 >> 3) | (z1 >> 7));   // top byte = flags of dword 0 | dword 1
    const uint32_t f23 = flags_to_top_byte((z2 >> 3) | (z3 >> 7));
    r.codes = __builtin_amdgcn_perm(c01, c23, 0x07060302u);
    r.bad = __builtin_amdgcn_perm(f01, f23, 0x0c0c0703u);
#else
    constexpr uint32_t kDigits = 0x01041040u;                            // byte 0 (oldest base) x 64, byte 1 x 16, byte 2 x 4, byte 3 x 1
    const uint32_t d0 = __builtin_amdgcn_udot4(t0, kDigits, 0u, false), d1 = __builtin_amdgcn_udot4(t1, kDigits, 0u, false);
    const uint32_t d2 = __builtin_amdgcn_udot4(t2, kDigits, 0u, false), d3 = __builtin_amdgcn_udot4(t3, kDigits, 0u, false);
    // (plain C, not the inline-assembly shl_or: a dot product's result needs wait states before a VALU instruction may read it,
    // and hipcc's hazard recogniser does not look into inline assembly -- the first build of this read stale registers)
    r.codes = (d0 << 24) | ((d1 << 16) | ((d2 << 8) | d3));
    constexpr uint32_t kHigh = 0x10204080u, kLow = 0x01020408u;          // byte 0 (oldest) the highest flag of its nibble
    const uint32_t f01 = __builtin_amdgcn_udot4(z1, kLow, __builtin_amdgcn_udot4(z0, kHigh, 0u, false), false);   // 0x80 x the eight flags
    const uint32_t f23 = __builtin_amdgcn_udot4(z3, kLow, __builtin_amdgcn_udot4(z2, kHigh, 0u, false), false);
    r.bad = (f01 << 1) | (f23 >> 7);
#endif
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

// wave_step on data that was fetched earlier (software prefetch): `raw` = fetch_chunk(s, c).  `edge` (wave-uniform)
// = the step may touch the ends of the fed range; true for an interior step is harmless (the range logic is then
// evaluated and changes nothing), so a caller may pass one flag for a whole run of steps.
template <int K>
__device__ __forceinline__ void encode_step(const Span &s, uint64_t step, const uint4 raw, Chunk &carry,
                                            uint64_t &window, uint32_t &mask, bool edge)
{
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)(step * 64 + lane);
    Chunk cur = encode16(raw);
    if (edge) range_fix(s, c, cur);   // wave-uniform branch
    const uint32_t pc = from_left_lane(cur.codes, carry.codes);
    const uint32_t pb = from_left_lane(cur.bad, carry.bad);
    carry.codes = __builtin_amdgcn_readlane(cur.codes, 63);
    carry.bad = __builtin_amdgcn_readlane(cur.bad, 63);
    window = ((uint64_t)pc << 32) | cur.codes;
    mask = emit_mask<K>(pb, cur.bad);
    if (edge) mask &= emit_from_mask(s, c);
}

// steps [step0, step1) and the chunk left of them lie inside the fed range and right of emit_from
__device__ __forceinline__ bool interior_steps(const Span &s, uint64_t step0, uint64_t step1)
{
    return step0 * 64 >= 1 && (step0 * 64 - 1) * 16 >= s.lo && step1 * 64 * 16 <= s.hi && s.emit_from <= step0 * 64 * 16;
}

template <int K>
__device__ __forceinline__ void encode_step(const Span &s, uint64_t step, const uint4 raw, Chunk &carry,
                                            uint64_t &window, uint32_t &mask)
{
    encode_step<K>(s, step, raw, carry, window, mask, !interior_steps(s, step, step + 1));
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

// One wave-step at absolute step index `step` (interior fast path chosen per wave).
template <int K>
__device__ __forceinline__ void part_step(const Span &s, uint64_t step, Chunk &carry, uint64_t &window, uint32_t &mask)
{
    const int lane = threadIdx.x & 63;
    if (interior_range(s, step * 64, step * 64 + 64)) wave_step<K, false>(s, (int64_t)(step * 64 + lane), carry, window, mask);
    else wave_step<K, true>(s, (int64_t)(step * 64 + lane), carry, window, mask);
}

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain the
// vector-memory counter, so line stores issued in a flush phase stay in flight while the next
// tile is placed.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

}  // namespace kpal
