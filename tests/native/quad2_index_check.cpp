// quad2_index_check.cpp -- host check of kpal_amd/csrc/quad2_index.hpp (compiled and run by
// tests/test_abi_and_host.py::test_quad2_index_arithmetic; g++, no GPU).
//
// The finalisation stage of the two-level quad pipeline (k = 13..16) is index arithmetic: where a histogram workgroup
// stages its bins (word_pos / bin_of_word), where the finalisation finds form i of a table entry (stage_pos), which
// entries one workgroup owns (set_base / entry / stream_entry) and where the reverse complement of an entry sits in
// the partner set (partner_lo7 / partner_hi7).  The device kernels use exactly these functions; here they drive a CPU
// emulation at k = 13 -- random forms and a random table, staged the way quad_hist_kernel stages them, finalised the
// way quad2_finalize_kernel does (both modes) -- compared with v = T + sum of the forms and out[i] = v[i] + v[rc(i)]
// computed directly.  For k = 14..16 (tables too large for a CPU test) the same properties are checked on sampled
// sets: bijections, coverage, partner relation, contiguity of the streams.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../kpal_amd/csrc/quad2_index.hpp"

using namespace kpal;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

#define CHECK(cond, ...)                         \
    do {                                         \
        if (!(cond)) {                           \
            fprintf(stderr, "FAIL k=%d: ", K);   \
            fprintf(stderr, __VA_ARGS__);        \
            fprintf(stderr, "\n");               \
            return 1;                            \
        }                                        \
    } while (0)

// table entry of bin `local` of form i in the histogram of scrambled bucket (sc, sf): quad_bin_index of quad_kernels.hpp restated
template <int K>
static uint64_t bin_entry(int i, uint32_t sc, uint32_t sf, uint32_t local)
{
    using Q = Quad2Index<K>;
    const int s = 7 + 2 * i;
    const uint32_t lopart = local & ((1u << s) - 1u), hipart = local >> s, t = lopart >> (s - 4);
    return ((uint64_t)hipart << (Q::CB + 9 + s)) | ((uint64_t)(sc ^ Q::smask1(t)) << (9 + s)) | ((uint64_t)(sf ^ Q::smask(t)) << s) | lopart;
}

template <int K>
static int check_properties(int sample_sets)
{
    using Q = Quad2Index<K>;
    const uint64_t mask = (1ull << (2 * K)) - 1ull;
    // reverse complement: involution, matches the digit definition
    for (int it = 0; it < 2000; ++it) {
        const uint64_t x = rnd() & mask;
        uint64_t want = 0;
        for (int d = 0; d < K; ++d) want |= (3ull - ((x >> (2 * d)) & 3ull)) << (2 * (K - 1 - d));
        CHECK(Q::revcomp(x) == want && Q::revcomp(want) == x, "revcomp(%llx)", (unsigned long long)x);
    }
    // staging: word_pos / bin_of_word (histogram side) against stage_pos (finalisation side)
    for (int it = 0; it < 200000; ++it) {
        const int i = (int)(rnd() & 3);
        const uint32_t sc = (uint32_t)rnd() & Q::kCoarseMask, sf = (uint32_t)rnd() & 511u, o = (uint32_t)rnd() & 8191u;
        const uint32_t local = Q::bin_of_word(i, o);
        CHECK(local < 8192u, "bin_of_word range");
        const uint64_t idx = bin_entry<K>(i, sc, sf, local);
        CHECK(idx <= mask, "entry range");
        CHECK(Q::stage_pos(i, idx) == Q::word_pos(i, sc, sf, o), "stage_pos != word_pos (form %d sc %u sf %u word %u)", i, sc, sf, o);
        CHECK(Q::stage_pos(i, idx) < (4ull << (2 * K)), "stage_pos range");
    }
    for (int i = 0; i < 4; ++i) {   // bin_of_word is a bijection of 0..8191, eight consecutive words are eight consecutive bins
        std::vector<char> seen(8192, 0);
        for (uint32_t o = 0; o < 8192; ++o) {
            const uint32_t l = Q::bin_of_word(i, o);
            CHECK(!seen[l], "bin_of_word not injective");
            seen[l] = 1;
            if (o & 7u) CHECK(l == Q::bin_of_word(i, o & ~7u) + (o & 7u), "bin_of_word: words of a vector are not consecutive bins");
        }
    }
    // sets
    CHECK(__builtin_popcountll(Q::kFree) == 14, "free bits");
    for (int it = 0; it < sample_sets; ++it) {
        const uint32_t R = (uint32_t)rnd() & (Q::kSets - 1u);
        const uint64_t base = Q::set_base(R);
        CHECK((base & Q::kFree) == 0 && base <= mask, "set_base");
        const uint64_t pbase = Q::partner_base(base);
        CHECK((pbase & Q::kFree) == 0 && Q::partner_base(pbase) == base, "partner_base is not an involution");
        if (K % 2 == 1) CHECK(pbase != base, "odd k has no self-paired set");
        for (int src = 0; src < 5; ++src) {
            std::vector<char> seen(16384, 0);
            uint64_t prev = 0;
            int breaks = 0;
            for (uint32_t q = 0; q < 16384; ++q) {
                uint32_t lo7, hi7;
                Q::stream_entry(src, q, lo7, hi7);
                CHECK(lo7 < 128 && hi7 < 128 && (lo7 & 7u) == (q & 7u), "stream_entry range");
                CHECK(!seen[hi7 * 128 + lo7], "stream_entry not injective (source %d)", src);
                seen[hi7 * 128 + lo7] = 1;
                const uint64_t idx = Q::entry(base, lo7, hi7);
                CHECK((idx & ~Q::kFree) == base, "entry leaves its set");
                // partner relation
                const uint64_t p = Q::revcomp(idx);
                CHECK(p == Q::entry(pbase, Q::partner_lo7(hi7), Q::partner_hi7(lo7)), "partner position");
                const uint64_t addr = src < 4 ? Q::stage_pos(src, idx) : idx;
                if (src < 4 && (q & 15u) == 0) CHECK((addr & 15u) == 0, "a vector of sixteen staged counts is not aligned (source %d)", src);
                if (src < 4 && (q & 15u) != 0) CHECK(addr == prev + 1, "the sixteen counts of a vector are not consecutive (source %d)", src);
                if (q && addr != prev + 1) ++breaks;
                prev = addr;
            }
            // pieces: forms 0..2 32 of 512 words (one t of one true bucket each), form 3 and the table 128 of 128
            static const int max_pieces[5] = {32, 32, 32, 128, 128};
            CHECK(breaks + 1 <= max_pieces[src], "source %d comes in %d pieces", src, breaks + 1);
        }
    }
    // every entry belongs to exactly one set; canonical pairs cover all sets once
    {
        uint64_t canonical = 0, self = 0;
        for (uint32_t R = 0; R < Q::kSets; ++R) {
            const uint64_t base = Q::set_base(R), pbase = Q::partner_base(base);
            if (base <= pbase) canonical += base == pbase ? 1 : 2;
            if (base == pbase) ++self;
        }
        CHECK(canonical == Q::kSets, "canonical pairs cover %llu of %u sets", (unsigned long long)canonical, Q::kSets);
        if (K % 2 == 0) CHECK(self > 0, "even k has self-paired sets");
    }
    return 0;
}

// Full emulation at one K (tables of 4^K entries on the host).
template <int K>
static int emulate()
{
    using Q = Quad2Index<K>;
    const uint64_t n = 1ull << (2 * K);
    std::vector<quad2_stage_t> F[4];
    std::vector<quad2_stage_t> stage(4 * n, 0);
    std::vector<int64_t> T(n, 0), out(n, 0), plain(n, 0);
    for (int i = 0; i < 4; ++i) F[i].assign(n, 0);
    for (uint64_t e = 0; e < n / 3; ++e) {   // a third of the entries touched, per form
        for (int i = 0; i < 4; ++i) F[i][rnd() & (n - 1)] = (quad2_stage_t)(1 + (rnd() % 9));
    }
    for (uint64_t e = 0; e < n / 50; ++e) T[rnd() & (n - 1)] = (int64_t)(rnd() % 1000);
    T[rnd() & (n - 1)] = (int64_t)1 << 40;   // counts beyond 32 bits stay exact
    for (int e = 0; e < 1000; ++e) F[rnd() & 3][rnd() & (n - 1)] = (quad2_stage_t)(kQuad2StageLimit - 1);   // the largest staged count
    // ---- staging, the way quad_hist_kernel does it: workgroup (sc, sf), plane i, word o <- its bin bin_of_word(i, o)
    for (uint32_t sc = 0; sc <= Q::kCoarseMask; ++sc)
        for (uint32_t sf = 0; sf < 512; ++sf)
            for (int i = 0; i < 4; ++i)
                for (uint32_t o = 0; o < 8192; ++o)
                    stage[Q::word_pos(i, sc, sf, o)] = F[i][bin_entry<K>(i, sc, sf, Q::bin_of_word(i, o))];
    for (uint64_t idx = 0; idx < n; idx += 1 + (rnd() & 3))
        for (int i = 0; i < 4; ++i) CHECK(stage[Q::stage_pos(i, idx)] == F[i][idx], "staged form %d of entry %llx", i, (unsigned long long)idx);
    // ---- finalisation
    std::vector<uint64_t> lds(128 * Q::kRowStride);
    std::vector<char> written(n, 0);
    for (int balance = 0; balance < 2; ++balance) {
        std::vector<int64_t> &dst = balance ? out : plain;
        std::fill(written.begin(), written.end(), 0);
        for (uint32_t R = 0; R < Q::kSets; ++R) {
            const uint64_t base = Q::set_base(R), pbase = Q::partner_base(base);
            if (balance && base > pbase) continue;
            const bool self = base == pbase;
            std::fill(lds.begin(), lds.end(), 0);
            for (int src = 0; src < 5; ++src)
                for (uint32_t q = 0; q < 16384; ++q) {
                    uint32_t lo7, hi7;
                    Q::stream_entry(src, q, lo7, hi7);
                    const uint64_t idx = Q::entry(base, lo7, hi7);
                    const uint64_t v = src < 4 ? stage[Q::stage_pos(src, idx)] : (uint64_t)T[idx];
                    lds[hi7 * Q::kRowStride + lo7] += v;
                    if (balance && self) lds[Q::partner_hi7(lo7) * Q::kRowStride + Q::partner_lo7(hi7)] += v;
                    if (balance && !self) {
                        const uint64_t jdx = Q::entry(pbase, lo7, hi7);
                        const uint64_t w = src < 4 ? stage[Q::stage_pos(src, jdx)] : (uint64_t)T[jdx];
                        lds[Q::partner_hi7(lo7) * Q::kRowStride + Q::partner_lo7(hi7)] += w;
                    }
                }
            for (uint32_t hi7 = 0; hi7 < 128; ++hi7)
                for (uint32_t lo7 = 0; lo7 < 128; ++lo7) {
                    const uint64_t idx = Q::entry(base, lo7, hi7);
                    CHECK(!written[idx], "entry written twice");
                    written[idx] = 1;
                    dst[idx] = (int64_t)lds[hi7 * Q::kRowStride + lo7];
                    if (balance && !self) {
                        const uint64_t jdx = Q::entry(pbase, lo7, hi7);
                        CHECK(!written[jdx], "entry written twice");
                        written[jdx] = 1;
                        dst[jdx] = (int64_t)lds[Q::partner_hi7(lo7) * Q::kRowStride + Q::partner_lo7(hi7)];
                    }
                }
        }
        for (uint64_t idx = 0; idx < n; ++idx) {
            CHECK(written[idx], "entry %llx never written", (unsigned long long)idx);
            const uint64_t p = Q::revcomp(idx);
            const int64_t v = T[idx] + F[0][idx] + F[1][idx] + F[2][idx] + F[3][idx];
            const int64_t vp = T[p] + F[0][p] + F[1][p] + F[2][p] + F[3][p];
            CHECK(dst[idx] == (balance ? v + vp : v), "%s entry %llx: %lld", balance ? "balanced" : "plain", (unsigned long long)idx, (long long)dst[idx]);
        }
    }
    return 0;
}

// The finalisation of single sets with forms and table given by a hash of (source, entry) instead of arrays: reaches the
// SELF-PAIRED sets of even k (rc maps the set onto itself: every source adds to its own entry and to the partner's).
static uint64_t hval(int src, uint64_t idx)
{
    uint64_t x = idx * 0x9E3779B97F4A7C15ull + (uint64_t)src * 0xC2B2AE3D27D4EB4Full;
    x ^= x >> 29;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 32;
    return (x & 7u) ? 0 : (src < 4 ? (x >> 8) & 0xFFFFu : (x >> 8) & 0xFFFFFFFFFFull);
}

template <int K>
static int emulate_sets()
{
    using Q = Quad2Index<K>;
    std::vector<uint32_t> todo;
    for (uint32_t R = 0; R < Q::kSets && todo.size() < 3; ++R)   // the first self-paired sets (even k)
        if (Q::partner_base(Q::set_base(R)) == Q::set_base(R)) todo.push_back(R);
    if (K % 2 == 0) CHECK(!todo.empty(), "no self-paired set found");
    for (int it = 0; it < 3; ++it) todo.push_back((uint32_t)rnd() & (Q::kSets - 1u));
    std::vector<uint64_t> lds(128 * Q::kRowStride);
    auto v = [](uint64_t idx) { return hval(0, idx) + hval(1, idx) + hval(2, idx) + hval(3, idx) + hval(4, idx); };
    for (uint32_t R : todo) {
        uint64_t base = Q::set_base(R), pbase = Q::partner_base(base);
        if (base > pbase) std::swap(base, pbase);
        const bool self = base == pbase;
        std::fill(lds.begin(), lds.end(), 0);
        for (int src = 0; src < 5; ++src)
            for (uint32_t q = 0; q < 16384; ++q) {
                uint32_t lo7, hi7;
                Q::stream_entry(src, q, lo7, hi7);
                const uint64_t a = hval(src, Q::entry(base, lo7, hi7));
                lds[hi7 * Q::kRowStride + lo7] += a;
                lds[Q::partner_hi7(lo7) * Q::kRowStride + Q::partner_lo7(hi7)] += self ? a : hval(src, Q::entry(pbase, lo7, hi7));
            }
        for (uint32_t hi7 = 0; hi7 < 128; ++hi7)
            for (uint32_t lo7 = 0; lo7 < 128; ++lo7) {
                const uint64_t idx = Q::entry(base, lo7, hi7), jdx = Q::entry(pbase, lo7, hi7);
                CHECK(lds[hi7 * Q::kRowStride + lo7] == v(idx) + v(Q::revcomp(idx)), "set %u entry %llx", R, (unsigned long long)idx);
                CHECK(lds[Q::partner_hi7(lo7) * Q::kRowStride + Q::partner_lo7(hi7)] == v(jdx) + v(Q::revcomp(jdx)), "set %u partner entry %llx", R,
                      (unsigned long long)jdx);
            }
    }
    return 0;
}

int main(int argc, char **argv)
{
    const bool full14 = argc > 1 && !strcmp(argv[1], "--k14");
    if (check_properties<12>(6) || check_properties<13>(6) || check_properties<14>(6) || check_properties<15>(4) || check_properties<16>(3)) return 1;
    if (emulate_sets<12>() || emulate_sets<13>() || emulate_sets<14>() || emulate_sets<15>() || emulate_sets<16>()) return 1;
    if (emulate<12>()) return 1;             // the one-level pipeline at k = 12 stages its forms the same way (even k: self-paired sets)
    if (emulate<13>()) return 1;
    if (full14 && emulate<14>()) return 1;   // 12 GB of host memory: on request (self-paired sets exist only for even k)
    puts("QUAD2_INDEX_OK");
    return 0;
}
