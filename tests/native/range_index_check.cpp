// range_index_check.cpp -- CPU emulation of the bin-range merge (kpal_amd/csrc/range_index.hpp): W ranks with random tables,
// reduce-scatter (every rank keeps the sum of its range), pack per destination, all-to-all, unpack + add -- every rank's range
// must equal the same range of Profile.balance (klib.py:285-298) of the summed table.  k = 2 .. 8 exhaustively for every
// W = 1 .. 64 the arithmetic allows, and the packing bijection at k = 13 .. 16 for W = 2, 4, 8 on samples.
// Test infrastructure; run by tests/test_abi_and_host.py::test_range_index_arithmetic.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../kpal_amd/csrc/range_index.hpp"

using kpal::RangeIndex;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

int main()
{
    int failures = 0, cases = 0;
    for (int k = 2; k <= 8; ++k)
        for (int w = 0; w <= 6; ++w) {
            const RangeIndex R{k, w};
            if (!R.valid()) continue;
            ++cases;
            const uint64_t B = R.bins(), n1 = R.range_bins(), n2 = R.pair_bins();
            const uint32_t W = 1u << w;
            // the merged (summed) table and its balance, the reference way
            std::vector<int64_t> merged(B), want(B);
            for (uint64_t i = 0; i < B; ++i) merged[i] = (int64_t)(rnd() % 1000);
            for (uint64_t i = 0; i < B; ++i) want[i] = merged[i] + merged[RangeIndex::revcomp(i, k)];
            // every rank packs its range for every destination
            std::vector<std::vector<int64_t>> send(W, std::vector<int64_t>(n1, -1));
            for (uint32_t r = 0; r < W; ++r)
                for (uint64_t l = 0; l < n1; ++l) {
                    const uint64_t j = (uint64_t)r * n1 + l;
                    if (R.owner(j) != r) ++failures;
                    const uint32_t q = R.owner(RangeIndex::revcomp(j, k));
                    const uint64_t p = R.pos(j);
                    if (p >= n2 || send[r][(uint64_t)q * n2 + p] != -1) ++failures;   // a bijection onto W blocks of n2
                    else send[r][(uint64_t)q * n2 + p] = merged[j];
                }
            // all-to-all: recv[r][q * n2 ..] = send[q][r * n2 ..]; unpack
            for (uint32_t r = 0; r < W && !failures; ++r)
                for (uint64_t l = 0; l < n1; ++l) {
                    const uint64_t i = (uint64_t)r * n1 + l;
                    const uint64_t j = RangeIndex::revcomp(i, k);
                    const uint32_t q = R.owner(j);
                    const int64_t mirror = send[q][(uint64_t)r * n2 + R.pos(j)];
                    if (merged[i] + mirror != want[i]) ++failures;
                }
            if (failures) {
                printf("k=%d W=%u: FAILED\n", k, W);
                return 1;
            }
        }
    // large k: the packing is a bijection on sampled slices (pos < n2, distinct positions for distinct entries of one destination)
    for (int k = 13; k <= 16; ++k)
        for (int w = 1; w <= 3; ++w) {
            const RangeIndex R{k, w};
            if (!R.valid()) { ++failures; continue; }
            ++cases;
            const uint64_t n1 = R.range_bins(), n2 = R.pair_bins();
            for (int t = 0; t < 200000; ++t) {
                const uint64_t j = rnd() & (R.bins() - 1);
                const uint64_t j2 = j ^ (1ull << (rnd() % (2 * k)));          // a neighbour differing in one bit
                const uint64_t rcj = RangeIndex::revcomp(j, k), rcj2 = RangeIndex::revcomp(j2, k);
                if (R.pos(j) >= n2 || RangeIndex::revcomp(rcj, k) != j) ++failures;
                if (R.owner(j) == R.owner(j2) && R.owner(rcj) == R.owner(rcj2) && R.pos(j) == R.pos(j2)) ++failures;
                (void)n1;
            }
        }
    if (failures) {
        printf("range_index_check: %d failure(s)\n", failures);
        return 1;
    }
    printf("range_index_check: %d (k, W) cases\nRANGE_INDEX_OK\n", cases);
    return 0;
}
