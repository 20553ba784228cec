// hex_index_check.cpp -- CPU emulation of the HEX pipeline's index arithmetic (kpal_amd/csrc/hex_index.hpp, the functions the
// kernels of hex_kernels.hpp call): a random byte stream (reads with line ends, N, lower case) is cut into lanes of 48 bytes,
// every lane's eight groups are extracted (group_x / emit48 / group_mask), split into full and half items, and every item is
// decoded back form by form (unpack / local_of / bin_entry) into a table that must equal the plain rolling-window count
// (kpal/klib.py:157-168).  Also: every bin of every (row, form) maps to a distinct table entry, all 6 x 2048 x 8192 entries of a
// form cover the table once.  Test infrastructure; run by tests/test_abi_and_host.py::test_hex_index_arithmetic.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../kpal_amd/csrc/hex_index.hpp"

using kpal::HexIndex;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

static int code_of(uint8_t c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
    }
}

int main()
{
    constexpr int K = 12;
    const size_t bins = (size_t)1 << (2 * K);
    int failures = 0;
    // ---- 1. every (row, form, bin) is a distinct entry; a form covers the table exactly once
    {
        std::vector<uint8_t> seen(bins);
        for (int i = 0; i < HexIndex::kForms; ++i) {
            memset(seen.data(), 0, bins);
            for (uint32_t row = 0; row < (uint32_t)HexIndex::kRows; ++row)
                for (uint32_t local = 0; local < (uint32_t)HexIndex::kFormBins; ++local) {
                    const uint32_t e = HexIndex::bin_entry(row, i, local);
                    if (e >= bins || seen[e]++) ++failures;
                }
        }
        if (failures) printf("bin_entry is not a bijection per form (%d)\n", failures);
    }
    // ---- 2. stream -> lanes -> groups -> items -> forms -> table  ==  rolling window
    for (int round = 0; round < 12 && !failures; ++round) {
        const size_t lanes = 3000 + (size_t)(rnd() % 2000);
        const size_t n = lanes * 48;
        std::vector<uint8_t> buf(n);
        const int read_len = round < 4 ? 150 : (int)(rnd() % 300) + 1;
        for (size_t i = 0; i < n; ++i) {
            const uint64_t r = rnd();
            uint8_t c = "ACGT"[r & 3];
            if (round == 2) c = 'A';                                   // homopolymer: x = 0, the all-zero payload
            if ((r >> 8) % 97 == 0) c = (uint8_t)(c | 0x20);           // lower case counts
            if ((r >> 20) % (round % 3 == 0 ? 1000 : 53) == 0) c = 'N';
            if (i % (size_t)(read_len + 1) == (size_t)read_len) c = '\n';
            buf[i] = c;
        }
        // the reference count (klib.py:157-168)
        std::vector<uint32_t> want(bins), got(bins);
        {
            uint64_t binary = 0;
            size_t run = 0;
            for (size_t i = 0; i < n; ++i) {
                const int c = code_of(buf[i]);
                if (c < 0) { run = 0; binary = 0; continue; }
                binary = ((binary << 2) | (uint64_t)c) & (bins - 1);
                if (++run >= (size_t)K) want[binary]++;
            }
        }
        // the pipeline's arithmetic
        size_t items = 0, halves = 0, groups = 0;
        uint32_t pc = 0, pb = 0xFFFF;                                  // left of the stream: nothing
        for (size_t l = 0; l < lanes; ++l) {
            uint32_t c[3] = {0, 0, 0}, b[3] = {0, 0, 0};
            for (int ch = 0; ch < 3; ++ch)
                for (int j = 0; j < 16; ++j) {
                    const int code = code_of(buf[l * 48 + (size_t)ch * 16 + (size_t)j]);
                    // the device encoder gives a flagged byte SOME code: any value must do
                    c[ch] = (c[ch] << 2) | (uint32_t)(code < 0 ? (int)(rnd() & 3) : code);
                    b[ch] = (b[ch] << 1) | (code < 0 ? 1u : 0u);
                }
            const uint64_t emit = HexIndex::emit48(pb, b[0], b[1], b[2]);
            for (int q = 0; q < 8; ++q) {
                const uint64_t x = HexIndex::group_x(pc, c[0], c[1], c[2], q);
                const uint32_t m6 = HexIndex::group_mask(emit, q);
                uint32_t row, a, bb;
                HexIndex::split(x, m6, row, a, bb);
                ++groups;
                const uint32_t it[2] = {a, bb};
                uint32_t back = 0;
                for (int h = 0; h < 2; ++h) {
                    if (!it[h]) continue;
                    if (it[h] >> 24) ++failures;                       // 24 bits
                    ++items;
                    if (!(it[h] & HexIndex::kFull)) ++halves;
                    uint32_t p23, mm;
                    HexIndex::unpack(it[h], p23, mm);
                    if (back & mm) ++failures;                         // no k-mer twice
                    back |= mm;
                    for (int i = 0; i < 6; ++i)
                        if ((mm >> (5 - i)) & 1u) got[HexIndex::kmer_of(row, p23, i)]++;
                }
                if (back != m6) ++failures;                            // every counting k-mer in exactly one item
            }
            pc = c[2];
            pb = b[2];
        }
        if (memcmp(want.data(), got.data(), bins * sizeof(uint32_t)) != 0) {
            ++failures;
            printf("round %d: the table built from the items differs from the rolling-window count\n", round);
        }
        if (round < 4) printf("round %d: %zu groups -> %zu items (%.3f per group), %zu of them half items\n", round, groups, items, (double)items / (double)groups, halves);
    }
    if (failures) {
        printf("hex_index_check: %d failure(s)\n", failures);
        return 1;
    }
    printf("HEX_INDEX_OK\n");
    return 0;
}
