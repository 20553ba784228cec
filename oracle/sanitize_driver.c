/* sanitize_driver.c -- runs the CPU oracle (kpal_oracle.c) under AddressSanitizer + UBSan or ThreadSanitizer
 * (oracle/Makefile: sanitize_asan, sanitize_tsan; tests/test_oracle_golden.py::test_oracle_sanitized).
 * Test infrastructure, like the oracle itself: exercises the single-thread count, the N-thread count in both of its
 * modes (private histograms + parallel merge for k <= 8, one shared table with relaxed atomic adds above), balance, split,
 * strand balance, the distances and the summaries on synthetic reads (kpal_oracle_synth_reads) and checks the invariants
 * a wrong result would break; a sanitizer finding aborts with a non-zero status. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int kpal_oracle_count_piece(const uint8_t *buf, size_t begin, size_t end, int k, int64_t *counts);
int kpal_oracle_count_flat_mt(const uint8_t *buf, size_t n, int k, int threads, int64_t *counts);
void kpal_oracle_synth_reads(uint64_t seed, uint64_t first_read, uint64_t n_reads, int read_len, int noisy, uint8_t *out);
void kpal_oracle_balance(int64_t *counts, int k);
size_t kpal_oracle_split(const int64_t *counts, int k, int64_t *forward, int64_t *reverse);
double kpal_oracle_strand_balance(const int64_t *counts, int k, int pairwise);
double kpal_oracle_distance(const int64_t *left, const int64_t *right, int k, int do_balance, int metric);
uint64_t kpal_oracle_reverse_complement(uint64_t number, int k);

static int fail(const char *what, int k)
{
    fprintf(stderr, "sanitize_driver: %s (k=%d)\n", what, k);
    return 1;
}

int main(void)
{
    const int read_len = 150;
    const uint64_t n_reads = 3000;
    const size_t nbytes = (size_t)n_reads * (read_len + 1);
    uint8_t *a = (uint8_t *)malloc(nbytes), *b = (uint8_t *)malloc(nbytes);
    if (!a || !b) return 2;
    kpal_oracle_synth_reads(7, 0, n_reads, read_len, 1, a);
    kpal_oracle_synth_reads(8, 100, n_reads, read_len, 1, b);
    static const int ks[] = {1, 6, 8, 9, 11};
    for (size_t ki = 0; ki < sizeof(ks) / sizeof(ks[0]); ki++) {
        const int k = ks[ki];
        const size_t bins = (size_t)1 << (2 * k);
        int64_t *c1 = (int64_t *)calloc(bins, 8), *cn = (int64_t *)calloc(bins, 8), *c3 = (int64_t *)calloc(bins, 8);
        int64_t *cb = (int64_t *)calloc(bins, 8), *fw = (int64_t *)calloc(bins, 8), *rv = (int64_t *)calloc(bins, 8);
        if (!c1 || !cn || !c3 || !cb || !fw || !rv) return 2;
        if (kpal_oracle_count_piece(a, 0, nbytes, k, c1)) return fail("count_piece", k);
        if (kpal_oracle_count_flat_mt(a, nbytes, k, 8, cn)) return fail("count_flat_mt", k);
        if (memcmp(c1, cn, bins * 8)) return fail("8-thread count differs from the sequential count", k);
        if (kpal_oracle_count_flat_mt(a, nbytes, k, 3, c3)) return fail("count_flat_mt", k);
        if (memcmp(c1, c3, bins * 8)) return fail("3-thread count differs from the sequential count", k);
        /* pieces with halo == one scan */
        memset(c3, 0, bins * 8);
        const size_t cut = nbytes / 3 + 5;
        kpal_oracle_count_piece(a, 0, cut, k, c3);
        kpal_oracle_count_piece(a, cut, nbytes, k, c3);
        if (memcmp(c1, c3, bins * 8)) return fail("two pieces differ from one scan", k);
        int64_t total = 0, btotal = 0;
        memcpy(cb, c1, bins * 8);
        kpal_oracle_balance(cb, k);
        for (size_t i = 0; i < bins; i++) {
            total += c1[i];
            btotal += cb[i];
            if (cb[i] != cb[kpal_oracle_reverse_complement(i, k)]) return fail("balanced table is not symmetric", k);
        }
        if (btotal != 2 * total) return fail("balance does not double the total", k);
        const size_t m = kpal_oracle_split(c1, k, fw, rv);
        if (m == 0 || m > bins) return fail("split length", k);
        const double sb = kpal_oracle_strand_balance(c1, k, 0);
        if (!(sb >= 0.0)) return fail("strand balance", k);
        memset(cn, 0, bins * 8);
        kpal_oracle_count_flat_mt(b, nbytes, k, 4, cn);
        for (int metric = 0; metric < 3; metric++)
            for (int bal = 0; bal < 2; bal++) {
                const double d = kpal_oracle_distance(c1, cn, k, bal, metric);
                if (!(d >= 0.0)) return fail("distance", k);
            }
        free(c1); free(cn); free(c3); free(cb); free(fw); free(rv);
    }
    free(a);
    free(b);
    puts("SANITIZE_OK");
    return 0;
}
