/* join_check.c -- AddressSanitizer / ThreadSanitizer harness of kpal_join_core.h (the copy phase behind Profile.from_sequences'
 * CPython gatherer): the byte-share split with empty shares, one giant item, single items, zero-length items, more threads than
 * items, and a buffer whose capacity is exactly the stream (a byte written past it is an ASan finding).  Test infrastructure;
 * run by tests/test_native_sanitized.py. */
#include <stdio.h>
#include <stdlib.h>

#include "../../kpal_amd/csrc/kpal_join_core.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd(void)
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

static int check_case(size_t n, const uint32_t *lens, int threads, uint64_t single_below)
{
    const char **ptr = (const char **)malloc(sizeof(char *) * (n ? n : 1));
    uint32_t *len = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint64_t *off = (uint64_t *)malloc(sizeof(uint64_t) * (n ? n : 1));
    char **items = (char **)malloc(sizeof(char *) * (n ? n : 1));
    uint64_t at = 0;
    for (size_t i = 0; i < n; i++) {
        items[i] = (char *)malloc(lens[i] ? lens[i] : 1);        /* exact-size allocations: an over-read is a finding */
        for (uint32_t j = 0; j < lens[i]; j++) items[i][j] = "ACGT"[rnd() & 3];
        ptr[i] = items[i];
        len[i] = lens[i];
        off[i] = at;
        at += (uint64_t)lens[i] + 1;
    }
    char *dst = (char *)malloc(at ? at : 1);                       /* capacity == stream length exactly */
    kpal_join_copy(ptr, len, off, n, at, dst, threads, single_below);
    int bad = 0;
    uint64_t p = 0;
    for (size_t i = 0; i < n && !bad; i++) {
        if (memcmp(dst + p, items[i], lens[i]) != 0 || dst[p + lens[i]] != '\n') bad = 1;
        p += (uint64_t)lens[i] + 1;
    }
    if (p != at) bad = 1;
    for (size_t i = 0; i < n; i++) free(items[i]);
    free(items); free(dst); free(ptr); free(len); free(off);
    return bad;
}

int main(void)
{
    int failures = 0, cases = 0;
    /* hand-made shapes */
    {
        uint32_t one_giant[] = {5, 0, 3000000, 0, 7};              /* one item holds nearly every byte: most shares are empty */
        failures += check_case(5, one_giant, 8, 0); cases++;
        uint32_t single[] = {1};
        failures += check_case(1, single, 64, 0); cases++;
        uint32_t empties[] = {0, 0, 0, 0, 0, 0, 0, 0, 0};            /* only separators */
        failures += check_case(9, empties, 4, 0); cases++;
        failures += check_case(0, empties, 4, 0); cases++;         /* nothing at all */
        uint32_t giant_first[] = {1000000, 1, 1, 1};
        failures += check_case(4, giant_first, 3, 0); cases++;
        uint32_t giant_last[] = {1, 1, 1, 1000000};
        failures += check_case(4, giant_last, 3, 0); cases++;
    }
    /* random shapes: read-like, ragged, fewer items than threads; threaded (single_below 0) and the 4 MiB rule of the product */
    for (int round = 0; round < 60; round++) {
        const size_t n = (size_t)(rnd() % (round < 20 ? 6 : 5000)) + 1;
        uint32_t *lens = (uint32_t *)malloc(sizeof(uint32_t) * n);
        for (size_t i = 0; i < n; i++) {
            const uint64_t r = rnd() % 100;
            lens[i] = r < 5 ? 0 : (r < 90 ? 150 : (uint32_t)(rnd() % 20000));
        }
        const int threads = (int)(rnd() % 70) - 2;                 /* below 1 and above 64 as well */
        failures += check_case(n, lens, threads, (round & 1) ? 0 : ((uint64_t)4 << 20)); cases++;
        free(lens);
    }
    if (failures) {
        printf("join_check: %d of %d cases FAILED\n", failures, cases);
        return 1;
    }
    printf("join_check: %d cases\nSANITIZE_OK\n", cases);
    return 0;
}
