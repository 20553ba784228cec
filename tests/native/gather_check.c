/* gather_check.c -- AddressSanitizer / ThreadSanitizer harness of kpal_amd/csrc/kpal_gather_core.h (the gatherer behind
 * Profile.from_sequences: the walk over the items AND the copies on several threads): item lists with items the gatherer does not
 * read at every position, zero-length items, one giant item, buffers of every size around the stream (each run's output buffer is
 * exactly `capacity` bytes: a byte written past it is an ASan finding), windows that end inside the list, more threads than items.
 * Every call is compared with a plain serial restatement of the contract (next item, bytes, status, the bytes themselves).
 * Test infrastructure; run by tests/test_native_sanitized.py. */
#include <stdio.h>
#include <stdlib.h>

#include "../../kpal_amd/csrc/kpal_gather_core.h"

static uint64_t rng_state = 0xD1B54A32D192ED03ull;
static uint64_t rnd(void)
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

typedef struct {
    char *data;         /* exact-size allocation: an over-read is a finding */
    uint64_t len;
    int readable;
} item_t;

static int describe(void *ctx, size_t i, const char **ptr, uint64_t *len)
{
    const item_t *it = (const item_t *)ctx + i;
    if (!it->readable) return 0;
    *ptr = it->data;
    *len = it->len;
    return 1;
}

/* the contract, serially: items [first, first + n) fit `capacity`, status as kpal_gather_result */
static void reference(const item_t *items, size_t first, size_t count, char *dst, uint64_t capacity, size_t *n, uint64_t *bytes, int *status)
{
    *n = 0; *bytes = 0; *status = 0;
    for (size_t i = 0; i < count; i++) {
        const item_t *it = items + first + i;
        if (!it->readable) { *status = 2; return; }
        if (*bytes + it->len + 1 > capacity) { *status = 1; return; }
        memcpy(dst + *bytes, it->data, it->len);
        dst[*bytes + it->len] = '\n';
        *bytes += it->len + 1;
        (*n)++;
    }
}

static int check_list(size_t count, const uint32_t *lens, const char *readable, uint64_t capacity, int threads)
{
    item_t *items = (item_t *)calloc(count ? count : 1, sizeof(item_t));
    for (size_t i = 0; i < count; i++) {
        items[i].len = lens[i];
        items[i].readable = readable[i];
        items[i].data = (char *)malloc(lens[i] ? lens[i] : 1);
        for (uint32_t j = 0; j < lens[i]; j++) items[i].data[j] = "ACGTN"[rnd() % 5];
    }
    char *got = (char *)malloc(capacity ? capacity : 1), *want = (char *)malloc(capacity ? capacity : 1);
    int bad = 0;
    size_t first = 0;
    /* the caller's loop: call, skip the item the caller would take itself, call again */
    for (int guard = 0; guard < 100000 && first <= count && !bad; guard++) {
        size_t n;
        uint64_t bytes;
        int status;
        reference(items, first, count - first, want, capacity, &n, &bytes, &status);
        const kpal_gather_result r = kpal_gather_run(describe, items, first, count - first, got, capacity, threads);
        if (r.n != n || r.bytes != bytes || r.status != status || memcmp(got, want, bytes) != 0) bad = 1;
        if (status == 0) break;
        first += n + ((status == 2 || n == 0) ? 1 : 0);
    }
    for (size_t i = 0; i < count; i++) free(items[i].data);
    free(items); free(got); free(want);
    return bad;
}

int main(void)
{
    int failures = 0, cases = 0;
    {   /* hand-made shapes */
        uint32_t giant[] = {5, 0, 3000000, 0, 7};
        char all[] = {1, 1, 1, 1, 1}, mid_bad[] = {1, 1, 0, 1, 1}, first_bad[] = {0, 1, 1, 1, 1}, last_bad[] = {1, 1, 1, 1, 0};
        failures += check_list(5, giant, all, 3000017, 8); cases++;            /* exactly the stream */
        failures += check_list(5, giant, all, 3000016, 8); cases++;            /* one byte short: the last item does not fit */
        failures += check_list(5, giant, all, 100, 3); cases++;                /* the giant never fits: the caller takes it */
        failures += check_list(5, giant, mid_bad, 1 << 20, 4); cases++;
        failures += check_list(5, giant, first_bad, 1 << 20, 4); cases++;
        failures += check_list(5, giant, last_bad, 1 << 22, 4); cases++;
        failures += check_list(0, giant, all, 10, 4); cases++;
        uint32_t empties[] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        char nine[] = {1, 1, 1, 1, 1, 1, 1, 1, 1};
        failures += check_list(9, empties, nine, 9, 64); cases++;
        failures += check_list(9, empties, nine, 4, 2); cases++;
    }
    /* random lists: read-like with a few ragged and a few unreadable items; capacities from a few items to everything; the threaded
     * walk needs windows of 4096 items and more */
    for (int round = 0; round < 24; round++) {
        const size_t count = (size_t)(rnd() % (round < 6 ? 40 : 40000)) + 1;
        uint32_t *lens = (uint32_t *)malloc(sizeof(uint32_t) * count);
        char *readable = (char *)malloc(count);
        uint64_t total = 0;
        for (size_t i = 0; i < count; i++) {
            const uint64_t r = rnd() % 1000;
            lens[i] = r < 30 ? 0 : (r < 960 ? 150 : (uint32_t)(rnd() % 9000));
            readable[i] = (rnd() % (round % 3 == 0 ? 50 : 20000)) != 0;
            total += lens[i] + 1;
        }
        const uint64_t caps[] = {total, total / 3 + 1, 151 * 5000, 4096, 151};
        const int threads = (int)(rnd() % 40) - 2;                 /* below 1 as well */
        failures += check_list(count, lens, readable, caps[round % 5], threads); cases++;
        free(lens); free(readable);
    }
    if (failures) {
        printf("gather_check: %d of %d cases FAILED\n", failures, cases);
        return 1;
    }
    printf("gather_check: %d cases\nSANITIZE_OK\n", cases);
    return 0;
}
