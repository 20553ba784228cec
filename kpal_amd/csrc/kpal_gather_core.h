/* kpal_gather_core.h -- the gatherer of Profile.from_sequences, free of Python.h: `count` items, each described by
 * a callback (data pointer + length, or "not mine"), become one flat byte stream, every item followed by '\n' -- with BOTH phases on
 * several threads.  kpal_gather.c binds it to CPython lists; tests/native/gather_check.c drives it under AddressSanitizer /
 * ThreadSanitizer (pytest -m "not gpu").
 *
 * Why both.  The first gatherer (rounds 3-5, csrc/kpal_join.c in the history) walked the list on ONE thread (pointer, length, offset
 * of every item) and only copied on several: for 150-byte reads the walk IS the cost -- every item is its own heap object, one cache
 * miss each, 20-50 ns -- and the copy threads waited for it (the same 60 ns per item with 1 or 8 threads on the development
 * container; 8 M str objects: 157 -> 85.5 ms with the walk on sixteen threads, profiles/r5/seqbench.log).  So the walk is cut over the
 * threads too:
 *   pass 1 (parallel)  describe(i) -> ptr[i], len[i]; every slice remembers its first item that cannot be described / is too long;
 *   serial             lengths -> offsets (sequential arrays: ~1 ns per item), the cut at the buffer's capacity or the first bad item;
 *   pass 2 (parallel)  the copies (kpal_join_core.h: equal byte shares).
 * The list is taken in windows sized from the first items' mean length, so that a call walks little more than what fits the buffer.
 * The callback must be safe to call from several threads at once for different i (kpal_gather.c: the caller keeps the GIL, the
 * objects are only read).
 */
#ifndef KPAL_GATHER_CORE_H
#define KPAL_GATHER_CORE_H
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "kpal_join_core.h"

/* 1: item i is *len bytes at *ptr; 0: not an item this gatherer reads */
typedef int (*kpal_describe_fn)(void *ctx, size_t i, const char **ptr, uint64_t *len);

typedef struct {
    size_t n;          /* items written: [first, first + n) */
    uint64_t bytes;    /* bytes written (every item + its '\n') */
    int status;        /* 0: all `count` items written; 1: the next item does not fit; 2: the next item cannot be described;
                          -1: out of memory (nothing written) */
} kpal_gather_result;

typedef struct {
    kpal_describe_fn fn;
    void *ctx;
    size_t first, begin, end;      /* items first + [begin, end) of the window */
    const char **ptr;
    uint32_t *len;
    size_t bad;                    /* first index of the slice that is not an item (or is >= 4 GiB), or `end` */
} kpal_walk_job;

static void *kpal_walk_worker(void *arg)
{
    kpal_walk_job *j = (kpal_walk_job *)arg;
    j->bad = j->end;
    for (size_t i = j->begin; i < j->end; i++) {
        uint64_t l = 0;
        if (!j->fn(j->ctx, j->first + i, &j->ptr[i], &l) || l >= 0xFFFFFFFFull) {
            j->bad = i;
            break;
        }
        j->len[i] = (uint32_t)l;
    }
    return NULL;
}

/* pass 1 over window items [0, w): returns the number of leading items that are described (w if all are) */
static size_t kpal_gather_walk(kpal_describe_fn fn, void *ctx, size_t first, size_t w, const char **ptr, uint32_t *len, int threads)
{
    int T = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
    if (w < 4096) T = 1;
    pthread_t th[64];
    kpal_walk_job jobs[64];
    char joinable[64];
    for (int t = 0; t < T; t++) {
        jobs[t].fn = fn; jobs[t].ctx = ctx; jobs[t].first = first; jobs[t].ptr = ptr; jobs[t].len = len;
        jobs[t].begin = w / (size_t)T * (size_t)t;
        jobs[t].end = t == T - 1 ? w : w / (size_t)T * (size_t)(t + 1);
        joinable[t] = 0;
        if (t > 0) joinable[t] = pthread_create(&th[t], NULL, kpal_walk_worker, &jobs[t]) == 0;
        if (t > 0 && !joinable[t]) kpal_walk_worker(&jobs[t]);
    }
    kpal_walk_worker(&jobs[0]);
    for (int t = 1; t < T; t++)
        if (joinable[t]) pthread_join(th[t], NULL);
    for (int t = 0; t < T; t++)
        if (jobs[t].bad < jobs[t].end) return jobs[t].bad;      /* (slices are in order: the first bad item of the window) */
    return w;
}

static kpal_gather_result kpal_gather_run(kpal_describe_fn fn, void *ctx, size_t first, size_t count, char *dst, uint64_t capacity, int threads)
{
    kpal_gather_result r = {0, 0, 0};
    size_t cap_items = 0;
    const char **ptr = NULL;
    uint32_t *len = NULL;
    uint64_t *off = NULL;
    while (r.n < count) {
        /* the window: what fits the rest of the buffer by the mean length of the next few items, and a little more */
        const size_t left = count - r.n;
        uint64_t sample = 0;
        size_t ns = left < 64 ? left : 64, described = 0;
        for (size_t i = 0; i < ns; i++) {
            const char *p;
            uint64_t l;
            if (!fn(ctx, first + r.n + i, &p, &l)) break;
            sample += l + 1;
            described++;
        }
        if (described == 0) {
            r.status = 2;
            break;
        }
        const uint64_t mean = sample / described ? sample / described : 1;
        const uint64_t room = capacity - r.bytes;
        uint64_t want = room / mean + room / mean / 16 + 1024;
        size_t w = want < (uint64_t)left ? (size_t)want : left;
        if (w > cap_items) {
            free(ptr); free(len); free(off);
            cap_items = w;
            ptr = (const char **)malloc(sizeof(char *) * cap_items);
            len = (uint32_t *)malloc(sizeof(uint32_t) * cap_items);
            off = (uint64_t *)malloc(sizeof(uint64_t) * cap_items);
            if (!ptr || !len || !off) {
                free(ptr); free(len); free(off);
                if (r.n == 0) r.status = -1;
                else r.status = 1;                      /* what was written stands: the caller hands the buffer over and calls again */
                return r;
            }
        }
        const size_t good = kpal_gather_walk(fn, ctx, first + r.n, w, ptr, len, threads);
        /* offsets, and the cut at the capacity */
        uint64_t at = 0;
        size_t n = 0;
        for (; n < good; n++) {
            if (at + (uint64_t)len[n] + 1 > room) break;
            off[n] = at;
            at += (uint64_t)len[n] + 1;
        }
        kpal_join_copy(ptr, len, off, n, at, dst + r.bytes, threads, (uint64_t)4 << 20);
        r.n += n;
        r.bytes += at;
        if (n < good) {                 /* the buffer is full */
            r.status = 1;
            break;
        }
        if (good < w) {                 /* an item this gatherer does not read -- or one of 4 GiB and more: that one "does not fit" */
            const char *p;
            uint64_t l;
            r.status = fn(ctx, first + r.n, &p, &l) ? 1 : 2;
            break;
        }
    }
    free(ptr); free(len); free(off);
    return r;
}
#endif
