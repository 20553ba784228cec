/* kpal_join_core.h -- the copy phase of kpal_join.c, free of Python.h: n payloads (pointer, length, output offset) into one
 * buffer, each followed by '\n', on several threads.  kpal_join.c includes it for the CPython extension,
 * tests/native/join_check.c for the AddressSanitizer / ThreadSanitizer harness (pytest -m "not gpu"). */
#ifndef KPAL_JOIN_CORE_H
#define KPAL_JOIN_CORE_H
#include <pthread.h>
#include <stdint.h>
#include <string.h>

typedef struct {
    const char **ptr;
    const uint32_t *len;
    const uint64_t *off;
    char *dst;
    size_t begin, end;
} kpal_copy_job;

static void *kpal_copy_worker(void *arg)
{
    kpal_copy_job *j = (kpal_copy_job *)arg;
    for (size_t i = j->begin; i < j->end; i++) {
        char *d = j->dst + j->off[i];
        memcpy(d, j->ptr[i], j->len[i]);
        d[j->len[i]] = '\n';
    }
    return NULL;
}

/* Item i (len[i] bytes at ptr[i]) goes to dst + off[i], followed by '\n'; off is ascending, `total` = off[n-1] + len[n-1] + 1.
 * Equal BYTE shares: the items of thread t are those whose offset falls into its share (a share may be empty; one giant item
 * is one thread's).  Streams below `single_below` bytes are copied by the caller alone.  A thread that cannot be created has
 * its share copied by the caller. */
static void kpal_join_copy(const char **ptr, const uint32_t *len, const uint64_t *off, size_t n, uint64_t total, char *dst, int threads,
                           uint64_t single_below)
{
    if (n == 0) return;
    int T = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
    if (total < single_below) T = 1;
    pthread_t th[64];
    kpal_copy_job jobs[64];
    char joinable[64];
    size_t next = 0;
    for (int t = 0; t < T; t++) {
        const uint64_t limit = total / (uint64_t)T * (uint64_t)(t + 1);
        size_t e = next;
        if (t == T - 1) e = n;
        else while (e < n && off[e] < limit) e++;
        jobs[t].ptr = ptr; jobs[t].len = len; jobs[t].off = off; jobs[t].dst = dst;
        jobs[t].begin = next; jobs[t].end = e;
        next = e;
        joinable[t] = 0;
        if (t > 0 && jobs[t].end > jobs[t].begin) joinable[t] = pthread_create(&th[t], NULL, kpal_copy_worker, &jobs[t]) == 0;
        if (t > 0 && !joinable[t]) kpal_copy_worker(&jobs[t]);
    }
    kpal_copy_worker(&jobs[0]);
    for (int t = 1; t < T; t++)
        if (joinable[t]) pthread_join(th[t], NULL);
}
#endif
