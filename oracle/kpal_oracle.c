/*
 * kpal_oracle.c -- CPU restatement of kPAL's k-mer counting / profile-distance hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: it restates, operation for
 * operation, the reference's pure-Python/NumPy algorithm so the HIP product path can be
 * checked bit-for-bit (integers) or to 1e-9 relative (fp64 sums).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * package (kpal_amd/) never imports, links or calls anything in oracle/.
 *
 * Parity pinning: every function here is checked against golden vectors produced by
 * importing the unmodified reference (tools/gen_golden.py -> tests/golden/), see
 * tests/test_oracle_golden.py.
 *
 * Citations are file:line into /root/reference (kPAL 2.1.2.dev).
 *
 * Build: gcc -O2 -fPIC -shared -o libkpal_oracle.so kpal_oracle.c -lm   (see Makefile)
 */
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------ *
 * a1: nucleotide -> 2-bit code.  kpal/klib.py:43-48  (A/a=0, C/c=1, G/g=2, T/t=3);
 * every other byte is outside the alphabet regex '[^AaCcGgTt]' (klib.py:152).
 * ------------------------------------------------------------------------------------ */
static inline int nucleotide_to_binary(uint8_t c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
    }
}

/* ------------------------------------------------------------------------------------ *
 * a2: Profile.from_sequences inner loops for ONE sequence.  kpal/klib.py:154-168.
 *   for part in alphabet.split(sequence):           (155)  maximal runs of AaCcGgTt
 *     if len(part) >= length:                        (156)
 *        fold the first k characters                 (157-161), counts[binary] += 1 (162)
 *        for every further char: roll + mask, += 1   (165-168)
 * counts is accumulated into (+=), so calling it once per sequence reproduces the outer
 * loop (154).  counts must hold 4^k int64.
 * ------------------------------------------------------------------------------------ */
ORACLE_API int kpal_oracle_count_sequence(const uint8_t *seq, size_t n, int k, int64_t *counts)
{
    if (k < 1 || k > 31) return -1;
    const uint64_t bitmask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1ULL);
    size_t i = 0;
    while (i < n) {
        /* skip separators */
        while (i < n && nucleotide_to_binary(seq[i]) < 0) i++;
        size_t start = i;
        while (i < n && nucleotide_to_binary(seq[i]) >= 0) i++;
        size_t len = i - start;               /* one `part` of alphabet.split */
        if (len >= (size_t)k) {
            uint64_t binary = 0;
            size_t j;
            for (j = 0; j < (size_t)k; j++)
                binary = (binary << 2) | (uint64_t)nucleotide_to_binary(seq[start + j]);
            counts[binary] += 1;
            for (; j < len; j++) {
                binary = ((binary << 2) | (uint64_t)nucleotide_to_binary(seq[start + j])) & bitmask;
                counts[binary] += 1;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ *
 * a5: Profile.reverse_complement.  kpal/klib.py:394-412: number = ~number; then k times
 * result = (result << 2) | (number & 3); number >>= 2.
 * ------------------------------------------------------------------------------------ */
ORACLE_API uint64_t kpal_oracle_reverse_complement(uint64_t number, int k)
{
    number = ~number;
    uint64_t result = 0;
    for (int i = 0; i < k; i++) {
        result = (result << 2) | (number & 3ULL);
        number >>= 2;
    }
    return result;
}

/* a6: Profile.balance, in place.  kpal/klib.py:285-298.  int64 adds wrap like NumPy. */
ORACLE_API void kpal_oracle_balance(int64_t *counts, int k)
{
    const uint64_t number = 1ULL << (2 * k);
    uint64_t *c = (uint64_t *)counts;
    for (uint64_t i = 0; i < number; i++) {
        uint64_t i_rc = kpal_oracle_reverse_complement(i, k);
        if (i < i_rc) {
            uint64_t temp = c[i];
            c[i] += c[i_rc];
            c[i_rc] += temp;
        } else if (i == i_rc) {
            c[i] += c[i];
        }
    }
}

/* The same loop with its index range dealt to T threads in blocks (test infrastructure: the single-threaded loop takes 22 s at k = 14).
 * Race-free: the pair {i, rc(i)} is read and written only by the thread whose range holds min(i, rc(i)) -- the thread that meets
 * the larger index skips it (neither i < rc(i) nor i == rc(i) holds there). */
typedef struct {
    uint64_t *c;
    uint64_t number;
    int k, T, t;
} balance_range_job;

#define BALANCE_BLOCK 4096ULL   /* indices per block; blocks are dealt round robin (the pairs with i < rc(i) crowd the low indices) */

static void *balance_range_worker(void *arg)
{
    balance_range_job *j = (balance_range_job *)arg;
    uint64_t *c = j->c;
    for (uint64_t b = (uint64_t)j->t * BALANCE_BLOCK; b < j->number; b += (uint64_t)j->T * BALANCE_BLOCK) {
        const uint64_t end = b + BALANCE_BLOCK < j->number ? b + BALANCE_BLOCK : j->number;
        for (uint64_t i = b; i < end; i++) {
            uint64_t i_rc = kpal_oracle_reverse_complement(i, j->k);
            if (i < i_rc) {
                uint64_t temp = c[i];
                c[i] += c[i_rc];
                c[i_rc] += temp;
            } else if (i == i_rc) {
                c[i] += c[i];
            }
        }
    }
    return NULL;
}

ORACLE_API void kpal_oracle_balance_mt(int64_t *counts, int k, int threads)
{
    const uint64_t number = 1ULL << (2 * k);
    int T = threads < 1 ? 1 : (threads > 256 ? 256 : threads);
    if ((uint64_t)T > number) T = (int)number;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * T);
    char *joinable = (char *)calloc(T, 1);
    balance_range_job *jobs = (balance_range_job *)malloc(sizeof(balance_range_job) * T);
    if (!th || !joinable || !jobs || T == 1) {
        kpal_oracle_balance(counts, k);
    } else {
        for (int t = 0; t < T; t++) {
            jobs[t].c = (uint64_t *)counts; jobs[t].k = k; jobs[t].number = number; jobs[t].T = T; jobs[t].t = t;
            joinable[t] = pthread_create(&th[t], NULL, balance_range_worker, &jobs[t]) == 0;
            if (!joinable[t]) balance_range_worker(&jobs[t]);
        }
        for (int t = 0; t < T; t++)
            if (joinable[t]) pthread_join(th[t], NULL);
    }
    free(th);
    free(joinable);
    free(jobs);
}

/* a7: Profile.split.  kpal/klib.py:300-327.  Returns the output length
 * ((4^k + #palindromes)/2); forward/reverse must have room for 4^k entries. */
ORACLE_API size_t kpal_oracle_split(const int64_t *counts, int k, int64_t *forward, int64_t *reverse)
{
    const uint64_t number = 1ULL << (2 * k);
    const uint64_t *c = (const uint64_t *)counts;
    size_t m = 0;
    for (uint64_t i = 0; i < number; i++) {
        uint64_t i_rc = kpal_oracle_reverse_complement(i, k);
        if (i < i_rc) {
            forward[m] = (int64_t)(c[i] * 2ULL);
            reverse[m] = (int64_t)(c[i_rc] * 2ULL);
            m++;
        } else if (i == i_rc) {
            forward[m] = counts[i];
            reverse[m] = counts[i];
            m++;
        }
    }
    return m;
}

/* ------------------------------------------------------------------------------------ *
 * a8: pairwise functions on int64.  kpal/metrics.py:159-162.
 *   prod: abs(x - y) / ((x + 1) * (y + 1))      sum: abs(x - y) / (x + y + 1)
 * NumPy evaluates numerator and denominator in int64 (silent wrap-around), then
 * true_divide converts both to float64 and divides (metrics.py:12 imports division).
 * ------------------------------------------------------------------------------------ */
static inline int64_t wrap_abs(int64_t v) { return v < 0 ? (int64_t)(0ULL - (uint64_t)v) : v; }

static inline double pairwise_prod_i64(int64_t x, int64_t y)
{
    int64_t num = wrap_abs((int64_t)((uint64_t)x - (uint64_t)y));
    int64_t den = (int64_t)(((uint64_t)x + 1ULL) * ((uint64_t)y + 1ULL));
    return (double)num / (double)den;
}

static inline double pairwise_sum_i64(int64_t x, int64_t y)
{
    int64_t num = wrap_abs((int64_t)((uint64_t)x - (uint64_t)y));
    int64_t den = (int64_t)((uint64_t)x + (uint64_t)y + 1ULL);
    return (double)num / (double)den;
}

static inline double pairwise_prod_f64(double x, double y) { return fabs(x - y) / ((x + 1.0) * (y + 1.0)); }
static inline double pairwise_sum_f64(double x, double y) { return fabs(x - y) / (x + y + 1.0); }

/* NumPy's float64 add.reduce on a contiguous array is pairwise summation
 * (numpy/core/src/umath/loops_utils.h.src, DOUBLE_pairwise_sum, block size 128, 8-way
 * unrolled).  Third-party arithmetic: NumPy is not under /root/reference (setup.py:9,
 * unpinned; goldens were generated with 1.26.4).  Restated from its published algorithm so
 * the oracle's sums track `distances.sum()` (metrics.py:123) as closely as possible. */
static double np_pairwise_sum(const double *a, size_t n)
{
    if (n < 8) {
        double res = 0.0;
        for (size_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        size_t i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8) {
            r[0] += a[i + 0]; r[1] += a[i + 1]; r[2] += a[i + 2]; r[3] += a[i + 3];
            r[4] += a[i + 4]; r[5] += a[i + 5]; r[6] += a[i + 6]; r[7] += a[i + 7];
        }
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        size_t n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
    }
}

/* ------------------------------------------------------------------------------------ *
 * a9: metrics.multiset.  kpal/metrics.py:101-123.
 *   nonzero   = np.where(np.logical_or(left, right))        (121)
 *   distances = pairwise(left[nonzero], right[nonzero])     (122)
 *   return distances.sum() / (len(distances) + 1)           (123)
 * pairwise: 0 = prod, 1 = sum.  *m_out (optional) receives len(distances).
 * ------------------------------------------------------------------------------------ */
static double multiset_i64_scratch(const int64_t *left, const int64_t *right, size_t n, int pairwise, int64_t *m_out, double *distances)
{
    size_t m = 0;
    for (size_t i = 0; i < n; i++) {
        if (left[i] != 0 || right[i] != 0)
            distances[m++] = pairwise == 0 ? pairwise_prod_i64(left[i], right[i])
                                           : pairwise_sum_i64(left[i], right[i]);
    }
    double s = np_pairwise_sum(distances, m);
    if (m_out) *m_out = (int64_t)m;
    return s / (double)(m + 1);
}

ORACLE_API double kpal_oracle_multiset_i64(const int64_t *left, const int64_t *right, size_t n,
                                           int pairwise, int64_t *m_out)
{
    double *distances = (double *)malloc((n ? n : 1) * sizeof(double));
    const double d = multiset_i64_scratch(left, right, n, pairwise, m_out, distances);
    free(distances);
    return d;
}

/* float64 inputs (profiles after do_scale, kdistlib.py:149-157) */
ORACLE_API double kpal_oracle_multiset_f64(const double *left, const double *right, size_t n,
                                           int pairwise, int64_t *m_out)
{
    double *distances = (double *)malloc((n ? n : 1) * sizeof(double));
    size_t m = 0;
    for (size_t i = 0; i < n; i++) {
        if (left[i] != 0.0 || right[i] != 0.0)
            distances[m++] = pairwise == 0 ? pairwise_prod_f64(left[i], right[i])
                                           : pairwise_sum_f64(left[i], right[i]);
    }
    double s = np_pairwise_sum(distances, m);
    free(distances);
    if (m_out) *m_out = (int64_t)m;
    return s / (double)(m + 1);
}

/* ------------------------------------------------------------------------------------ *
 * a10: metrics.euclidean = vector_length(np.subtract(left, right))  metrics.py:126-135,
 * vector_length = np.sqrt(np.dot(v, v))                             metrics.py:36-46.
 * For int64 inputs the dot is an exact int64 (wrap-around); sqrt is IEEE.
 * *dot_out (optional) receives the int64 dot product.
 * ------------------------------------------------------------------------------------ */
ORACLE_API double kpal_oracle_euclidean_i64(const int64_t *left, const int64_t *right, size_t n,
                                            int64_t *dot_out)
{
    uint64_t acc = 0;
    for (size_t i = 0; i < n; i++) {
        uint64_t d = (uint64_t)left[i] - (uint64_t)right[i];
        acc += d * d;
    }
    if (dot_out) *dot_out = (int64_t)acc;
    return sqrt((double)(int64_t)acc);
}

/* ------------------------------------------------------------------------------------ *
 * a11: ProfileDistance.distance, default + do_balance paths.  kpal/kdistlib.py:126-161.
 *   copies (136-137); if do_balance: balance both (139-141); then multiset(pairwise) or
 *   distance_function (159-161).  metric: 0 = multiset/prod, 1 = multiset/sum, 2 = euclidean.
 * Inputs are never modified (tests/test_kdistlib.py:124-135).
 * ------------------------------------------------------------------------------------ */
ORACLE_API double kpal_oracle_distance(const int64_t *left, const int64_t *right, int k,
                                       int do_balance, int metric)
{
    const size_t n = (size_t)1 << (2 * k);
    int64_t *l = (int64_t *)malloc(n * sizeof(int64_t));
    int64_t *r = (int64_t *)malloc(n * sizeof(int64_t));
    memcpy(l, left, n * sizeof(int64_t));
    memcpy(r, right, n * sizeof(int64_t));
    if (do_balance) {
        kpal_oracle_balance(l, k);
        kpal_oracle_balance(r, k);
    }
    double d = metric == 2 ? kpal_oracle_euclidean_i64(l, r, n, NULL)
                           : kpal_oracle_multiset_i64(l, r, n, metric, NULL);
    free(l);
    free(r);
    return d;
}

/* ------------------------------------------------------------------------------------ *
 * a11 with every option: ProfileDistance.distance, kpal/kdistlib.py:126-161.
 *   positive  (143-145, metrics.py:89-98)  left = left*bool(right); right = right*bool(left)
 *   smoothing (147-148, kdistlib.py:53-124) recursive collapse of sub-profiles, restated as the
 *             same top-down recursion
 *   scaling   (149-157, metrics.py:49-86)   get_scale on np.sum totals, optional scale_down,
 *             profiles become float64
 *   metric    0/1 multiset prod/sum, 2 euclidean, 3 cosine_similarity (metrics.py:138-147)
 * summary: 0 = np.min, 1 = np.mean, 2 = np.median of the four quarter sums.
 * ------------------------------------------------------------------------------------ */
ORACLE_API void kpal_oracle_positive(int64_t *left, int64_t *right, size_t n)
{
    for (size_t i = 0; i < n; i++) left[i] = right[i] != 0 ? left[i] : 0;   /* left * bool(right) */
    for (size_t i = 0; i < n; i++) right[i] = left[i] != 0 ? right[i] : 0;  /* right * bool(new left) */
}

static int cmp_i64(const void *a, const void *b)
{
    const int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return x < y ? -1 : x > y;
}

static double summary4(const int64_t *c, int summary)
{
    if (summary == 0) {                     /* np.min */
        int64_t m = c[0];
        for (int i = 1; i < 4; i++) if (c[i] < m) m = c[i];
        return (double)m;
    }
    if (summary == 1) {                     /* np.mean: float64 accumulation, then / 4 */
        double s = 0.0;
        for (int i = 0; i < 4; i++) s += (double)c[i];
        return s / 4.0;
    }
    int64_t t[4] = {c[0], c[1], c[2], c[3]};  /* np.median: mean of the two middle values */
    qsort(t, 4, sizeof(int64_t), cmp_i64);
    return ((double)t[1] + (double)t[2]) / 2.0;
}

static void collapse4(const int64_t *v, size_t start, size_t length, int64_t *out)
{   /* kdistlib.py:53-69: reshape (4, length/4), sum over axis 1 (int64, wrapping) */
    const size_t q = length / 4;
    for (int i = 0; i < 4; i++) {
        uint64_t s = 0;
        for (size_t j = 0; j < q; j++) s += (uint64_t)v[start + (size_t)i * q + j];
        out[i] = (int64_t)s;
    }
}

static void dynamic_smooth_rec(int64_t *l, int64_t *r, size_t start, size_t length, int summary, double threshold)
{   /* kdistlib.py:71-110 */
    if (length == 1) return;
    int64_t lc[4], rc[4];
    collapse4(l, start, length, lc);
    collapse4(r, start, length, rc);
    const double fl = summary4(lc, summary), fr = summary4(rc, summary);
    if ((fl < fr ? fl : fr) <= threshold) {
        l[start] = (int64_t)((uint64_t)lc[0] + (uint64_t)lc[1] + (uint64_t)lc[2] + (uint64_t)lc[3]);
        r[start] = (int64_t)((uint64_t)rc[0] + (uint64_t)rc[1] + (uint64_t)rc[2] + (uint64_t)rc[3]);
        for (size_t i = start + 1; i < start + length; i++) l[i] = r[i] = 0;
        return;
    }
    for (int i = 0; i < 4; i++) dynamic_smooth_rec(l, r, start + (size_t)i * (length / 4), length / 4, summary, threshold);
}

ORACLE_API void kpal_oracle_dynamic_smooth(int64_t *left, int64_t *right, int k, int summary, double threshold)
{
    dynamic_smooth_rec(left, right, 0, (size_t)1 << (2 * k), summary, threshold);
}

ORACLE_API double kpal_oracle_profile_distance(const int64_t *left, const int64_t *right, int k, int do_balance,
                                               int do_positive, int do_smooth, int summary, double threshold,
                                               int do_scale, int down, int metric)
{
    const size_t n = (size_t)1 << (2 * k);
    int64_t *l = (int64_t *)malloc(n * sizeof(int64_t));
    int64_t *r = (int64_t *)malloc(n * sizeof(int64_t));
    memcpy(l, left, n * sizeof(int64_t));
    memcpy(r, right, n * sizeof(int64_t));
    if (do_balance) {
        kpal_oracle_balance(l, k);
        kpal_oracle_balance(r, k);
    }
    if (do_positive) kpal_oracle_positive(l, r, n);
    if (do_smooth) kpal_oracle_dynamic_smooth(l, r, k, summary, threshold);
    double d;
    if (do_scale) {
        uint64_t tl = 0, tr = 0;
        for (size_t i = 0; i < n; i++) {
            tl += (uint64_t)l[i];
            tr += (uint64_t)r[i];
        }
        double ls = 1.0, rs = 1.0;              /* metrics.py:59-72 */
        if ((int64_t)tl < (int64_t)tr) ls = (double)(int64_t)tr / (double)(int64_t)tl;
        else rs = (double)(int64_t)tl / (double)(int64_t)tr;
        if (down) {                             /* metrics.py:84-86 */
            const double top = ls > rs ? ls : rs;
            ls /= top;
            rs /= top;
        }
        double *fl = (double *)malloc(n * sizeof(double));
        double *fr = (double *)malloc(n * sizeof(double));
        for (size_t i = 0; i < n; i++) {
            fl[i] = (double)l[i] * ls;
            fr[i] = (double)r[i] * rs;
        }
        if (metric <= 1) {
            d = kpal_oracle_multiset_f64(fl, fr, n, metric, NULL);
        } else {
            double lr = 0.0, ll = 0.0, rr = 0.0, dd = 0.0;
            for (size_t i = 0; i < n; i++) {
                lr += fl[i] * fr[i];
                ll += fl[i] * fl[i];
                rr += fr[i] * fr[i];
                dd += (fl[i] - fr[i]) * (fl[i] - fr[i]);
            }
            d = metric == 2 ? sqrt(dd) : lr / (sqrt(ll) * sqrt(rr));
        }
        free(fl);
        free(fr);
    } else if (metric <= 1) {
        d = kpal_oracle_multiset_i64(l, r, n, metric, NULL);
    } else if (metric == 2) {
        d = kpal_oracle_euclidean_i64(l, r, n, NULL);
    } else {
        uint64_t lr = 0, ll = 0, rr = 0;        /* np.dot on int64: exact, wrapping */
        for (size_t i = 0; i < n; i++) {
            lr += (uint64_t)l[i] * (uint64_t)r[i];
            ll += (uint64_t)l[i] * (uint64_t)l[i];
            rr += (uint64_t)r[i] * (uint64_t)r[i];
        }
        d = (double)(int64_t)lr / (sqrt((double)(int64_t)ll) * sqrt((double)(int64_t)rr));
    }
    free(l);
    free(r);
    return d;
}

/* a12: kdistlib.distance_matrix values.  kpal/kdistlib.py:179-186: rows i = 1..P-1,
 * columns j < i, value = dist.distance(profiles[i], profiles[j]).  out holds P(P-1)/2
 * doubles, row-major in that order.  profiles: P contiguous vectors of 4^k int64. */
ORACLE_API void kpal_oracle_distance_matrix(const int64_t *profiles, int P, int k, int do_balance,
                                            int metric, double *out)
{
    const size_t n = (size_t)1 << (2 * k);
    size_t o = 0;
    for (int i = 1; i < P; i++)
        for (int j = 0; j < i; j++)
            out[o++] = kpal_oracle_distance(profiles + (size_t)i * n, profiles + (size_t)j * n, k,
                                            do_balance, metric);
}

/* The same matrix on T threads (test infrastructure for BASELINE config 5 at its stated size: 2016 pairs of 4^12 bins are
 * ~200 s on one core).  Every profile is balanced ONCE when do_balance is set -- identical to the balanced copies
 * kdistlib.py:136-141 makes inside every pair -- and the pairs are dealt to the threads; each pair's value is computed by
 * the single-threaded functions above (same summation order). */
typedef struct {
    const int64_t *profiles;
    int P, k, metric, T, t;
    double *out;
} matrix_job;

static void *matrix_worker(void *arg)
{
    matrix_job *m = (matrix_job *)arg;
    const size_t n = (size_t)1 << (2 * m->k);
    /* kpal_oracle_distance without its copies (nothing is balanced here, the inputs are not modified) and with ONE scratch
     * array per thread: 64 threads allocating and freeing 128 MiB blocks per pair spend their time in the kernel's mmap lock */
    double *scratch = m->metric == 2 ? NULL : (double *)malloc((n ? n : 1) * sizeof(double));
    size_t o = 0;
    for (int i = 1; i < m->P; i++)
        for (int j = 0; j < i; j++, o++)
            if ((int)(o % (size_t)m->T) == m->t) {
                const int64_t *l = m->profiles + (size_t)i * n, *r = m->profiles + (size_t)j * n;
                if (m->metric == 2) m->out[o] = kpal_oracle_euclidean_i64(l, r, n, NULL);
                else if (scratch) m->out[o] = multiset_i64_scratch(l, r, n, m->metric, NULL, scratch);
                else m->out[o] = kpal_oracle_multiset_i64(l, r, n, m->metric, NULL);
            }
    free(scratch);
    return NULL;
}

typedef struct {
    int64_t *profiles;
    int P, k, T, t;
} balance_job;

static void *balance_worker(void *arg)
{
    balance_job *b = (balance_job *)arg;
    const size_t n = (size_t)1 << (2 * b->k);
    for (int p = b->t; p < b->P; p += b->T) kpal_oracle_balance(b->profiles + (size_t)p * n, b->k);
    return NULL;
}

ORACLE_API int kpal_oracle_distance_matrix_mt(const int64_t *profiles, int P, int k, int do_balance, int metric, int threads,
                                              double *out)
{
    if (P < 2) return 0;
    const size_t n = (size_t)1 << (2 * k);
    int T = threads < 1 ? 1 : (threads > 256 ? 256 : threads);
    int64_t *work = NULL;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * T);
    char *joinable = (char *)calloc(T, 1);
    matrix_job *jobs = (matrix_job *)malloc(sizeof(matrix_job) * T);
    balance_job *bjobs = (balance_job *)malloc(sizeof(balance_job) * T);
    int rc = (!th || !joinable || !jobs || !bjobs) ? -2 : 0;
    if (rc == 0 && do_balance) {
        work = (int64_t *)malloc((size_t)P * n * sizeof(int64_t));
        if (!work) rc = -2;
        else {
            memcpy(work, profiles, (size_t)P * n * sizeof(int64_t));
            for (int t = 0; t < T; t++) {
                bjobs[t].profiles = work; bjobs[t].P = P; bjobs[t].k = k; bjobs[t].T = T; bjobs[t].t = t;
                joinable[t] = pthread_create(&th[t], NULL, balance_worker, &bjobs[t]) == 0;
                if (!joinable[t]) balance_worker(&bjobs[t]);
            }
            for (int t = 0; t < T; t++)
                if (joinable[t]) pthread_join(th[t], NULL);
        }
    }
    if (rc == 0) {
        for (int t = 0; t < T; t++) {
            jobs[t].profiles = work ? work : profiles; jobs[t].P = P; jobs[t].k = k; jobs[t].metric = metric;
            jobs[t].T = T; jobs[t].t = t; jobs[t].out = out;
            joinable[t] = pthread_create(&th[t], NULL, matrix_worker, &jobs[t]) == 0;
            if (!joinable[t]) matrix_worker(&jobs[t]);
        }
        for (int t = 0; t < T; t++)
            if (joinable[t]) pthread_join(th[t], NULL);
    }
    free(work);
    free(bjobs);
    free(jobs);
    free(joinable);
    free(th);
    return rc;
}

/* a13: strand-balance score of `kpal showbalance`.  kpal/kmer.py:243-245:
 *   forward, reverse = profile.split(); metrics.multiset(forward, reverse, pairwise['prod']) */
ORACLE_API double kpal_oracle_strand_balance(const int64_t *counts, int k, int pairwise)
{
    const size_t n = (size_t)1 << (2 * k);
    int64_t *f = (int64_t *)malloc(n * sizeof(int64_t));
    int64_t *r = (int64_t *)malloc(n * sizeof(int64_t));
    size_t m = kpal_oracle_split(counts, k, f, r);
    double d = kpal_oracle_multiset_i64(f, r, m, pairwise, NULL);
    free(f);
    free(r);
    return d;
}

/* ------------------------------------------------------------------------------------ *
 * Profile summaries.  kpal/klib.py:193-225: non_zero = np.count_nonzero(counts),
 * total = counts.sum() (int64, wraps), mean = counts.mean(), median = np.median(counts),
 * std = counts.std().  NumPy (third party, see np_pairwise_sum above) computes
 *   mean : add.reduce(counts, dtype=float64) / n -- the int64 -> float64 cast runs through the
 *          ufunc buffer, 8192 elements at a time, each buffer summed pairwise and the buffer
 *          sums accumulated in order;
 *   std  : sqrt(add.reduce(((double)counts - mean)**2) / n), one pairwise sum (no cast);
 *   median: for even n the mean of the two middle elements of the sorted vector, in float64.
 * out[0..6] = total, non_zero, min, max (as int64 bit patterns in doubles is lossy, so:)
 * integers go to iout[0..3] = total, non_zero, min, max; dout[0..2] = mean, median, std.
 * ------------------------------------------------------------------------------------ */
ORACLE_API void kpal_oracle_stats(const int64_t *c, size_t n, int64_t *iout, double *dout)
{
    uint64_t total = 0;
    int64_t nz = 0, mn = INT64_MAX, mx = INT64_MIN;
    for (size_t i = 0; i < n; i++) {
        total += (uint64_t)c[i];
        nz += c[i] != 0;
        if (c[i] < mn) mn = c[i];
        if (c[i] > mx) mx = c[i];
    }
    iout[0] = (int64_t)total;
    iout[1] = nz;
    iout[2] = mn;
    iout[3] = mx;
    double *d = (double *)malloc((n ? n : 1) * sizeof(double));
    double sum = 0.0;
    for (size_t b = 0; b < n; b += 8192) {
        const size_t len = n - b < 8192 ? n - b : 8192;
        for (size_t i = 0; i < len; i++) d[i] = (double)c[b + i];
        sum += np_pairwise_sum(d, len);
    }
    const double mean = sum / (double)n;
    for (size_t i = 0; i < n; i++) {
        const double x = (double)c[i] - mean;
        d[i] = x * x;
    }
    dout[0] = mean;
    dout[2] = sqrt(np_pairwise_sum(d, n) / (double)n);
    free(d);
    int64_t *sorted = (int64_t *)malloc((n ? n : 1) * sizeof(int64_t));
    memcpy(sorted, c, n * sizeof(int64_t));
    qsort(sorted, n, sizeof(int64_t), cmp_i64);
    dout[1] = (n % 2) ? (double)sorted[n / 2] : ((double)sorted[n / 2 - 1] + (double)sorted[n / 2]) / 2.0;
    free(sorted);
}

/* metrics.mergers, kpal/metrics.py:174-179, as used by Profile.merge (klib.py:269-283):
 *   0 sum : x + y        1 xor : (x + y) * logical_xor(x, y)
 *   2 int : x * bool(y)  3 nint: x * logical_not(y)            (int64, wrapping) */
ORACLE_API void kpal_oracle_merge(const int64_t *x, const int64_t *y, size_t n, int merger, int64_t *out)
{
    for (size_t i = 0; i < n; i++) {
        const uint64_t a = (uint64_t)x[i], b = (uint64_t)y[i];
        uint64_t r;
        switch (merger) {
        case 0: r = a + b; break;
        case 1: r = (a + b) * (uint64_t)((a != 0) != (b != 0)); break;
        case 2: r = a * (uint64_t)(b != 0); break;
        default: r = a * (uint64_t)(b == 0); break;
        }
        out[i] = (int64_t)r;
    }
}

/* Profile.shrink(factor), kpal/klib.py:329-352: new_counts[j] = sum(counts[j*4^f : (j+1)*4^f]). */
ORACLE_API void kpal_oracle_shrink(const int64_t *c, int k, int factor, int64_t *out)
{
    const size_t n = (size_t)1 << (2 * k), m = (size_t)1 << (2 * factor);
    for (size_t j = 0; j < n / m; j++) {
        uint64_t s = 0;
        for (size_t i = 0; i < m; i++) s += (uint64_t)c[j * m + i];
        out[j] = (int64_t)s;
    }
}

/* ------------------------------------------------------------------------------------ *
 * Synthetic read generator (OURS, not the reference's: SURVEY.md section 8d).  Shared spec
 * for bench/test inputs; the HIP generator kernel must reproduce these bytes exactly.
 *   mix = splitmix64 finaliser; g = read*read_len + pos;
 *   base = "ACGT"[(mix(seed*0xD1342543DE82EF95 + (g>>5)) >> (2*(g&31))) & 3]
 *   layout: read r at bytes [r*(read_len+1), +read_len), then '\n'.
 * noisy != 0 (robustness variant): N iff mix(~seed + g) % 1000 == 0, lower-case iff % 100 == 1.
 * ------------------------------------------------------------------------------------ */
static inline uint64_t mix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

ORACLE_API void kpal_oracle_synth_reads(uint64_t seed, uint64_t first_read, uint64_t n_reads,
                                        int read_len, int noisy, uint8_t *out)
{
    static const char acgt[4] = {'A', 'C', 'G', 'T'};
    const uint64_t stride = (uint64_t)read_len + 1;
    for (uint64_t r = 0; r < n_reads; r++) {
        uint8_t *dst = out + r * stride;
        for (int pos = 0; pos < read_len; pos++) {
            uint64_t g = (first_read + r) * (uint64_t)read_len + (uint64_t)pos;
            uint64_t w = mix64(seed * 0xD1342543DE82EF95ULL + (g >> 5));
            uint8_t c = (uint8_t)acgt[(w >> (2 * (g & 31))) & 3];
            if (noisy) {
                uint64_t h = mix64(~seed + g);
                if (h % 1000 == 0) c = 'N';
                else if (h % 100 == 1) c = (uint8_t)(c | 0x20);
            }
            dst[pos] = c;
        }
        dst[read_len] = '\n';
    }
}

/* Multi-threaded counting for the at-scale bit-exactness check and the all-cores CPU
 * baseline: the byte stream is cut at arbitrary positions; each piece is extended to the
 * left by (k-1) bytes of halo and only k-mers ENDING inside the piece are counted, which is
 * the same set of windows as one sequential scan.  This
 * function counts one piece [begin, end) of buf (halo handled here). */
ORACLE_API int kpal_oracle_count_piece(const uint8_t *buf, size_t begin, size_t end, int k,
                                       int64_t *counts)
{
    if (k < 1 || k > 31) return -1;
    const uint64_t bitmask = (1ULL << (2 * k)) - 1ULL;
    size_t i = begin >= (size_t)(k - 1) ? begin - (size_t)(k - 1) : 0;
    uint64_t binary = 0;
    size_t run = 0;
    for (; i < end; i++) {
        int code = nucleotide_to_binary(buf[i]);
        if (code < 0) { run = 0; binary = 0; continue; }
        binary = ((binary << 2) | (uint64_t)code) & bitmask;
        run++;
        if (run >= (size_t)k && i >= begin) counts[binary] += 1;
    }
    return 0;
}

/* All-cores variant of the count (baseline + at-scale oracle): T pthreads, each scans one piece
 * of the stream with the reference's rolling window (kpal/klib.py:157-168).
 *   - small tables (4^k * 8 B <= 1 MiB, k <= 8): a private histogram per thread (stays in the
 *     core's L2), summed into `counts` at the end;
 *   - larger tables: ONE shared table, every increment a relaxed atomic add.  Replicating a
 *     128 MiB (k = 12) or 8 GiB (k = 15) table per thread makes the baseline measure calloc and
 *     the merge instead of counting; the shared table costs one locked add per k-mer, which at
 *     these sizes is a cache miss either way.
 * Integer adds: the result is identical to the sequential scan in both forms. */
typedef struct {
    const uint8_t *buf;
    size_t begin, end;
    int k;
    int64_t *hist;
    int shared;
} count_job;

static void count_piece_shared(const uint8_t *buf, size_t begin, size_t end, int k, int64_t *counts)
{
    const uint64_t bitmask = (k >= 32) ? ~(uint64_t)0 : (((uint64_t)1 << (2 * k)) - 1);
    size_t i = begin >= (size_t)(k - 1) ? begin - (size_t)(k - 1) : 0;
    uint64_t binary = 0;
    size_t run = 0;
    for (; i < end; i++) {
        int code = nucleotide_to_binary(buf[i]);
        if (code < 0) { run = 0; binary = 0; continue; }
        binary = ((binary << 2) | (uint64_t)code) & bitmask;
        run++;
        if (run >= (size_t)k && i >= begin) __atomic_fetch_add(&counts[binary], 1, __ATOMIC_RELAXED);
    }
}

static void *count_worker(void *arg)
{
    count_job *j = (count_job *)arg;
    if (j->shared) count_piece_shared(j->buf, j->begin, j->end, j->k, j->hist);
    else kpal_oracle_count_piece(j->buf, j->begin, j->end, j->k, j->hist);
    return NULL;
}

typedef struct {
    int64_t **hists;
    int64_t *counts;
    int T;
    size_t lo, hi;
} merge_job;

static void *merge_worker(void *arg)
{
    merge_job *m = (merge_job *)arg;
    for (int t = 0; t < m->T; t++)
        for (size_t i = m->lo; i < m->hi; i++) m->counts[i] += m->hists[t][i];
    return NULL;
}

/* mode 0: the rule above (private histograms for k <= 8, else one shared table); 1: one shared table; 2: a private
 * histogram per thread whatever its size (the all-cores CPU baseline at k = 12: T x 128 MiB, calloc and the merge inside
 * the caller's timed region).  The result is the same in every mode. */
ORACLE_API int kpal_oracle_count_flat_mt_mode(const uint8_t *buf, size_t n, int k, int threads, int mode, int64_t *counts)
{
    if (k < 1 || k > 31 || threads < 1 || mode < 0 || mode > 2) return -1;
    const size_t bins = (size_t)1 << (2 * k);
    const int T = threads > 256 ? 256 : threads;
    const int shared = mode == 1 || (mode == 0 && bins * sizeof(int64_t) > ((size_t)1 << 20));
    int rc = 0;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * T);
    char *joinable = (char *)calloc(T, 1);
    count_job *jobs = (count_job *)malloc(sizeof(count_job) * T);
    merge_job *merges = (merge_job *)malloc(sizeof(merge_job) * T);
    int64_t **hists = (int64_t **)calloc(T, sizeof(int64_t *));
    /* every buffer exists before the first thread starts: a failed allocation never leaves a thread running */
    if (!th || !joinable || !jobs || !merges || !hists) rc = -2;
    for (int t = 0; rc == 0 && !shared && t < T; t++) {
        hists[t] = (int64_t *)calloc(bins, sizeof(int64_t));
        if (!hists[t]) rc = -2;
    }
    if (rc == 0) {
        for (int t = 0; t < T; t++) {
            jobs[t].buf = buf;
            jobs[t].begin = n * (size_t)t / T;
            jobs[t].end = n * (size_t)(t + 1) / T;
            jobs[t].k = k;
            jobs[t].hist = shared ? counts : hists[t];
            jobs[t].shared = shared;
            joinable[t] = pthread_create(&th[t], NULL, count_worker, &jobs[t]) == 0;
            if (!joinable[t]) count_worker(&jobs[t]);   /* no thread to be had: this piece on the calling thread */
        }
        for (int t = 0; t < T; t++)
            if (joinable[t]) pthread_join(th[t], NULL);
    }
    if (rc == 0 && !shared) {
        /* private histograms (k <= 8): merged in parallel, thread t sums bins [lo, hi) of all of them */
        for (int t = 0; t < T; t++) {
            merges[t].hists = hists;
            merges[t].counts = counts;
            merges[t].T = T;
            merges[t].lo = bins * (size_t)t / T;
            merges[t].hi = bins * (size_t)(t + 1) / T;
            joinable[t] = pthread_create(&th[t], NULL, merge_worker, &merges[t]) == 0;
            if (!joinable[t]) merge_worker(&merges[t]);
        }
        for (int t = 0; t < T; t++)
            if (joinable[t]) pthread_join(th[t], NULL);
    }
    if (hists)
        for (int t = 0; t < T; t++) free(hists[t]);
    free(hists);
    free(merges);
    free(jobs);
    free(joinable);
    free(th);
    return rc;
}

ORACLE_API int kpal_oracle_count_flat_mt(const uint8_t *buf, size_t n, int k, int threads, int64_t *counts)
{
    return kpal_oracle_count_flat_mt_mode(buf, n, k, threads, 0, counts);
}

/* ------------------------------------------------------------------------------------ *
 * At-scale check of LARGE tables (k = 15: 8 GiB) without holding one on the host: the same rolling
 * window (kpal/klib.py:157-168), but only the counts of SELECTED blocks of 2^block_bits consecutive
 * table entries are kept.
 *   plain [s * 2^block_bits + j] = counts[sel[s] * 2^block_bits + j]
 *   mirror[s * 2^block_bits + j] = counts[rc(sel[s] * 2^block_bits + j)]    (rc: klib.py:394-412)
 * so that Profile.balance of the selected entries (klib.py:285-298: counts[i] + counts[rc(i)]) is
 * plain + mirror.  A k-mer x contributes to mirror iff rc(x) lies in a selected block; the top
 * 2k - block_bits bits of rc(x) are the complemented, digit-reversed LOW bits of x, so the membership
 * test is a table lookup on x's low bits and rc(x) itself is computed (by the restated
 * reverse_complement above) only for the few k-mers that pass it.
 * T threads, relaxed atomic adds into the two shared arrays (as count_flat_mt's shared table).
 * block_bits must be even (whole 2-bit digits), 2 <= block_bits <= 2k.
 * ------------------------------------------------------------------------------------ */
typedef struct {
    const uint8_t *buf;
    size_t begin, end;
    int k, block_bits;
    const int32_t *slot_of_block;    /* [2^(2k - block_bits)]: slot of a block, -1 = not selected */
    const int32_t *slot_of_lowkey;   /* [2^(2k - block_bits)]: slot of the block rc(x) falls in, by x's low bits */
    int64_t *plain, *mirror;
} blocks_job;

static void *blocks_worker(void *arg)
{
    blocks_job *j = (blocks_job *)arg;
    const int k = j->k;
    const uint64_t bitmask = (k >= 32) ? ~(uint64_t)0 : (((uint64_t)1 << (2 * k)) - 1);
    const int nb = 2 * k - j->block_bits;
    const uint64_t lowmask = ((uint64_t)1 << nb) - 1;
    const uint64_t inblock = ((uint64_t)1 << j->block_bits) - 1;
    size_t i = j->begin >= (size_t)(k - 1) ? j->begin - (size_t)(k - 1) : 0;
    uint64_t binary = 0;
    size_t run = 0;
    for (; i < j->end; i++) {
        int code = nucleotide_to_binary(j->buf[i]);
        if (code < 0) { run = 0; binary = 0; continue; }
        binary = ((binary << 2) | (uint64_t)code) & bitmask;
        run++;
        if (run >= (size_t)k && i >= j->begin) {
            const int32_t s = j->slot_of_block[binary >> j->block_bits];
            if (s >= 0) __atomic_fetch_add(&j->plain[((uint64_t)s << j->block_bits) | (binary & inblock)], 1, __ATOMIC_RELAXED);
            const int32_t m = j->slot_of_lowkey[binary & lowmask];
            if (m >= 0) {
                const uint64_t rc = kpal_oracle_reverse_complement(binary, k);
                __atomic_fetch_add(&j->mirror[((uint64_t)m << j->block_bits) | (rc & inblock)], 1, __ATOMIC_RELAXED);
            }
        }
    }
    return NULL;
}

ORACLE_API int kpal_oracle_count_blocks_mt(const uint8_t *buf, size_t n, int k, int block_bits, const uint64_t *sel, int nsel,
                                           int threads, int64_t *plain, int64_t *mirror)
{
    if (k < 1 || k > 31 || threads < 1 || nsel < 1 || block_bits < 2 || block_bits > 2 * k || (block_bits & 1)) return -1;
    const int nb = 2 * k - block_bits;
    if (nb > 30) return -1;
    const size_t nblocks = (size_t)1 << nb;
    const int T = threads > 256 ? 256 : threads;
    int rc = 0;
    int32_t *slot_of_block = (int32_t *)malloc(nblocks * sizeof(int32_t));
    int32_t *slot_of_lowkey = (int32_t *)malloc(nblocks * sizeof(int32_t));
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * T);
    char *joinable = (char *)calloc(T, 1);
    blocks_job *jobs = (blocks_job *)malloc(sizeof(blocks_job) * T);
    if (!slot_of_block || !slot_of_lowkey || !th || !joinable || !jobs) rc = -2;
    if (rc == 0) {
        for (size_t b = 0; b < nblocks; b++) slot_of_block[b] = slot_of_lowkey[b] = -1;
        for (int s = 0; s < nsel && rc == 0; s++) {
            if (sel[s] >= nblocks || slot_of_block[sel[s]] >= 0) { rc = -1; break; }   /* out of range / listed twice */
            slot_of_block[sel[s]] = s;
            /* x with rc(x) in block sel[s]: the low nb bits of x are the reverse complement of the block number,
             * read as a k-mer of nb/2 digits */
            slot_of_lowkey[nb ? kpal_oracle_reverse_complement(sel[s], nb / 2) : 0] = s;
        }
    }
    if (rc == 0) {
        for (int t = 0; t < T; t++) {
            jobs[t].buf = buf;
            jobs[t].begin = n * (size_t)t / T;
            jobs[t].end = n * (size_t)(t + 1) / T;
            jobs[t].k = k;
            jobs[t].block_bits = block_bits;
            jobs[t].slot_of_block = slot_of_block;
            jobs[t].slot_of_lowkey = slot_of_lowkey;
            jobs[t].plain = plain;
            jobs[t].mirror = mirror;
            joinable[t] = pthread_create(&th[t], NULL, blocks_worker, &jobs[t]) == 0;
            if (!joinable[t]) blocks_worker(&jobs[t]);
        }
        for (int t = 0; t < T; t++)
            if (joinable[t]) pthread_join(th[t], NULL);
    }
    free(jobs);
    free(joinable);
    free(th);
    free(slot_of_lowkey);
    free(slot_of_block);
    return rc;
}
