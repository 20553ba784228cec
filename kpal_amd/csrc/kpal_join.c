/* kpal_join.c -- CPython extension `kpal_amd._kpal_join`: the host side of Profile.from_sequences for LISTS of short
 * sequences (a million 150-base reads as bytes / str objects).
 *
 * kpal/klib.py:154 walks the sequences in the interpreter; the drop-in hands the GPU one flat byte stream, sequences
 * separated by '\n' (kpal_amd/klib.py).  Building that stream with b'\n'.join costs ~100 ns per object on the interpreter's
 * one core -- 1.6 Gbases/s for 150-base reads, 25 times below what the link takes.  gather() does it in two phases:
 *   1. one pass over the list -- data pointer, length and output offset of every item (bytes, bytearray, and str whose
 *      characters all fit one byte: ASCII / latin-1, whose storage IS its latin-1 encoding);
 *   2. the payloads are copied into the caller's buffer (each followed by '\n') by several threads.
 * The calling thread keeps the GIL throughout (the copy threads touch raw memory only): no interpreter code can free or
 * change an item meanwhile, so the items need no INCREF / DECREF -- which would write to a million object headers twice.
 * Items it does not understand (str with characters above U+00FF, memoryview, anything else) end the call early: the
 * caller encodes that one item itself (klib._encode) and calls again.  No HIP, no link to libkpal_hip.so: the buffer it
 * fills is host memory the caller got from kpal_host_alloc (pinned) or NumPy.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    const char **ptr;
    const uint32_t *len;
    const uint64_t *off;
    char *dst;
    size_t begin, end;
} copy_job;

static void *copy_worker(void *arg)
{
    copy_job *j = (copy_job *)arg;
    for (size_t i = j->begin; i < j->end; i++) {
        char *d = j->dst + j->off[i];
        memcpy(d, j->ptr[i], j->len[i]);
        d[j->len[i]] = '\n';
    }
    return NULL;
}

/* gather(seq, start, address, capacity, threads) -> (next, nbytes, status)
 *   seq: list or tuple; items seq[start:next] were written to `address` (each followed by '\n'), nbytes in total.
 *   status 0: the end of seq was reached; 1: the buffer is full (seq[next] did not fit); 2: seq[next] is not a bytes /
 *   bytearray / one-byte-per-character str object. */
static PyObject *gather(PyObject *self, PyObject *args)
{
    PyObject *seq;
    Py_ssize_t start;
    unsigned long long address, capacity;
    int threads;
    if (!PyArg_ParseTuple(args, "OnKKi", &seq, &start, &address, &capacity, &threads)) return NULL;
    if (!PyList_Check(seq) && !PyTuple_Check(seq)) {
        PyErr_SetString(PyExc_TypeError, "gather() needs a list or a tuple");
        return NULL;
    }
    const Py_ssize_t total = PySequence_Fast_GET_SIZE(seq);
    if (start < 0 || start > total) {
        PyErr_SetString(PyExc_ValueError, "start out of range");
        return NULL;
    }
    PyObject **items = PySequence_Fast_ITEMS(seq);
    const size_t room = (size_t)(total - start);
    const char **ptr = (const char **)malloc(sizeof(char *) * (room ? room : 1));
    uint32_t *len = (uint32_t *)malloc(sizeof(uint32_t) * (room ? room : 1));
    uint64_t *off = (uint64_t *)malloc(sizeof(uint64_t) * (room ? room : 1));
    if (!ptr || !len || !off) {
        free(ptr); free(len); free(off);
        return PyErr_NoMemory();
    }
    size_t n = 0;
    uint64_t at = 0;
    int status = 0;
    for (Py_ssize_t i = start; i < total; i++) {
        PyObject *it = items[i];
        const char *p;
        Py_ssize_t l;
        if (PyBytes_CheckExact(it)) {
            p = PyBytes_AS_STRING(it);
            l = PyBytes_GET_SIZE(it);
        } else if (PyUnicode_CheckExact(it) && PyUnicode_IS_READY(it) && PyUnicode_KIND(it) == PyUnicode_1BYTE_KIND) {
            p = (const char *)PyUnicode_1BYTE_DATA(it);
            l = PyUnicode_GET_LENGTH(it);
        } else if (PyByteArray_CheckExact(it)) {
            p = PyByteArray_AS_STRING(it);
            l = PyByteArray_GET_SIZE(it);
        } else {
            status = 2;
            break;
        }
        if ((uint64_t)l >= 0xFFFFFFFFull || at + (uint64_t)l + 1 > capacity) {
            status = 1;
            break;
        }
        ptr[n] = p;
        len[n] = (uint32_t)l;
        off[n] = at;
        at += (uint64_t)l + 1;
        n++;
    }
    char *dst = (char *)(uintptr_t)address;
    if (n) {
        int T = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
        if (at < ((uint64_t)4 << 20)) T = 1;
        pthread_t th[64];
        copy_job jobs[64];
        char joinable[64];
        /* equal BYTE shares: the items of thread t are those whose offset falls into its share */
        size_t next = 0;
        for (int t = 0; t < T; t++) {
            const uint64_t limit = at / (uint64_t)T * (uint64_t)(t + 1);
            size_t e = next;
            if (t == T - 1) e = n;
            else while (e < n && off[e] < limit) e++;
            jobs[t].ptr = ptr; jobs[t].len = len; jobs[t].off = off; jobs[t].dst = dst;
            jobs[t].begin = next; jobs[t].end = e;
            next = e;
            joinable[t] = 0;
            if (t > 0 && jobs[t].end > jobs[t].begin) joinable[t] = pthread_create(&th[t], NULL, copy_worker, &jobs[t]) == 0;
            if (t > 0 && !joinable[t]) copy_worker(&jobs[t]);
        }
        copy_worker(&jobs[0]);
        for (int t = 1; t < T; t++)
            if (joinable[t]) pthread_join(th[t], NULL);
    }
    free(ptr); free(len); free(off);
    return Py_BuildValue("nKi", start + (Py_ssize_t)n, (unsigned long long)at, status);
}

static PyMethodDef methods[] = {
    {"gather", gather, METH_VARARGS, "gather(seq, start, address, capacity, threads) -> (next, nbytes, status)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_kpal_join", "flat byte stream of a list of sequences (see kpal_join.c)", -1, methods};

PyMODINIT_FUNC PyInit__kpal_join(void) { return PyModule_Create(&module); }
