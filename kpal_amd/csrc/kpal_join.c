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
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "kpal_join_core.h"

/* (PyUnicode_IS_READY is deprecated since Python 3.12, where every str is in its canonical form) */
#if PY_VERSION_HEX < 0x030C0000
#define KPAL_UNICODE_READY(o) PyUnicode_IS_READY(o)
#else
#define KPAL_UNICODE_READY(o) 1
#endif

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
        } else if (PyUnicode_CheckExact(it) && KPAL_UNICODE_READY(it) && PyUnicode_KIND(it) == PyUnicode_1BYTE_KIND) {
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
    kpal_join_copy(ptr, len, off, n, at, dst, threads, (uint64_t)4 << 20);
    free(ptr); free(len); free(off);
    return Py_BuildValue("nKi", start + (Py_ssize_t)n, (unsigned long long)at, status);
}

static PyMethodDef methods[] = {
    {"gather", gather, METH_VARARGS, "gather(seq, start, address, capacity, threads) -> (next, nbytes, status)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_kpal_join", "flat byte stream of a list of sequences (see kpal_join.c)", -1, methods};

PyMODINIT_FUNC PyInit__kpal_join(void) { return PyModule_Create(&module); }
