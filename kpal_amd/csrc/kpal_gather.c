/* kpal_gather.c -- CPython extension `kpal_amd._kpal_gather`: the host side of Profile.from_sequences for LISTS of short sequences
 * (a million 150-base reads as bytes / str objects): gather(sequence, first, address, capacity, threads) -> (next, nbytes, status)
 * copies the items from `first` on into the page-locked buffer at `address`, each followed by '\n', with the walk over the list's
 * objects AND the copies cut over the threads (kpal_gather_core.h says why and how).
 *
 * kpal/klib.py:154 walks the sequences in the interpreter; the drop-in hands the GPU one flat byte stream, sequences separated by
 * '\n' (kpal_amd/klib.py).  Items read here: bytes, bytearray, and str whose characters all fit one byte (ASCII / latin-1, whose
 * storage IS its latin-1 encoding); anything else ends the call early and the caller encodes that one item itself.
 *
 * The calling thread keeps the GIL throughout: no interpreter code can free or change an item meanwhile, so the items need no
 * INCREF / DECREF, and the worker threads only READ object headers and payloads (type pointer, length, kind, data pointer) -- never
 * a reference count, never through an API that may allocate or raise.  No HIP, no link to libkpal_hip.so.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

#include "kpal_gather_core.h"

/* (PyUnicode_IS_READY is deprecated since Python 3.12, where every str is in its canonical form) */
#if PY_VERSION_HEX < 0x030C0000
#define KPAL_UNICODE_READY(o) PyUnicode_IS_READY(o)
#else
#define KPAL_UNICODE_READY(o) 1
#endif

static int describe(void *ctx, size_t i, const char **ptr, uint64_t *len)
{
    PyObject *it = ((PyObject **)ctx)[i];
    if (PyBytes_CheckExact(it)) {
        *ptr = PyBytes_AS_STRING(it);
        *len = (uint64_t)PyBytes_GET_SIZE(it);
    } else if (PyUnicode_CheckExact(it) && KPAL_UNICODE_READY(it) && PyUnicode_KIND(it) == PyUnicode_1BYTE_KIND) {
        *ptr = (const char *)PyUnicode_1BYTE_DATA(it);
        *len = (uint64_t)PyUnicode_GET_LENGTH(it);
    } else if (PyByteArray_CheckExact(it)) {
        *ptr = PyByteArray_AS_STRING(it);
        *len = (uint64_t)PyByteArray_GET_SIZE(it);
    } else {
        return 0;
    }
    return 1;
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
    const kpal_gather_result r = kpal_gather_run(describe, (void *)PySequence_Fast_ITEMS(seq), (size_t)start, (size_t)(total - start),
                                                 (char *)(uintptr_t)address, (uint64_t)capacity, threads);
    if (r.status < 0) return PyErr_NoMemory();
    return Py_BuildValue("nKi", start + (Py_ssize_t)r.n, (unsigned long long)r.bytes, r.status);
}

static PyMethodDef methods[] = {
    {"gather", gather, METH_VARARGS, "gather(seq, start, address, capacity, threads) -> (next, nbytes, status)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_kpal_gather", "flat byte stream of a list of sequences (see kpal_gather.c)", -1, methods};

PyMODINIT_FUNC PyInit__kpal_gather(void) { return PyModule_Create(&module); }
