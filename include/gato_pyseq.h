/* Python sequence <-> contiguous C array, for the two Python bindings of the drop-in (bindings/pybind11, bindings/fastseq).
 * The reference's binding takes std::vector<int> / std::vector<float> through pybind11/stl.h (gpu_library.cu:21,85-87): any
 * Python sequence of numbers is accepted and COPIED element by element, ints are accepted where floats are expected
 * (test_pendulum_5.py:18 passes c_val = [0, ...]).  What its callers really pass are lists of tens of thousands of Python
 * floats, and that copy is most of a call's host time (0.28 of 0.43 ms at 14/7/50 through array.array, more through the
 * generic pybind11 caster).  Here: exact-type fast paths (PyFloat_AS_DOUBLE, one-digit PyLong), the buffer protocol for numpy
 * arrays (a memcpy when the item type already matches), the generic number protocol for everything else.
 * Header only, CPython C API only (no numpy, no pybind11).  Every function returns 0 or sets a Python error and returns -1. */
#ifndef GATO_PYSEQ_H
#define GATO_PYSEQ_H
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <string.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

static inline int gato_pyseq_item_double(PyObject *it, double *out)
{
    if (PyFloat_CheckExact(it)) { *out = PyFloat_AS_DOUBLE(it); return 0; }
    if (PyLong_CheckExact(it)) {
        const double v = PyLong_AsDouble(it);
        if (v == -1.0 && PyErr_Occurred()) return -1;
        *out = v;
        return 0;
    }
    if (PyUnicode_Check(it) || PyBytes_Check(it)) { PyErr_SetString(PyExc_TypeError, "expected a number, got a string"); return -1; }
    {
        const double v = PyFloat_AsDouble(it);            /* numpy scalars, Fraction, anything with __float__ / __index__ */
        if (v == -1.0 && PyErr_Occurred()) return -1;
        *out = v;
        return 0;
    }
}

static inline int gato_pyseq_item_int(PyObject *it, int *out)
{
    long v;
    if (PyFloat_Check(it)) { PyErr_SetString(PyExc_TypeError, "expected an integer, got a float"); return -1; }   /* as pybind11's int caster */
    v = PyLong_AsLong(it);                                /* exact ints and anything with __index__ */
    if (v == -1 && PyErr_Occurred()) return -1;
    if (v < -2147483647L - 1 || v > 2147483647L) { PyErr_SetString(PyExc_OverflowError, "index does not fit 32 bits"); return -1; }
    *out = (int)v;
    return 0;
}

/* kind: 'f' float32, 'd' float64, 'i' int32.  On success *data is malloc'd (caller frees) and *n its element count. */
static inline int gato_pyseq_pack(PyObject *obj, char kind, void **data, Py_ssize_t *n)
{
    const size_t esz = kind == 'd' ? 8 : 4;
    *data = NULL; *n = 0;
    if (PyUnicode_Check(obj) || PyBytes_Check(obj)) { PyErr_SetString(PyExc_TypeError, "expected a sequence of numbers, got a string"); return -1; }
    /* numpy arrays and other exporters of a one-dimensional contiguous buffer */
    if (!PyList_CheckExact(obj) && !PyTuple_CheckExact(obj) && PyObject_CheckBuffer(obj)) {
        Py_buffer vw;
        if (PyObject_GetBuffer(obj, &vw, PyBUF_FORMAT | PyBUF_C_CONTIGUOUS) == 0) {
            const char *f = vw.format ? vw.format : "B";
            char c;
            Py_ssize_t cnt, i;
            void *buf;
            int ovf = 0;
            while (*f == '@' || *f == '=' || *f == '<') ++f;
            c = *f;
            if (vw.ndim == 1 && f[1] == '\0' && (c == 'f' || c == 'd' || c == 'i' || c == 'l' || c == 'q' || c == 'I' || c == 'L' || c == 'Q') &&
                !(kind == 'i' && (c == 'f' || c == 'd'))) {
                cnt = vw.shape ? vw.shape[0] : vw.len / vw.itemsize;
                buf = malloc(cnt ? (size_t)cnt * esz : 1);
                if (!buf) { PyBuffer_Release(&vw); PyErr_NoMemory(); return -1; }
#define GATO_PYSEQ_CONV(SRC_T)                                                                                         \
    do {                                                                                                               \
        const SRC_T *s_ = (const SRC_T *)vw.buf;                                                                       \
        if (kind == 'f') for (i = 0; i < cnt; ++i) ((float *)buf)[i] = (float)s_[i];                                   \
        else if (kind == 'd') for (i = 0; i < cnt; ++i) ((double *)buf)[i] = (double)s_[i];                            \
        else for (i = 0; i < cnt; ++i) {                                                                               \
            const SRC_T x_ = s_[i];                                                                                    \
            const int y_ = (int)x_;                                                                                    \
            if ((SRC_T)y_ != x_ || ((x_ < (SRC_T)0) != (y_ < 0))) ovf = 1;      /* as the list path: no silent wrap */ \
            ((int *)buf)[i] = y_;                                                                                      \
        }                                                                                                              \
    } while (0)
                if ((c == 'f' && kind == 'f') || (c == 'd' && kind == 'd') || (c == 'i' && kind == 'i')) memcpy(buf, vw.buf, (size_t)cnt * esz);
                else if (c == 'f') GATO_PYSEQ_CONV(float);
                else if (c == 'd') GATO_PYSEQ_CONV(double);
                else if (c == 'i') GATO_PYSEQ_CONV(int);
                else if (c == 'I') GATO_PYSEQ_CONV(unsigned int);
                else if (vw.itemsize == 8 && (c == 'l' || c == 'q')) GATO_PYSEQ_CONV(long long);
                else if (vw.itemsize == 8) GATO_PYSEQ_CONV(unsigned long long);
                else if (c == 'l') GATO_PYSEQ_CONV(int);
                else GATO_PYSEQ_CONV(unsigned int);
#undef GATO_PYSEQ_CONV
                PyBuffer_Release(&vw);
                if (ovf) { free(buf); PyErr_SetString(PyExc_OverflowError, "index does not fit 32 bits"); return -1; }
                *data = buf; *n = cnt;
                return 0;
            }
            PyBuffer_Release(&vw);                        /* another item type or rank: element by element below */
        } else PyErr_Clear();
    }
    {
        PyObject *fast = PySequence_Fast(obj, "expected a sequence of numbers");
        PyObject **items;
        Py_ssize_t cnt, i;
        void *buf;
        if (!fast) return -1;
        cnt = PySequence_Fast_GET_SIZE(fast);
        items = PySequence_Fast_ITEMS(fast);
        buf = malloc(cnt ? (size_t)cnt * esz : 1);
        if (!buf) { Py_DECREF(fast); PyErr_NoMemory(); return -1; }
        if (kind == 'i') {
            int *d = (int *)buf;
            for (i = 0; i < cnt; ++i)
                if (gato_pyseq_item_int(items[i], d + i)) { free(buf); Py_DECREF(fast); return -1; }
        } else {
            for (i = 0; i < cnt; ++i) {
                PyObject *it = items[i];
                double v;
                if (PyFloat_CheckExact(it)) v = PyFloat_AS_DOUBLE(it);
                else if (gato_pyseq_item_double(it, &v)) { free(buf); Py_DECREF(fast); return -1; }
                if (kind == 'f') ((float *)buf)[i] = (float)v;            /* narrowing as std::vector<float> does */
                else ((double *)buf)[i] = v;
            }
        }
        Py_DECREF(fast);
        *data = buf; *n = cnt;
        return 0;
    }
}

/* new list of Python floats from n float32 / float64 values (the reference widens float32 the same way, gpu_library.cu:221-229) */
static inline PyObject *gato_pyseq_list(const void *data, Py_ssize_t n, char kind)
{
    PyObject *l = PyList_New(n);
    Py_ssize_t i;
    if (!l) return NULL;
    for (i = 0; i < n; ++i) {
        PyObject *f = PyFloat_FromDouble(kind == 'f' ? (double)((const float *)data)[i] : ((const double *)data)[i]);
        if (!f) { Py_DECREF(l); return NULL; }
        PyList_SET_ITEM(l, i, f);
    }
    return l;
}

#ifdef __cplusplus
}
#endif
#endif
