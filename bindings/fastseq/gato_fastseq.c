/* CPython extension `_gato_fastseq` (no numpy, no pybind11): the list <-> buffer copies of the ctypes drop-in
 * (gato_python_amd/linsys.py) at C speed - include/gato_pyseq.h has the why.  pack(seq, kind) -> bytearray with the
 * values as float32 ('f'), float64 ('d') or int32 ('i'); unpack(buffer, kind) -> list of Python floats. */
#include "gato_pyseq.h"

static PyObject *fs_pack(PyObject *self, PyObject *args)
{
    PyObject *seq, *out;
    const char *kind;
    void *data;
    Py_ssize_t n;
    (void)self;
    if (!PyArg_ParseTuple(args, "Os", &seq, &kind)) return NULL;
    if ((kind[0] != 'f' && kind[0] != 'd' && kind[0] != 'i') || kind[1]) { PyErr_SetString(PyExc_ValueError, "kind: 'f', 'd' or 'i'"); return NULL; }
    if (gato_pyseq_pack(seq, kind[0], &data, &n)) return NULL;
    out = PyByteArray_FromStringAndSize((const char *)data, n * (kind[0] == 'd' ? 8 : 4));
    free(data);
    return out;
}

static PyObject *fs_unpack(PyObject *self, PyObject *args)
{
    Py_buffer vw;
    const char *kind;
    PyObject *out;
    (void)self;
    if (!PyArg_ParseTuple(args, "y*s", &vw, &kind)) return NULL;
    if ((kind[0] != 'f' && kind[0] != 'd') || kind[1]) { PyBuffer_Release(&vw); PyErr_SetString(PyExc_ValueError, "kind: 'f' or 'd'"); return NULL; }
    out = gato_pyseq_list(vw.buf, vw.len / (kind[0] == 'd' ? 8 : 4), kind[0]);
    PyBuffer_Release(&vw);
    return out;
}

static PyMethodDef fs_methods[] = {
    {"pack", fs_pack, METH_VARARGS, "pack(sequence, kind) -> bytearray of float32 ('f') / float64 ('d') / int32 ('i') values"},
    {"unpack", fs_unpack, METH_VARARGS, "unpack(buffer, kind) -> list of floats from float32 ('f') / float64 ('d') values"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef fs_module = {PyModuleDef_HEAD_INIT, "_gato_fastseq", "list <-> buffer copies for the gato drop-in", -1, fs_methods, NULL, NULL, NULL, NULL};
PyMODINIT_FUNC PyInit__gato_fastseq(void) { return PyModule_Create(&fs_module); }
