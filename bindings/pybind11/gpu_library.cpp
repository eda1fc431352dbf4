// pybind11 module `gpu_library` over the C ABI of libgato_hip.so - the binding a maintainer of the
// reference would put in place of gpu_library.cu:85-239 (same module name, same single function, same
// 14 positional arguments, same (list, list) return, return_value_policy::move, no keyword names).
// STATE_SIZE / CONTROL_SIZE / KNOT_POINTS are runtime values: taken from the environment
// (GATO_STATE_SIZE, GATO_CONTROL_SIZE, GATO_KNOT_POINTS - the `install.bash S C K` of the reference) or
// inferred from the argument lengths (gato_infer_shape).
#include <pybind11/pybind11.h>

#include <cstdio>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <vector>

#include "gato_hip.h"
#include "gato_pyseq.h"

namespace py = pybind11;

// The nine array arguments are taken as Python objects and copied by include/gato_pyseq.h instead of through
// std::vector casters of pybind11/stl.h (gpu_library.cu:21): the same acceptance - any sequence of numbers, ints where
// floats are expected, copied - but exact-type fast paths for the lists the reference's callers pass (58 k elements per
// 14/7/50 call) and the buffer protocol for numpy arrays, which the generic caster walks element by element through
// Python objects (measured on the GPU box, 14/7/50: lists 0.37 ms per call, numpy arrays 0.95 ms with the stl casters).
template <typename T> struct Packed {
    T *p = nullptr;
    Py_ssize_t n = 0;
    Packed(const py::object &o, char kind)
    {
        void *d = nullptr;
        if (gato_pyseq_pack(o.ptr(), kind, &d, &n)) throw py::error_already_set();
        p = static_cast<T *>(d);
    }
    ~Packed() { free(p); }
    Packed(const Packed &) = delete;
    Packed &operator=(const Packed &) = delete;
};

// side channel (extra information never changes the returned tuple): what the most recent call printed, as data
static int g_last_iters = -1, g_last_S = 0, g_last_C = 0, g_last_K = 0;
static std::vector<float> g_last_ms;

static py::tuple main_call(py::object sG_indptr, py::object sG_indices, py::object sG_data, py::object sC_indptr,
                           py::object sC_indices, py::object sC_data, py::object g_, py::object c_, py::object input_lambda_,
                           int testiters, float exit_tol, int max_iters, bool warm_start, float rho)
{
    Packed<int> G_row(sG_indptr, 'i'), G_col(sG_indices, 'i'), C_row(sC_indptr, 'i'), C_col(sC_indices, 'i');
    Packed<float> G_val(sG_data, 'f'), C_val(sC_data, 'f'), g(g_, 'f'), c(c_, 'f'), input_lambda(input_lambda_, 'f');
    int S = 0, C = 0, K = 0;
    const char *eS = getenv("GATO_STATE_SIZE"), *eC = getenv("GATO_CONTROL_SIZE"), *eK = getenv("GATO_KNOT_POINTS");
    if (eS && eC && eK) { S = atoi(eS); C = atoi(eC); K = atoi(eK); }
    else if (gato_infer_shape(C_row.p, (int)C_row.n, (int)g.n, (int)c.n, &S, &C, &K))
        throw py::value_error(gato_last_error());
    if ((int)input_lambda.n < S * K) throw py::value_error("input_lambda shorter than STATE_SIZE*KNOT_POINTS");
    if (testiters < 1) throw py::value_error("testiters must be >= 1");
    std::vector<float> lambda((size_t)S * K), dz((size_t)(S + C) * K - C), ms(testiters);
    int iters = -1, rc;
    {
        py::gil_scoped_release nogil;   // the reference holds the GIL for the whole solve
        rc = gato_linsys_solve_f32(G_row.p, (int)G_row.n, G_col.p, G_val.p, (int)G_val.n, C_row.p, (int)C_row.n, C_col.p,
                                   C_val.p, (int)C_val.n, g.p, (int)g.n, c.p, (int)c.n, input_lambda.p, S, C, K, testiters,
                                   exit_tol, max_iters, warm_start ? 1 : 0, rho, lambda.data(), dz.data(), &iters, ms.data());
    }
    if (rc == GATO_EINVAL || rc == GATO_ESHAPE) throw py::value_error(gato_last_error());
    if (rc) throw std::runtime_error(gato_last_error());
    g_last_iters = iters; g_last_S = S; g_last_C = C; g_last_K = K; g_last_ms = ms;
    const char *verbose = getenv("GATO_VERBOSE");
    if (!verbose || verbose[0] != '0') {
        float sum = 0;
        for (float t : ms) sum += t;
        printf("first run PCG terminated in %d iterations, time:  %f\n", iters, ms[0]);   // gpu_library.cu:190
        printf("avg time: %f\n", sum / testiters);                                           // gpu_library.cu:198
    }
    PyObject *pl = gato_pyseq_list(lambda.data(), (Py_ssize_t)lambda.size(), 'f');           // gpu_library.cu:221-229
    PyObject *pd = pl ? gato_pyseq_list(dz.data(), (Py_ssize_t)dz.size(), 'f') : nullptr;
    if (!pl || !pd) { Py_XDECREF(pl); throw py::error_already_set(); }
    return py::make_tuple(py::reinterpret_steal<py::list>(pl), py::reinterpret_steal<py::list>(pd));
}

PYBIND11_MODULE(gpu_library, m)
{
    m.def("linsys_solve", &main_call, py::return_value_policy::move);
    // not in the reference (it only prints the count, gpu_library.cu:190): iterations of the FIRST repeat and the time of every
    // repeat of the most recent call, as gpu_library.py's last_stats() of the ctypes drop-in
    m.def("last_stats", []() {
        py::dict d;
        py::list ms;
        for (float t : g_last_ms) ms.append(t);
        d["iters"] = g_last_iters; d["ms"] = ms; d["S"] = g_last_S; d["C"] = g_last_C; d["K"] = g_last_K; d["precision"] = "f32";
        return d;
    });
}
