// pybind11 module `gpu_library` over the C ABI of libgato_hip.so - the binding a maintainer of the
// reference would put in place of gpu_library.cu:85-239 (same module name, same single function, same
// 14 positional arguments, same (list, list) return, return_value_policy::move, no keyword names).
// STATE_SIZE / CONTROL_SIZE / KNOT_POINTS are runtime values: taken from the environment
// (GATO_STATE_SIZE, GATO_CONTROL_SIZE, GATO_KNOT_POINTS - the `install.bash S C K` of the reference) or
// inferred from the argument lengths (gato_infer_shape).
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <vector>

#include "gato_hip.h"

namespace py = pybind11;

static py::tuple main_call(std::vector<int> sG_indptr, std::vector<int> sG_indices, std::vector<float> sG_data,
                           std::vector<int> sC_indptr, std::vector<int> sC_indices, std::vector<float> sC_data,
                           std::vector<float> g, std::vector<float> c, std::vector<float> input_lambda, int testiters,
                           float exit_tol, int max_iters, bool warm_start, float rho)
{
    int S = 0, C = 0, K = 0;
    const char *eS = getenv("GATO_STATE_SIZE"), *eC = getenv("GATO_CONTROL_SIZE"), *eK = getenv("GATO_KNOT_POINTS");
    if (eS && eC && eK) { S = atoi(eS); C = atoi(eC); K = atoi(eK); }
    else if (gato_infer_shape(sC_indptr.data(), (int)sC_indptr.size(), (int)g.size(), (int)c.size(), &S, &C, &K))
        throw py::value_error(gato_last_error());
    if ((int)input_lambda.size() < S * K) throw py::value_error("input_lambda shorter than STATE_SIZE*KNOT_POINTS");
    if (testiters < 1) throw py::value_error("testiters must be >= 1");
    std::vector<float> lambda((size_t)S * K), dz((size_t)(S + C) * K - C), ms(testiters);
    int iters = -1, rc;
    {
        py::gil_scoped_release nogil;   // the reference holds the GIL for the whole solve
        rc = gato_linsys_solve_f32(sG_indptr.data(), (int)sG_indptr.size(), sG_indices.data(), sG_data.data(),
                                   (int)sG_data.size(), sC_indptr.data(), (int)sC_indptr.size(), sC_indices.data(),
                                   sC_data.data(), (int)sC_data.size(), g.data(), (int)g.size(), c.data(), (int)c.size(),
                                   input_lambda.data(), S, C, K, testiters, exit_tol, max_iters, warm_start ? 1 : 0, rho,
                                   lambda.data(), dz.data(), &iters, ms.data());
    }
    if (rc == GATO_EINVAL || rc == GATO_ESHAPE) throw py::value_error(gato_last_error());
    if (rc) throw std::runtime_error(gato_last_error());
    const char *verbose = getenv("GATO_VERBOSE");
    if (!verbose || verbose[0] != '0') {
        float sum = 0;
        for (float t : ms) sum += t;
        printf("first run PCG terminated in %d iterations, time:  %f\n", iters, ms[0]);   // gpu_library.cu:190
        printf("avg time: %f\n", sum / testiters);                                           // gpu_library.cu:198
    }
    py::list p_lambda, p_dz;                                                                 // gpu_library.cu:221-229
    for (float v : lambda) p_lambda.append(v);
    for (float v : dz) p_dz.append(v);
    return py::make_tuple(p_lambda, p_dz);
}

PYBIND11_MODULE(gpu_library, m)
{
    m.def("linsys_solve", &main_call, py::return_value_policy::move);
}
