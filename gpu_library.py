"""Drop-in for the reference's pybind11 module `gpu_library` (gpu_library.cu:236-239):

    import gpu_library
    l, dz = gpu_library.linsys_solve(G_row, G_col, G_val, C_row, C_col, C_val, g_val, c_val,
                                     input_lambda, testiters, exit_tol, max_iters, warm_start, rho)

Thin host layer over the C ABI of libgato_hip.so (include/gato_hip.h); see INTEGRATION.md for the
pybind11 binding a maintainer of the reference would add instead.
"""
from gato_python_amd.linsys import (clear_problem_size, last_stats, linsys_solve,  # noqa: F401
                                    set_precision, set_problem_size)
