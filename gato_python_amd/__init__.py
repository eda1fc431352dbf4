"""gato_python_amd - MI355X-native PCG / Schur-complement KKT solve (the hot path of gato-python).

Public surface:
  gato_python_amd.linsys_solve(...)      drop-in for gpu_library.linsys_solve (gpu_library.cu:236-239)
  gato_python_amd.Solver                 device-resident stage-level API over include/gato_hip.h
  gato_python_amd.synth                  synthetic OCP inputs (the reference ships pendulum data only)
"""
from .linsys import (clear_problem_size, last_stats, linsys_solve, set_precision,  # noqa: F401
                     set_problem_size)


def __getattr__(name):
    if name == "Solver":              # torch import deferred: linsys_solve itself needs only ctypes
        from .solver import Solver
        return Solver
    raise AttributeError(name)
