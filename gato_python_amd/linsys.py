"""Host mirror of the reference's Python surface, gpu_library.linsys_solve (gpu_library.cu:85-239).

Same 14 positional arguments, same (lambda list, dz list) return, same `testiters` repeat
semantics and the same two stdout lines (gpu_library.cu:190,198; silence with GATO_VERBOSE=0).
(S, C, K) are runtime values here: taken from set_problem_size() / GATO_STATE_SIZE,
GATO_CONTROL_SIZE, GATO_KNOT_POINTS (mirroring `install.bash S C K`), else inferred from the
argument lengths.  Arithmetic is float32 like the reference unless set_precision("f64").
All compute happens in libgato_hip.so; a missing library or GPU raises.
"""
from __future__ import annotations

import array
import ctypes as ct
import os

import numpy as np

from . import _lib

_state = dict(shape=None, precision=os.environ.get("GATO_PRECISION", "f32"), stats={})


def set_problem_size(S: int, C: int, K: int):
    _state["shape"] = (int(S), int(C), int(K))


def clear_problem_size():
    _state["shape"] = None


def set_precision(p: str):
    if p not in ("f32", "f64"):
        raise ValueError("precision must be 'f32' or 'f64'")
    _state["precision"] = p


def last_stats() -> dict:
    """iters of the first repeat, per-repeat times (ms) - what the reference only prints."""
    return dict(_state["stats"])


def _shape_from(C_row, len_g, len_c):
    if _state["shape"] is not None:
        return _state["shape"]
    env = [os.environ.get(k) for k in ("GATO_STATE_SIZE", "GATO_CONTROL_SIZE", "GATO_KNOT_POINTS")]
    if all(env):
        return tuple(int(v) for v in env)
    S, C, K = ct.c_int(), ct.c_int(), ct.c_int()
    rc = _lib.lib().gato_infer_shape(C_row.__array_interface__["data"][0], len(C_row), len_g, len_c,
                                     ct.byref(S), ct.byref(C), ct.byref(K))
    if rc != 0:
        raise ValueError(_lib.lib().gato_last_error().decode())
    return S.value, C.value, K.value


try:                                    # bindings/fastseq: the list <-> buffer copies at C speed (built by __graft_entry__.build())
    from . import _gato_fastseq as _fs
except ImportError:                     # not built: array.array / numpy below (1.6 x / 4 x slower on lists, same values)
    _fs = None


def _from_list(a, code, dt):
    """Python list / tuple -> contiguous array.  The reference's callers pass lists (test_pendulum_5.py:9-25) and the copy of
    their tens of thousands of Python floats is most of a call's host time: _gato_fastseq.pack (exact-type fast paths in C,
    include/gato_pyseq.h) where it is built - its errors ARE the binding's errors (an index that does not fit 32 bits raises
    OverflowError, a float in an index list TypeError: exactly what the pybind11 module raises for the same input) - else
    array.array (1.6 x faster than numpy), with the same double -> float narrowing and the same refusals for index lists."""
    if _fs is not None:
        return np.frombuffer(_fs.pack(a, code), dt)
    try:
        return np.frombuffer(array.array(code, a), dt)
    except (TypeError, OverflowError):
        if code == "i":                 # never numpy's silent truncation / wrap of an index
            raise
        return np.ascontiguousarray(np.asarray(a, np.float64), dt)


def _index_array(a):
    """CSR index array (list, tuple, numpy array of any integer type) -> contiguous int32, REFUSING what does not fit: an int64 /
    unsigned entry beyond 32 bits raises OverflowError, a float array TypeError - as the pybind11 module does (gato_pyseq.h);
    never numpy's silent wrap."""
    if isinstance(a, np.ndarray) and a.dtype == np.int32 and a.flags.c_contiguous:
        return a
    if isinstance(a, (list, tuple)):
        return _from_list(a, "i", np.int32)
    if _fs is not None:
        return np.frombuffer(_fs.pack(a, "i"), np.int32)
    arr = np.asarray(a)
    if arr.dtype.kind not in "iu":
        raise TypeError(f"index array of dtype {arr.dtype}: integers expected")
    if arr.size and (int(arr.max()) > 2 ** 31 - 1 or int(arr.min()) < -2 ** 31):
        raise OverflowError("index does not fit 32 bits")
    return np.ascontiguousarray(arr, np.int32)


def linsys_solve(G_row, G_col, G_val, C_row, C_col, C_val, g_val, c_val, input_lambda,
                 testiters, exit_tol, max_iters, warm_start, rho):
    f64 = _state["precision"] == "f64"
    dt = np.float64 if f64 else np.float32
    i32 = _index_array
    # narrowing as std::vector<float> does
    fl = lambda a: (_from_list(a, "d" if f64 else "f", dt) if isinstance(a, (list, tuple))
                    else np.ascontiguousarray(np.asarray(a, np.float64), dt))
    G_row, G_col, C_row, C_col = i32(G_row), i32(G_col), i32(C_row), i32(C_col)
    G_val, C_val, g, c, lam_in = fl(G_val), fl(C_val), fl(g_val), fl(c_val), fl(input_lambda)
    testiters = int(testiters)
    S, C, K = _shape_from(C_row, len(g), len(c))
    if len(lam_in) < S * K:                     # the reference reads input_lambda[i], i < S*K (gpu_library.cu:162-163)
        raise ValueError(f"input_lambda has {len(lam_in)} entries, STATE_SIZE*KNOT_POINTS = {S * K}")
    if testiters < 1:
        raise ValueError("testiters must be >= 1")
    lam = np.empty(S * K, dt)
    dz = np.empty((S + C) * K - C, dt)
    iters = ct.c_int(-1)
    ms = np.zeros(testiters, np.float32)
    p = lambda a: a.__array_interface__["data"][0]       # plain address: the argtypes are declared (ctypes.data_as costs 1.5 us each)
    fn = _lib.lib().gato_linsys_solve_f64 if f64 else _lib.lib().gato_linsys_solve_f32
    rc = fn(p(G_row), len(G_row), p(G_col), p(G_val), len(G_val), p(C_row), len(C_row), p(C_col), p(C_val),
            len(C_val), p(g), len(g), p(c), len(c), p(lam_in), S, C, K, testiters, float(exit_tol),
            int(max_iters), int(bool(warm_start)), float(rho), p(lam), p(dz), ct.byref(iters), p(ms))
    if rc == -1 or rc == -2:
        raise ValueError(_lib.lib().gato_last_error().decode())
    _lib.check(rc)
    _state["stats"] = dict(iters=iters.value, ms=ms.tolist(), S=S, C=C, K=K, precision=_state["precision"])
    if os.environ.get("GATO_VERBOSE", "1") != "0":
        print("first run PCG terminated in %d iterations, time:  %f" % (iters.value, ms[0]))   # gpu_library.cu:190
        print("avg time: %f" % (float(ms.sum()) / testiters))                                   # gpu_library.cu:198
    if _fs is not None:                 # Python floats widened from float32 like the reference's (gpu_library.cu:221-229)
        k = "d" if f64 else "f"
        return _fs.unpack(lam, k), _fs.unpack(dz, k)
    return lam.astype(np.float64).tolist(), dz.astype(np.float64).tolist()
