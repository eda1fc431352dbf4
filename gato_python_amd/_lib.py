"""ctypes binding of libgato_hip.so (include/gato_hip.h).

The HIP library IS the product path: if it is missing this module raises, there is no CPU
fallback anywhere in gato_python_amd (the oracle under oracle/ is test infrastructure only).
"""
from __future__ import annotations

import ctypes as ct
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("GATO_HIP_LIB") or os.path.join(_HERE, "libgato_hip.so")   # override: A/B builds only
_LIB = None

GATO_F32, GATO_F64 = 0, 1
PCG_AUTO, PCG_RESIDENT, PCG_STREAMING = 0, 1, 2
PRECON_STAIR, PRECON_BLOCK_JACOBI, PRECON_POINT_JACOBI = 0, 1, 2

ERRORS = {-1: "EINVAL", -2: "ESHAPE", -3: "EHIP", -4: "ENODEV", -5: "ETIMEOUT"}

# Every symbol include/gato_hip.h declares (tests check the .so exports all of them).
SYMBOLS = [
    "gato_last_error", "gato_version", "gato_num_shapes", "gato_shape", "gato_device_info",
    "gato_infer_shape", "gato_solver_create", "gato_solver_create_batched", "gato_linsys_device_batched",
    "gato_solver_destroy", "gato_solver_buffer",
    "gato_solver_set_option", "gato_solver_get_option", "gato_convert", "gato_form_schur",
    "gato_form_ss", "gato_pcg", "gato_pcg_status", "gato_pcg_last_ms", "gato_compute_dz", "gato_linsys_device",
    "gato_linsys_solve_f32", "gato_linsys_solve_f64",
    "gato_shard_pcg_init", "gato_shard_pcg_phase_a", "gato_shard_pcg_phase_b", "gato_shard_pcg_finish",
    "gato_shard_pcg_done", "gato_linsys_device_blocks", "gato_release_cache", "gato_solver_recover",
    "gato_cluster_knot_range", "gato_cluster_create", "gato_cluster_local_mirror", "gato_cluster_connect",
    "gato_cluster_pcg", "gato_cluster_linsys", "gato_cluster_destroy", "gato_cluster_launches_left", "gato_cluster_rewind", "gato_cluster_fits", "gato_last_stage_ms", "gato_solver_tune",
]


class GatoError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libgato_hip: {ERRORS.get(code, code)}: {msg}")
        self.code = code


def build(force: bool = False) -> str:
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(SO_PATH):
        cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4", "-s"] + (["-B"] if force else [])
        subprocess.check_call(cmd)
    return SO_PATH


def lib() -> ct.CDLL:
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH) and "GATO_HIP_LIB" not in os.environ:
            try:                       # source tree without the built library: compile it (hipcc; a clean build of all shapes is ~3 min at -j8, ~6 CPU-minutes); no fallback
                build()
            except Exception as e:     # noqa: BLE001
                raise ImportError(f"{SO_PATH} is missing and building it failed: {e}") from e
        if not os.path.exists(SO_PATH):
            raise ImportError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C gato_python_amd/csrc`.  There is no CPU fallback.")
        L = ct.CDLL(SO_PATH)
        L.gato_last_error.restype = ct.c_char_p
        L.gato_solver_buffer.restype = ct.c_void_p
        vp, ip, i, d = ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_double
        L.gato_solver_create.argtypes = [i, i, i, i, i, ct.POINTER(ct.c_void_p)]
        L.gato_solver_create_batched.argtypes = [i, i, i, i, i, i, ct.POINTER(ct.c_void_p)]
        L.gato_linsys_device_batched.argtypes = [vp, ip, ip, vp, i, ip, ip, vp, i, vp, vp, d, i, d, vp, vp, vp, vp]
        L.gato_solver_destroy.argtypes = [vp]
        L.gato_solver_buffer.argtypes = [vp, i]
        L.gato_solver_set_option.argtypes = [vp, ct.c_char_p, i]
        L.gato_solver_get_option.argtypes = [vp, ct.c_char_p, ct.POINTER(ct.c_int)]
        L.gato_convert.argtypes = [vp, ip, ip, vp, ip, ip, vp, d, vp, vp, vp]
        L.gato_form_schur.argtypes = [vp] * 10
        L.gato_form_ss.argtypes = [vp, vp, vp, vp]
        L.gato_pcg.argtypes = [vp, vp, vp, vp, vp, d, i, vp, vp]
        L.gato_pcg_status.argtypes = [vp, ct.POINTER(ct.c_int)]
        L.gato_pcg_last_ms.argtypes = [vp, ct.POINTER(ct.c_float)]
        L.gato_last_stage_ms.argtypes = [vp, ct.POINTER(ct.c_float)]
        L.gato_compute_dz.argtypes = [vp] * 7
        L.gato_linsys_device.argtypes = [vp, ip, ip, vp, ip, ip, vp, vp, vp, d, i, d, vp, vp, vp]
        L.gato_infer_shape.argtypes = [ip, i, i, i] + [ct.POINTER(ct.c_int)] * 3
        L.gato_device_info.argtypes = [i, ct.POINTER(ct.c_int), ct.POINTER(ct.c_int), ct.c_char_p, i]
        L.gato_shard_pcg_init.argtypes = [vp, i, i, i, i, vp, vp, vp, d, i, vp, vp]
        L.gato_shard_pcg_phase_a.argtypes = [vp, i, vp, vp, vp, vp]
        L.gato_shard_pcg_phase_b.argtypes = [vp, i, vp, vp, vp, vp]
        L.gato_shard_pcg_finish.argtypes = [vp, vp, vp, vp, vp]
        L.gato_shard_pcg_done.argtypes = [vp, ct.POINTER(ct.c_int), vp]
        L.gato_linsys_device_blocks.argtypes = [vp, vp, vp, vp, vp, d, i, d, vp, vp, vp]
        L.gato_solver_recover.argtypes = [vp, ct.POINTER(ct.c_int), vp]
        L.gato_cluster_knot_range.argtypes = [i, i, i, ct.POINTER(ct.c_int), ct.POINTER(ct.c_int)]
        L.gato_cluster_create.argtypes = [vp, i, i, vp]
        L.gato_cluster_local_mirror.argtypes = [vp]
        L.gato_cluster_local_mirror.restype = ct.c_void_p
        L.gato_cluster_connect.argtypes = [vp, vp, vp]
        L.gato_cluster_pcg.argtypes = [vp, vp, vp, vp, vp, d, i, vp, vp]
        L.gato_cluster_linsys.argtypes = [vp, ip, ip, vp, ip, ip, vp, vp, vp, d, i, d, vp, vp, vp, vp]
        L.gato_cluster_destroy.argtypes = [vp]
        L.gato_cluster_launches_left.argtypes = [vp, i, ct.POINTER(ct.c_longlong)]
        L.gato_cluster_rewind.argtypes = [vp]
        L.gato_cluster_fits.argtypes = [vp, ct.POINTER(ct.c_int), ct.POINTER(ct.c_int)]
        L.gato_solver_tune.argtypes = [vp, vp]
        f = ct.c_float
        L.gato_linsys_solve_f32.argtypes = [ip, i, ip, vp, i, ip, i, ip, vp, i, vp, i, vp, i, vp,
                                            i, i, i, i, f, i, i, f, vp, vp, vp, vp]
        L.gato_linsys_solve_f64.argtypes = [ip, i, ip, vp, i, ip, i, ip, vp, i, vp, i, vp, i, vp,
                                            i, i, i, i, d, i, i, d, vp, vp, vp, vp]
        _LIB = L
    return _LIB


def check(rc: int):
    if rc != 0:
        raise GatoError(rc, lib().gato_last_error().decode())


def shapes():
    L = lib()
    out = []
    for k in range(L.gato_num_shapes()):
        s, c = ct.c_int(), ct.c_int()
        L.gato_shape(k, ct.byref(s), ct.byref(c))
        out.append((s.value, c.value))
    return out
