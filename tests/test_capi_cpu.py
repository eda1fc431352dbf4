"""CPU suite: the C-ABI library loads, exports every symbol include/gato_hip.h declares, and the
host logic (shape inference, argument validation) behaves - no compute calls without a GPU."""
import ctypes as ct
import os
import re

import numpy as np
import pytest

from gato_python_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "gato_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gato_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/gato_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == names


def test_compiled_shapes():
    for sh in [(2, 1), (14, 7), (32, 16), (4, 2), (6, 3), (12, 6)]:
        assert sh in _lib.shapes()


@pytest.mark.parametrize("S,C,K", [(2, 1, 5), (14, 7, 50), (32, 16, 7), (14, 7, 2)])
def test_infer_shape(S, C, K):
    s = synth.make_system(S, C, K, seed=0)
    a, b, c = ct.c_int(), ct.c_int(), ct.c_int()
    rc = _lib.lib().gato_infer_shape(s.C_row.ctypes.data_as(ct.c_void_p), len(s.C_row), len(s.g), len(s.c),
                                     ct.byref(a), ct.byref(b), ct.byref(c))
    assert rc == 0 and (a.value, b.value, c.value) == (S, C, K)


def test_infer_shape_pendulum_literals():
    p = synth.pendulum_system()
    a, b, c = ct.c_int(), ct.c_int(), ct.c_int()
    rc = _lib.lib().gato_infer_shape(p.C_row.ctypes.data_as(ct.c_void_p), len(p.C_row), len(p.g), len(p.c),
                                     ct.byref(a), ct.byref(b), ct.byref(c))
    assert rc == 0 and (a.value, b.value, c.value) == (2, 1, 5)


def test_infer_shape_rejects_inconsistent_lengths():
    p = synth.pendulum_system()
    a, b, c = ct.c_int(), ct.c_int(), ct.c_int()
    rc = _lib.lib().gato_infer_shape(p.C_row.ctypes.data_as(ct.c_void_p), len(p.C_row) - 1, len(p.g), len(p.c),
                                     ct.byref(a), ct.byref(b), ct.byref(c))
    assert rc == -1 and b"len(C_row)" in _lib.lib().gato_last_error()


def test_unknown_shape_is_an_error_not_a_fallback():
    import gpu_library
    with pytest.raises(ValueError, match="not a compiled shape"):
        gpu_library.linsys_solve([0, 1], [0], [1.], [0, 1], [0], [1.], [1.], [0.], [0.], 1, 1e-6, 10, False, 1e-3)


def test_short_input_lambda_rejected():
    import gpu_library
    P = synth.PENDULUM
    with pytest.raises(ValueError, match="input_lambda"):
        gpu_library.linsys_solve(P["G_row"], P["G_col"], P["G_val"], P["C_row"], P["C_col"], P["C_val"],
                                 P["g_val"], P["c_val"], [0.] * 3, 1, 1e-6, 10, False, 1e-3)


@pytest.mark.parametrize("B", [0, -3, 65536, 1 << 20])
def test_batch_size_outside_the_launch_grid_is_refused(B):
    """One grid row per system: a batch beyond 65 535 systems (or below 1) is refused when the solver is created, with the reason
    (checked before any device call: the same answer with and without a GPU)."""
    h = ct.c_void_p()
    rc = _lib.lib().gato_solver_create_batched(14, 7, 50, B, 0, 0, ct.byref(h))
    assert rc == -1 and b"batch must be in 1 .. 65535" in _lib.lib().gato_last_error()


def test_no_gpu_fails_loudly():
    """Without a GPU the product path must raise, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import gpu_library
    P = synth.PENDULUM
    with pytest.raises(_lib.GatoError, match="ENODEV|EHIP"):
        gpu_library.linsys_solve(P["G_row"], P["G_col"], P["G_val"], P["C_row"], P["C_col"], P["C_val"],
                                 P["g_val"], P["c_val"], P["input_lambda"], 1, 1e-6, 10, False, 1e-3)


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gato_python_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libgato_oracle" not in txt, f
    assert "oracle" not in open(os.path.join(ROOT, "gpu_library.py")).read().replace("KKT oracle", "")


def test_pybind11_module_builds_and_imports():
    """The pybind11 `gpu_library` (bindings/pybind11) exposes the reference's surface - `linsys_solve` - and the one side channel
    SURVEY.md 8(b) allows beside it (`last_stats()`: what the reference only prints), nothing else."""
    import subprocess
    import sys
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "bindings", "pybind11"), "-s"])
    code = ("import gpu_library, inspect; assert gpu_library.__file__.endswith('.so'), gpu_library.__file__; "
            "names=[n for n in dir(gpu_library) if not n.startswith('_')]; assert names==['last_stats', 'linsys_solve'], names; "
            "print('ok')")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "bindings", "pybind11", "build"))
    r = subprocess.run([sys.executable, "-c", code], cwd="/tmp", env=env, capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_malformed_csr_is_rejected_on_the_host():
    """Out-of-range CSR indices would be out-of-bounds device writes in the scatter: refused before any launch."""
    import gpu_library
    P = dict(synth.PENDULUM)
    bad_col = list(P["C_col"])
    bad_col[5] = 99
    with pytest.raises(ValueError, match="outside"):
        gpu_library.linsys_solve(P["G_row"], P["G_col"], P["G_val"], P["C_row"], bad_col, P["C_val"], P["g_val"],
                                 P["c_val"], P["input_lambda"], 1, 1e-6, 10, False, 1e-3)
    bad_row = list(P["G_row"])
    bad_row[3] = 1
    with pytest.raises(ValueError, match="monotone|indptr"):
        gpu_library.linsys_solve(bad_row, P["G_col"], P["G_val"], P["C_row"], P["C_col"], P["C_val"], P["g_val"],
                                 P["c_val"], P["input_lambda"], 1, 1e-6, 10, False, 1e-3)


def test_built_library_is_not_older_than_its_sources():
    """The in-tree libgato_hip.so travels to the GPU box as built: a stale one would run old kernels there."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "gato_python_amd", "libgato_hip.so")
    if not os.path.exists(so):
        pytest.skip("library not built yet (the loader builds it on first use)")
    srcs = glob.glob(os.path.join(root, "gato_python_amd", "csrc", "*.hip")) + \
        glob.glob(os.path.join(root, "gato_python_amd", "csrc", "*.h")) + [os.path.join(root, "include", "gato_hip.h")]
    newest = max(srcs, key=os.path.getmtime)
    assert os.path.getmtime(so) >= os.path.getmtime(newest), f"{newest} is newer than libgato_hip.so: run make -C gato_python_amd/csrc"
