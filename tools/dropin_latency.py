"""End-to-end latency of the drop-in call gpu_library.linsys_solve (host lists in, host lists out).
python tools/dropin_latency.py            the ctypes mirror at the repo root (gpu_library.py)
python tools/dropin_latency.py pybind11   the pybind11 module a maintainer of the reference would build (bindings/pybind11)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GATO_VERBOSE"] = "0"
PYB = len(sys.argv) > 1 and sys.argv[1] == "pybind11"
if PYB:
    sys.path.insert(0, os.path.join(ROOT, "bindings", "pybind11", "build"))
import gpu_library
assert PYB == gpu_library.__file__.endswith(".so"), gpu_library.__file__
print("binding:", "pybind11 module" if PYB else "ctypes mirror", gpu_library.__file__)
from gato_python_amd import synth
if PYB:
    class _Stats:                                  # the pybind11 module has the reference's surface only
        @staticmethod
        def last_stats():
            return {"ms": [float("nan")], "iters": -1}
    gpu_library.last_stats = _Stats.last_stats
for (S, C, K) in [(2, 1, 5), (14, 7, 50), (14, 7, 512)]:
    s = synth.pendulum_system() if K == 5 else synth.make_system(S, C, K, seed=0)
    args_np = (s.G_row, s.G_col, s.G_val, s.C_row, s.C_col, s.C_val, s.g, s.c, np.zeros(S * K), 1, 1e-6, 100, False, s.rho)
    args_list = tuple(a.tolist() if isinstance(a, np.ndarray) else a for a in args_np)
    for name, args in (("numpy", args_np), ("lists", args_list)):
        for _ in range(3): gpu_library.linsys_solve(*args)
        t0 = time.perf_counter(); n = 30
        for _ in range(n): gpu_library.linsys_solve(*args)
        dt = (time.perf_counter() - t0) / n
        st = gpu_library.last_stats()
        print(f"{S}/{C}/{K} {name}: {dt*1e3:.3f} ms per call (device-timed solve {st['ms'][0]:.3f} ms, iters {st['iters']})", flush=True)

if PYB:
    sys.exit(0)
# PCIe-inclusive rate of the bench workload (BASELINE configs[1]: 14/7/50, f64, exactly 100 iterations)
gpu_library.set_precision("f64")
s = synth.make_system(14, 7, 50, seed=0)
args = (s.G_row, s.G_col, s.G_val, s.C_row, s.C_col, s.C_val, s.g, s.c, np.zeros(700), 1, 0.0, 100, False, s.rho)
for _ in range(5): gpu_library.linsys_solve(*args)
t0 = time.perf_counter(); n = 50
for _ in range(n): gpu_library.linsys_solve(*args)
dt = (time.perf_counter() - t0) / n
print(f"bench workload through the host boundary (numpy in, lists out): {dt*1e3:.3f} ms per call = {100/dt:.0f} PCG iterations/s PCIe-inclusive "
      f"(device-timed {gpu_library.last_stats()['ms'][0]:.3f} ms)")
