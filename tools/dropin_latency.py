"""End-to-end latency of the drop-in call gpu_library.linsys_solve (host lists in, host lists out)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GATO_VERBOSE"] = "0"
import gpu_library
from gato_python_amd import synth
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
