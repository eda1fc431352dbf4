"""Is a whole step host-bound?  us per step (assembly -> PCG -> dz, enqueue loop as in bench.py) against the iteration count:
the intercept is what the host + the fixed part of the launches cost.  python tools/host_bound.py"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gato_python_amd import synth
from gato_python_amd.solver import Solver
for dt in (np.float64, np.float32):
    s = synth.make_system(14, 7, 50, seed=0)
    sol = Solver(14, 7, 50, dt)
    dev = sol.upload_system(s)
    lam, dz = sol.new(700), sol.new(sol.N)
    for mi in (1, 10, 50, 100):
        for _ in range(50): sol.linsys(*dev, 0.0, mi, s.rho, lam, dz)
        torch.cuda.synchronize()
        best = 1e9
        for blk in range(5):
            t0 = time.perf_counter()
            for _ in range(400): sol.linsys(*dev, 0.0, mi, s.rho, lam, dz)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 400)
        print(f"14/7/50 {np.dtype(dt).name} max_iters={mi}: {best * 1e6:.2f} us per step (enqueue alone {1e6 * (t1 - t0) / 400:.2f})", flush=True)
    sol.close()
