"""Streaming PCG at sizes beyond register residency: HBM-bound regime.  us/iteration and algorithmic GB/s."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gato_python_amd import synth, _lib
from gato_python_amd.solver import Solver

def run(S, C, K, dt, iters=20, reps=5, mode=2):
    t0 = time.time()
    sysm = synth.make_system(S, C, K, seed=0)
    t1 = time.time()
    sol = Solver(S, C, K, dt)
    sol.set_option("pcg_mode", mode)
    dev = sol.upload_system(sysm)
    lam, dz = sol.new(S * K), sol.new(sol.N)
    sol.linsys(*dev, 0.0, iters, sysm.rho, lam, dz)
    torch.cuda.synchronize(); sol.check_status()
    sol.set_option("time_pcg", 1)
    b = [sol.buffer_ptr(i) for i in (3, 4, 5)]
    ms = []
    for i in range(reps):
        sol.pcg(b[0], b[1], b[2], 0.0, iters, lam=lam, check=False); ms.append(sol.pcg_last_ms())
    w = np.dtype(dt).itemsize
    b_iter = ((6 * K - 4) * S * S + 13 * S * K) * w
    us = 1e3 * float(np.median(ms)) / iters
    r = dict(S=S, K=K, dtype=np.dtype(dt).name, mode=sol.get_option("last_mode"), us_per_iter=us,
             algorithmic_GBps=b_iter / us / 1e3, frac_8TBs=b_iter / us / 1e3 / 8000, gen_s=round(t1 - t0, 1))
    sol.close()
    return r

if __name__ == "__main__":
    cases = [(14, 7, 16384, np.float32), (14, 7, 65536, np.float32), (14, 7, 131072, np.float32),
             (14, 7, 65536, np.float64), (32, 16, 16384, np.float32)]
    if len(sys.argv) > 1 and sys.argv[1] == "quick":
        cases = [(14, 7, 131072, np.float32), (14, 7, 65536, np.float64), (32, 16, 16384, np.float32)]
    for (S, C, K, dt) in cases:
        print(json.dumps(run(S, C, K, dt)), flush=True)
