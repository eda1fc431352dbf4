"""Whole-step time with the fused assembly launch (asm_mode=2) against the stage kernels (asm_mode=1)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gato_python_amd import synth
from gato_python_amd.solver import Solver

def run(S, C, K, dt, B, mode, iters=10, reps=30):
    systems = [synth.make_system(S, C, K, seed=b) for b in range(min(B, 4))]
    systems = [systems[b % len(systems)] for b in range(B)]
    sol = Solver(S, C, K, dt, batch=B); sol.set_option("asm_mode", mode)
    lam, dz = sol.new(B * S * K), sol.new(B * sol.N)
    if B == 1:
        dev = sol.upload_system(systems[0]); call = lambda: sol.linsys(*dev, 0.0, iters, systems[0].rho, lam, dz)
    else:
        dev = sol.upload_batch(systems); call = lambda: sol.linsys_batched(*dev, 0.0, iters, systems[0].rho, lam, dz)
    for _ in range(5): call()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(reps): call()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / reps)
    sol.close()
    return best * 1e6

if __name__ == "__main__":
  for (S, C, K, dt, B) in [(14, 7, 50, np.float64, 1), (14, 7, 50, np.float32, 1), (14, 7, 512, np.float32, 1), (14, 7, 2048, np.float32, 1),
                         (14, 7, 4096, np.float32, 1), (14, 7, 4096, np.float64, 1), (32, 16, 50, np.float64, 1), (32, 16, 1024, np.float32, 1),
                         (14, 7, 50, np.float64, 8), (14, 7, 50, np.float64, 64), (14, 7, 50, np.float64, 512), (14, 7, 50, np.float32, 512),
                         (2, 1, 5, np.float32, 1), (14, 7, 16384, np.float32, 1)]:
    a, b = run(S, C, K, dt, B, 1), run(S, C, K, dt, B, 2)
    print(f"{S}/{C}/{K} {np.dtype(dt).name} B={B}: stage kernels {a:8.1f} us   fused {b:8.1f} us   (10 PCG iterations included)", flush=True)
