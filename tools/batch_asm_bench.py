"""Whole-step time of batched and large-K solves (10 PCG iterations): where the stage kernels' throughput shows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from asm_crossover import run
for (S, C, K, dt, B) in [(14, 7, 50, np.float64, 512), (14, 7, 50, np.float32, 512), (14, 7, 4096, np.float32, 1), (14, 7, 4096, np.float64, 1),
                         (32, 16, 1024, np.float32, 1), (14, 7, 50, np.float64, 64)]:
    print(f"{S}/{C}/{K} {np.dtype(dt).name} B={B}: {run(S, C, K, dt, B, 0):8.1f} us per step (10 PCG iterations)", flush=True)
