"""(scratch build: sleep before the first poll = bits 8..15 of option ablate, in units of 64 cycles) us per iteration by sleep."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tune_pcg import run
for (S, C, K, dt) in ((14, 7, 4096, np.float32), (14, 7, 3000, np.float32), (14, 7, 2048, np.float32), (14, 7, 1400, np.float32), (14, 7, 4096, np.float64), (14, 7, 3000, np.float64), (14, 7, 2048, np.float64), (14, 7, 1200, np.float64),
                      (32, 16, 1024, np.float32), (32, 16, 2048, np.float32), (32, 16, 600, np.float32), (32, 16, 1024, np.float64), (32, 16, 400, np.float64), (12, 6, 3000, np.float32)):
    out = []
    for sl in (8, 10, 12, 13, 14, 15, 16, 17, 18, 20):
        r = run(S, C, K, dt, reps=16, opts={"ablate": sl << 8})
        out.append(f"{sl}: {r['us_per_iter']:.3f}")
    print(f"{S}/{C}/{K} {np.dtype(dt).name} ({r['groups']}x{r['threads']}): " + "  ".join(out), flush=True)
