"""us per PCG iteration of the wave-published resident launch by workgroup size (fewer, larger workgroups = fewer participants
in the all-to-all, more waves per sweep)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tune_pcg import run
for (S, C, K, dt) in ((14, 7, 512, np.float32), (14, 7, 1024, np.float32), (14, 7, 512, np.float64), (14, 7, 1024, np.float64), (32, 16, 256, np.float32)):
    out = []
    for t in (0, 256, 320, 384, 448, 512, 576, 640, 704, 768):
        r = run(S, C, K, dt, threads=t, reps=20)
        if r: out.append(f"{t}: {r['us_per_iter']:.3f} ({r['groups']}x{r['threads']})")
    print(f"{S}/{C}/{K} {np.dtype(dt).name}: " + " | ".join(out), flush=True)
