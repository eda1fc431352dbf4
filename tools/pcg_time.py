"""us per PCG iteration of the default launch for a few bench shapes (A/B of kernel changes): python tools/pcg_time.py"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tune_pcg import run
for (S, C, K, dt) in ((14, 7, 50, np.float64), (14, 7, 50, np.float32), (14, 7, 512, np.float32), (14, 7, 4096, np.float32),
                      (14, 7, 4096, np.float64), (32, 16, 1024, np.float32), (14, 7, 1024, np.float32), (14, 7, 1024, np.float64), (32, 16, 256, np.float32)):
    opts = dict(kv.split("=") for kv in sys.argv[1:])
    r = run(S, C, K, dt, reps=20, opts={k: int(v) for k, v in opts.items()})
    print(f"{S}/{C}/{K} {np.dtype(dt).name}: {r['us_per_iter']:.3f} us/iter  ({r['groups']} x {r['threads']}) {opts}", flush=True)
