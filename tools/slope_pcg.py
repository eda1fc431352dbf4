"""Launch time of the one-workgroup PCG kernels as a function of the iteration count: slope = us per iteration, intercept =
what a launch costs before and after its loop (HIP events around the launch)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tune_pcg import run
for dt in (np.float64, np.float32):
    for opts in ({}, {"no_pair": 1}):
        t = {}
        for it in (1, 100, 200, 400):
            r = run(14, 7, 50, dt, iters=it, reps=20, opts=opts)
            t[it] = r["us_per_iter"] * it
        slope = (t[400] - t[100]) / 300
        print(np.dtype(dt).name, opts, {k: round(v, 1) for k, v in t.items()}, "slope us/iter", round(slope, 3), "intercept us", round(t[100] - 100 * slope, 1), flush=True)
