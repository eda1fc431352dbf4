"""K = 50 (the headline shape) through multi-workgroup one-XCD launches instead of the one-workgroup kernel: us per
iteration by workgroup count (python tools/k50_groups.py)."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tune_pcg import run
for dt in (np.float64, np.float32):
    for K in (50, 100, 128):
        r = run(14, 7, K, dt, reps=20)
        print(f"14/7/{K} {np.dtype(dt).name} default: {json.dumps(r)}", flush=True)
        for g in (2, 3, 4, 5, 7, 10, 13, 17, 25):
            if g > K: continue
            for v in (0, 1):
                r = run(14, 7, K, dt, groups=g, reps=20, opts={"pcg_variant": v})
                if r: print(f"14/7/{K} {np.dtype(dt).name} groups={g} variant={v}: {json.dumps(r)}", flush=True)
