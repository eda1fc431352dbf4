"""us per PCG iteration at 14/7/50 with wave-private (default) and shared operand windows: python tools/pw_time.py [K ...]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tune_pcg import run
Ks = [int(x) for x in sys.argv[1:]] or [50]
for K in Ks:
    for dt in (np.float64, np.float32):
        for rep in range(2):
            for sw in (1, 0):
                r = run(14, 7, K, dt, reps=30, opts={"shared_windows": sw})
                print(f"{os.environ.get('GATO_HIP_LIB', 'default')[-12:]} 14/7/{K} {np.dtype(dt).name} shared_windows={sw}: {r['us_per_iter']:.4f} us/iter", flush=True)
