"""us per whole step at 50 and 100 iterations (slope = us per iteration in the step, intercept = everything else) and us per
iteration of PCG-only launches, for the library GATO_HIP_LIB names: python tools/step_slope.py [f32|f64]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from gato_python_amd import synth
from gato_python_amd.solver import Solver
from tune_pcg import run
tag = os.path.basename(os.environ.get("GATO_HIP_LIB", "libgato_hip.so"))
for name in (sys.argv[1:] or ["f32", "f64"]):
    dt = np.float32 if name == "f32" else np.float64
    s = synth.make_system(14, 7, 50, seed=0)
    sol = Solver(14, 7, 50, dt)
    dev = sol.upload_system(s)
    lam, dz = sol.new(700), sol.new(sol.N)
    t = {}
    for mi in (50, 100):
        for _ in range(50): sol.linsys(*dev, 0.0, mi, s.rho, lam, dz)
        torch.cuda.synchronize()
        best = 1e9
        for blk in range(6):
            t0 = time.perf_counter()
            for _ in range(400): sol.linsys(*dev, 0.0, mi, s.rho, lam, dz)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 400)
        t[mi] = best * 1e6
    sol.close()
    r = run(14, 7, 50, dt, reps=30)
    print(f"{tag} 14/7/50 {name}: step(100) {t[100]:.2f} us, step(50) {t[50]:.2f} us, slope {(t[100] - t[50]) / 50:.4f} us/iter, intercept {2 * t[50] - t[100]:.2f} us; "
          f"PCG-only launch/100 {r['us_per_iter']:.4f} us", flush=True)
