"""Wave-private operand windows of the one-workgroup kernels (option shared_windows = 0, default) against the shared-window
form (shared_windows = 1): the same bits (lambda, dz, iters, eta history) and us per PCG iteration.  python tools/pw_check.py"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from gato_python_amd import synth
from gato_python_amd.solver import Solver
from tune_pcg import run

bad = 0
for dt in (np.float64, np.float32):
    for K in (50, 49, 41, 37, 23, 10, 5, 3, 2):
        for (tol, mi) in ((0.0, 30), (1e-8 if dt == np.float64 else 1e-4, 200)):
            for warm in (0, 1):
                s = synth.make_system(14, 7, K, seed=K)
                out = {}
                for sw in (1, 0):
                    sol = Solver(14, 7, K, dt)
                    sol.set_option("shared_windows", sw)
                    sol.set_option("record_eta", 1)
                    if warm:
                        sol.set_option("true_warm_start", 1)
                    dev = sol.upload_system(s)
                    lam, dz = sol.new(14 * K), sol.new(sol.N)
                    if warm:
                        lam.copy_(torch.from_numpy(np.random.default_rng(K).standard_normal(14 * K).astype(dt)))
                    for rep in range(2):
                        if rep == 0 and warm:
                            lam0 = lam.clone()
                        if rep == 1 and warm:
                            lam.copy_(lam0)
                        sol.linsys(*dev, tol, mi, s.rho, lam, dz)
                        torch.cuda.synchronize(); sol.check_status()
                    out[sw] = (lam.cpu().numpy().copy(), dz.cpu().numpy().copy(), sol.get_option("last_pair"))
                    sol.close()
                same = np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
                fin = np.isfinite(out[0][0]).all()
                if not (same and fin):
                    bad += 1
                print(f"{np.dtype(dt).name} K={K} tol={tol} warm={warm} pair={out[0][2]}: {'same bits' if same else 'DIFFERENT'}"
                      f"{'' if fin else ' NON-FINITE'}", flush=True)
# batches: one workgroup per system
for dt in (np.float64, np.float32):
    for K in (50, 17):
        B = 24
        systems = [synth.make_system(14, 7, K, seed=100 + b) for b in range(B)]
        out = {}
        for sw in (1, 0):
            sol = Solver(14, 7, K, dt, batch=B)
            sol.set_option("shared_windows", sw)
            dev = sol.upload_batch(systems)
            lam, dz = sol.new(B * 14 * K), sol.new(B * sol.N)
            sol.linsys_batched(*dev, 0.0, 25, systems[0].rho, lam, dz)
            torch.cuda.synchronize(); sol.check_status()
            out[sw] = (lam.cpu().numpy().copy(), dz.cpu().numpy().copy())
            sol.close()
        same = np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
        if not same:
            bad += 1
        print(f"batch {B} x K={K} {np.dtype(dt).name}: {'same bits' if same else 'DIFFERENT'}", flush=True)
print("MISMATCHES", bad)
for dt in (np.float64, np.float32):
    for rep in range(2):
        for sw in (1, 0):
            r = run(14, 7, 50, dt, reps=30, opts={"shared_windows": sw})
            print(f"14/7/50 {np.dtype(dt).name} shared_windows={sw}: {r['us_per_iter']:.4f} us/iter", flush=True)
