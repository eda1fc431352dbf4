"""DPP-row layout of the resident PCG launches (solver option dpp_rows) against the LDS-window form on the same box: bits of
(lambda, iters) and us per iteration (python tools/dpp_ab.py)."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tune_pcg import run
from gato_python_amd import synth
from gato_python_amd.solver import Solver

def solve(S, C, K, dt, dpp, tol, iters):
    sysm = synth.make_system(S, C, K, seed=0)
    sol = Solver(S, C, K, dt); sol.set_option("dpp_rows", dpp)
    dev = sol.upload_system(sysm)
    lam, dz = sol.new(S * K), sol.new(sol.N)
    sol.linsys(*dev, tol, iters, sysm.rho, lam, dz)
    torch.cuda.synchronize(); sol.check_status()
    import ctypes as ct
    buf = (ct.c_int * 1)()
    assert ct.CDLL("libamdhip64.so").hipMemcpy(buf, ct.c_void_p(sol.buffer_ptr(8)), 4, 2) == 0
    out = (lam.cpu().numpy().copy(), dz.cpu().numpy().copy(), int(buf[0]), sol.get_option("last_dpp"), sol.get_option("last_groups"), sol.get_option("last_threads"))
    sol.close()
    return out

cases = [(14, 7, 20, np.float64), (14, 7, 32, np.float64), (14, 7, 100, np.float64), (14, 7, 100, np.float32), (14, 7, 512, np.float32), (14, 7, 512, np.float64),
         (14, 7, 1024, np.float32), (14, 7, 1024, np.float64), (14, 7, 4096, np.float32), (14, 7, 4096, np.float64),
         (32, 16, 256, np.float32), (32, 16, 1024, np.float32), (32, 16, 1024, np.float64), (12, 6, 300, np.float32), (12, 6, 300, np.float64)]
for (S, C, K, dt) in cases:
    for tol, iters in ((1e-8 if dt == np.float64 else 1e-5, 200), (0.0, 25)):
        a = solve(S, C, K, dt, 0, tol, iters); b = solve(S, C, K, dt, 1, tol, iters)
        same = a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes() and a[2] == b[2]
        print(f"{S}/{C}/{K} {np.dtype(dt).name} tol={tol} iters {a[2]}/{b[2]}  lds {a[4]}x{a[5]} dpp({b[3]}) {b[4]}x{b[5]}  bits {'IDENTICAL' if same else 'DIFFER max|dlam|=%g' % np.abs(a[0]-b[0]).max()}", flush=True)
    ra = run(S, C, K, dt, reps=20, opts={"dpp_rows": 0}); rb = run(S, C, K, dt, reps=20, opts={"dpp_rows": 1})
    ra2 = run(S, C, K, dt, reps=20, opts={"dpp_rows": 0}); rb2 = run(S, C, K, dt, reps=20, opts={"dpp_rows": 1})
    print(f"    us/iter: lds {ra['us_per_iter']:.3f} / {ra2['us_per_iter']:.3f} ({ra['groups']}x{ra['threads']})   dpp {rb['us_per_iter']:.3f} / {rb2['us_per_iter']:.3f} ({rb['groups']}x{rb['threads']})", flush=True)
