"""Timing-only ablations of the resident PCG iteration (option `ablate`, diagnostic builds): which phase costs what."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gato_python_amd import synth
from gato_python_amd.solver import Solver
for (S, C, K, dt, thr) in [(14, 7, 50, np.float32, 0), (14, 7, 50, np.float64, 512), (14, 7, 4096, np.float32, 512)]:
    s = synth.make_system(S, C, K, seed=0)
    sol = Solver(S, C, K, dt); sol.set_option("pcg_threads", thr)
    dev = sol.upload_system(s); lam, dz = sol.new(S * K), sol.new(sol.N)
    sol.linsys(*dev, 0.0, 100, s.rho, lam, dz); torch.cuda.synchronize()
    sol.set_option("time_pcg", 1)
    sol.set_option("stamp_pcg", 2)      # the ablation switches exist only in the diagnostic builds (2 = switches alone, no cycle stamps)
    b = [sol.buffer_ptr(i) for i in (3, 4, 5)]
    for abl, name in [(0, "full"), (1, "-spmv1"), (3, "-spmv1-spmv2"), (4, "-reductions"), (8, "-B3B6"), (12, "-red-B3B6"), (15, "nothing")]:
        sol.set_option("ablate", abl)
        ms = []
        for i in range(8):
            sol.pcg(b[0], b[1], b[2], 0.0, 100, lam=lam, check=False); ms.append(sol.pcg_last_ms())
        print(f"{S}/{C}/{K} {np.dtype(dt).name} W={sol.get_option('last_groups')} {name:14s} {1e3*np.median(ms[2:])/100:.3f} us/iter", flush=True)
    sol.close()
