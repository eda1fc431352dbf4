"""Diagnostic: where one PCG iteration of the resident kernel spends its cycles (wave 0 of workgroup 0)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as ct
from gato_python_amd import synth
from gato_python_amd.solver import Solver
names = ["spmv S.p", "reduce v (+handoff)", "update lam,r + barrier", "spmv Pinv.r", "reduce eta (+handoff)", "update p + barrier",
         "  both hand-offs: entry -> past B1", "  both hand-offs: own total + poll"]
for (S, C, K, dt, thr) in [(14, 7, 50, np.float32, 0), (14, 7, 50, np.float64, 0), (14, 7, 512, np.float32, 512), (14, 7, 4096, np.float32, 512)]:
    s = synth.make_system(S, C, K, seed=0)
    sol = Solver(S, C, K, dt); sol.set_option("pcg_threads", thr); sol.set_option("stamp_pcg", 1); sol.set_option("no_pair", 1)
    dev = sol.upload_system(s); lam, dz = sol.new(S * K), sol.new(sol.N)
    iters = 100
    for _ in range(2): sol.linsys(*dev, 0.0, iters, s.rho, lam, dz)
    torch.cuda.synchronize()
    buf = (ct.c_ulonglong * 10)()
    torch.cuda.synchronize()
    import ctypes
    hip = ct.CDLL("libamdhip64.so")
    hip.hipMemcpy(buf, ct.c_void_p(sol.buffer_ptr(9)), 80, 2)
    v = list(buf)
    clk = v[8] / v[9] * 100e6
    print(f"{S}/{C}/{K} {np.dtype(dt).name} groups={sol.get_option('last_groups')} threads={sol.get_option('last_threads')}: total {v[8]} cyc, "
          f"{v[9]/100:.1f} us, clock {clk/1e9:.2f} GHz, per-iter {v[8]/iters:.0f} cyc")
    for n, c in zip(names, v[:8]): print(f"    {n:28s} {c/iters:8.0f} cyc/iter")
    sol.close()
