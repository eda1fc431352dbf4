"""Diagnostic: phase boundaries of the fused assembly launch (knot 2's workgroup, thread 0), 10 ns ticks."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as ct
from gato_python_amd import synth
from gato_python_amd.solver import Solver
names = ["gather CSR -> LDS", "write dense + 6 inversions", "Schur blocks + theta^-1 (x2)", "stair products"]
hip = ct.CDLL("libamdhip64.so")
for (S, C, K, dt) in [(14, 7, 50, np.float64), (14, 7, 50, np.float32), (32, 16, 50, np.float64), (14, 7, 512, np.float32)]:
    s = synth.make_system(S, C, K, seed=0)
    sol = Solver(S, C, K, dt); sol.set_option("stamp_asm", 1); sol.set_option("asm_mode", 2)
    dev = sol.upload_system(s); lam, dz = sol.new(S * K), sol.new(sol.N)
    for _ in range(3):
        sol.linsys(*dev, 0.0, 5, s.rho, lam, dz)
    torch.cuda.synchronize()
    buf = (ct.c_ulonglong * 8)()
    hip.hipMemcpy(buf, ct.c_void_p(sol.buffer_ptr(9)), 64, 2)
    v = list(buf)
    print(f"{S}/{C}/{K} {np.dtype(dt).name}: total {(v[4] - v[0]) / 100:.2f} us")
    for i, n in enumerate(names):
        print(f"    {n:28s} {(v[i + 1] - v[i]) / 100:7.2f} us")
    sol.close()
