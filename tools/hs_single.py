"""(scratch build with s_memrealtime stamps in pcg_single_f64m_kernel; option ablate bits 8..15 = a sleep of that many x 512 cycles
before the matrix loads) where the one-workgroup fp64 launch spends its time outside the loop."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gato_python_amd import synth
from gato_python_amd.solver import Solver
S, C, K = 14, 7, 50
s = synth.make_system(S, C, K, seed=0)
sol = Solver(S, C, K, np.float64); sol.set_option("record_eta", 1); sol.set_option("time_pcg", 1)
dev = sol.upload_system(s); lam, dz = sol.new(S * K), sol.new(sol.N)
for sl in (0, 2, 4, 6, 8, 12, 16, 0):
    sol.set_option("ablate", sl << 8)
    for _ in range(3): sol.linsys(*dev, 0.0, 100, s.rho, lam, dz)
    torch.cuda.synchronize()
    h = sol.eta_history(210)[200:206] / 100.0     # 100 MHz ticks -> us
    print(f"sleep {sl} x 512 cycles: events around the launch {1e3*sol.pcg_last_ms():.1f} us; in the kernel (us since its first instruction): matrices loaded {h[1]:.2f}, windows zeroed {h[2]:.2f}, loop entered {h[3]:.2f}, loop left {h[4]:.2f}, lambda stored {h[5]:.2f}")
sol.close()
