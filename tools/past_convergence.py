"""fp32 error of the GPU PCG, the C oracle and the numpy oracle against the fp64 iterates after n fixed iterations, on a tiny
system (14/7/2: 28 unknowns) that CG solves in a handful of steps: up to convergence all three agree to rounding; past it the
iteration runs on rounding noise (eta, v -> 0) and the three summation orders drift apart by constant factors."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gato_python_amd import synth
from gato_python_amd.solver import Solver
from oracle import c_oracle as co, gato_oracle as o

def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / np.abs(b).max()

S, C, K = 14, 7, 2
s = synth.make_system(S, C, K, seed=1)
Gd, Cd = co.convert(*s.csr_args()[:6], S, C, K, s.rho, np.float32)
Sb, Pb, gam, _ = co.form_schur(Gd, Cd, s.g, s.c, S, C, K)
Pb = co.form_ss(Sb, Pb, S, K)
S64, P64, g64 = Sb.astype(np.float64), Pb.astype(np.float64), gam.astype(np.float64)
conv = co.pcg(S64, P64, g64, S, K, 1e-20, 200)[0]
for groups in (0, 2):
    sol = Solver(S, C, K, np.float32)
    sol.set_option("pcg_groups", groups)
    dS, dP, dg = sol.to_device(Sb), sol.to_device(Pb), sol.to_device(gam)
    print(f"pcg_groups={groups}")
    for n in (1, 2, 3, 4, 5, 6, 8, 10, 12, 15, 20, 40, 60):
        lam, it = sol.pcg(dS, dP, dg, 0.0, n)
        t = co.pcg(S64, P64, g64, S, K, 0.0, n)[0]
        lc, _, hist = co.pcg(Sb, Pb, gam, S, K, 0.0, n, return_history=True)
        ln = o.pcg(Sb, Pb, gam, S, K, 0.0, n)[0]
        g = lam.cpu().numpy()
        print(f"  n={n:3d} |eta|={abs(hist[min(n, len(hist)-1)]):.2e}  vs f64 iterate: gpu {rel(g,t):.2e} C {rel(lc,t):.2e} numpy {rel(ln,t):.2e}   vs converged: gpu {rel(g,conv):.2e} C {rel(lc,conv):.2e} numpy {rel(ln,conv):.2e}")
    sol.close()
