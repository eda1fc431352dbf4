"""What the two window barriers (B3, B6) of an iteration cost: timing-only ablation 8 of the diagnostic build (python tools/b3b6.py)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gato_python_amd import synth
from gato_python_amd.solver import Solver
for (S, C, K, dt, dpp) in [(14, 7, 512, np.float32, 0), (14, 7, 512, np.float32, 1), (14, 7, 512, np.float64, 1), (14, 7, 4096, np.float32, 0),
                           (14, 7, 4096, np.float64, 1), (32, 16, 1024, np.float32, 1), (14, 7, 20, np.float64, 1)]:
    s = synth.make_system(S, C, K, seed=0)
    sol = Solver(S, C, K, dt); sol.set_option("dpp_rows", dpp)
    dev = sol.upload_system(s); lam, dz = sol.new(S * K), sol.new(sol.N)
    sol.linsys(*dev, 0.0, 100, s.rho, lam, dz); torch.cuda.synchronize()
    sol.set_option("time_pcg", 1); sol.set_option("stamp_pcg", 2)
    b = [sol.buffer_ptr(i) for i in (3, 4, 5)]
    out = []
    for abl in (0, 8, 4, 12):
        sol.set_option("ablate", abl)
        ms = []
        for i in range(12):
            sol.pcg(b[0], b[1], b[2], 0.0, 100, lam=lam, check=False); ms.append(sol.pcg_last_ms())
        out.append(1e3 * np.median(ms[2:]) / 100)
    print(f"{S}/{C}/{K} {np.dtype(dt).name} dpp={sol.get_option('last_dpp')} W={sol.get_option('last_groups')}x{sol.get_option('last_threads')}: full {out[0]:.3f}  without B3/B6 {out[1]:.3f}  | no hand-offs {out[2]:.3f}  no hand-offs, no B3/B6 {out[3]:.3f} us/iter", flush=True)
    sol.close()
