"""Where the fp64 multi-workgroup launches lose their ~1 us per iteration against fp32: timing-only ablations of the diagnostic
build (which mirrors the production hand-off forms) for both types at the same geometry (python tools/f64_gap.py)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gato_python_amd import synth
from gato_python_amd.solver import Solver
for (S, C, K, groups) in [(14, 7, 512, 0), (14, 7, 4096, 0), (32, 16, 1024, 0), (32, 16, 256, 0)]:
    for dt, dpp in ((np.float32, -1), (np.float64, 0), (np.float64, 1)) if S == 14 else ((np.float32, 0), (np.float32, 1)):
        s = synth.make_system(S, C, K, seed=0)
        sol = Solver(S, C, K, dt); sol.set_option("pcg_groups", groups); sol.set_option("dpp_rows", dpp)
        dev = sol.upload_system(s); lam, dz = sol.new(S * K), sol.new(sol.N)
        sol.linsys(*dev, 0.0, 100, s.rho, lam, dz); torch.cuda.synchronize()
        sol.set_option("time_pcg", 1)
        b = [sol.buffer_ptr(i) for i in (3, 4, 5)]
        for diag, abl, name in [(0, 0, "production"), (2, 0, "diag full"), (2, 3, "-products"), (2, 4, "-hand-offs"), (2, 7, "-both"), (2, 15, "skeleton")]:
            sol.set_option("stamp_pcg", diag); sol.set_option("ablate", abl)
            ms = []
            for i in range(12):
                sol.pcg(b[0], b[1], b[2], 0.0, 100, lam=lam, check=False); ms.append(sol.pcg_last_ms())
            print(f"{S}/{C}/{K} {np.dtype(dt).name} dpp={sol.get_option('last_dpp')} W={sol.get_option('last_groups')}x{sol.get_option('last_threads')} {name:12s} {1e3*np.median(ms[2:])/100:.3f} us/iter", flush=True)
        sol.close()
