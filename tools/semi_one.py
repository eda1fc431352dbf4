"""A few persistent PCG launches at K = 131072 f32 (for rocprofv3 --pmc passes): argv = [K [pcg_semi]], pcg_semi as the solver
option (default: auto = the LDS-DMA ring at this size; 1 = semi-resident)."""
import sys, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gato_python_amd import synth
from gato_python_amd.solver import Solver
S, C, K, dt = 14, 7, int(sys.argv[1]) if len(sys.argv) > 1 else 131072, np.float32
s = synth.make_system(S, C, K, seed=0)
sol = Solver(S, C, K, dt)
if len(sys.argv) > 2:
    sol.set_option("pcg_semi", int(sys.argv[2]))
dev = sol.upload_system(s)
lam, dz = sol.new(S * K), sol.new(sol.N)
sol.linsys(*dev, 0.0, 10, s.rho, lam, dz)
b = [sol.buffer_ptr(i) for i in (3, 4, 5)]
for _ in range(3):
    sol.pcg(b[0], b[1], b[2], 0.0, 10, lam=lam, check=False)
torch.cuda.synchronize()
