"""One whole solve step a few times (for rocprofv3 --kernel-trace + tools/trace_gaps.py): S C K B dtype [asm_mode]."""
import sys, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from gato_python_amd import synth
from gato_python_amd.solver import Solver
S, C, K, B = [int(x) for x in sys.argv[1:5]]
dt = np.float32 if sys.argv[5] == "f32" else np.float64
base = synth.make_system(S, C, K, seed=0)
sol = Solver(S, C, K, dt, batch=B)
if len(sys.argv) > 6: sol.set_option("asm_mode", int(sys.argv[6]))
lam, dz = sol.new(B * S * K), sol.new(B * sol.N)
if B == 1:
    dev = sol.upload_system(base); call = lambda: sol.linsys(*dev, 0.0, 10, base.rho, lam, dz)
else:
    dev = sol.upload_batch([base] * B); call = lambda: sol.linsys_batched(*dev, 0.0, 10, base.rho, lam, dz)
for _ in range(6): call()
torch.cuda.synchronize()
