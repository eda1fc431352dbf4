"""Assembly time (us, HIP events around the assembly of the whole-solve entry) by asm_mode: 1 = stage kernels, 2 = one launch
with a workgroup per knot, 3 = chunked launch (asm_chunk knots per workgroup).  usage: asm_modes.py [asm_chunk ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gato_python_amd import synth
from gato_python_amd.solver import Solver

chunks = [int(x) for x in sys.argv[1:]] or [0]
cases = [(14, 7, 50, np.float64, 512), (14, 7, 50, np.float32, 512), (14, 7, 4096, np.float32, 1), (14, 7, 4096, np.float64, 1),
         (32, 16, 1024, np.float32, 1), (14, 7, 512, np.float32, 1)]
for S, C, K, dt, B in cases:
    base = synth.make_system(S, C, K, seed=0)
    out = []
    for mode, ch in [(1, 0), (2, 0)] + [(3, c) for c in chunks]:
        if mode == 2 and K * B > 8192:
            continue
        sol = Solver(S, C, K, dt, batch=B)
        d = sol.upload_batch([base] * B) if B > 1 else sol.upload_system(base)
        sol.set_option("asm_mode", mode); sol.set_option("asm_chunk", ch); sol.set_option("time_stages", 1)
        lam, dz = sol.new(B * S * K), sol.new(B * sol.N)
        call = (lambda: sol.linsys_batched(*d, 0.0, 10, base.rho, lam, dz)) if B > 1 else (lambda: sol.linsys(*d, 0.0, 10, base.rho, lam, dz))
        for _ in range(4):
            call()
        st = []
        for _ in range(10):
            call(); st.append(sol.last_stage_ms())
        out.append(f"mode {mode}" + (f" chunk {ch}" if mode == 3 else "") + f": {1e3 * np.median([x['assembly'] for x in st]):7.1f}")
        sol.close()
    print(f"{S}/{C}/{K} {np.dtype(dt).name} B={B}: " + " | ".join(out), flush=True)
