"""(scratch) reads the ad-hoc hand-off stamps of a -DHS build: GATO_HIP_LIB=.../libgato_hip_hs.so python tools/hs_read.py"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gato_python_amd import synth
from gato_python_amd.solver import Solver
names = ["wave sum + own stores", "gather + own total + offsets", "poll + decode", "wave sum + bc write", "B2 + bc read", "B1 (gathered form)", "update + window barrier + product", "-"]
for (S, C, K, dt, dpp) in [(14, 7, 512, np.float32, 0), (14, 7, 512, np.float64, 1), (14, 7, 4096, np.float32, 0), (14, 7, 4096, np.float32, 1), (14, 7, 4096, np.float64, 1), (32, 16, 1024, np.float32, 0)]:
    s = synth.make_system(S, C, K, seed=0)
    sol = Solver(S, C, K, dt); sol.set_option("dpp_rows", dpp); sol.set_option("record_eta", 1)
    dev = sol.upload_system(s); lam, dz = sol.new(S * K), sol.new(sol.N)
    for _ in range(3): sol.linsys(*dev, 0.0, 100, s.rho, lam, dz)
    torch.cuda.synchronize()
    h = sol.eta_history(210)[200:208]
    tot = h.sum()
    print(f"{S}/{C}/{K} {np.dtype(dt).name} dpp={sol.get_option('last_dpp')} W={sol.get_option('last_groups')}x{sol.get_option('last_threads')}: {tot/100:.0f} cycles per iteration (wave 0 of workgroup 0)")
    for n, c in zip(names, h):
        if n != "-": print(f"    {n:36s} {c/200:7.0f} cycles per hand-off")
    sol.close()
