"""Transposed S / Pinv images for the one-workgroup kernels (written by the fused assembly launch; option no_image = 1 loads from
the bd arrays as before): the same bits, and us per whole step (assembly -> PCG -> dz) either way.  python tools/img_check.py"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gato_python_amd import synth
from gato_python_amd.solver import Solver

bad = 0
for dt in (np.float64, np.float32):
    for K in (50, 49, 37, 23, 9, 2, 1):
        s = synth.make_system(14, 7, K, seed=K) if K > 1 else synth.blocks_to_csr(*synth.make_blocks(14, 7, 1, 3, False))
        out = {}
        for ni in (1, 0):
            sol = Solver(14, 7, K, dt)
            sol.set_option("no_image", ni)
            dev = sol.upload_system(s)
            lam, dz = sol.new(14 * K), sol.new(sol.N)
            for rep in range(2):
                sol.linsys(*dev, 1e-9 if dt == np.float64 else 1e-5, 60, s.rho, lam, dz)
                torch.cuda.synchronize(); sol.check_status()
            out[ni] = (lam.cpu().numpy().copy(), dz.cpu().numpy().copy(), sol.get_option("last_image"), sol.get_option("last_pair"))
            sol.close()
        same = np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and np.isfinite(out[0][0]).all()
        bad += not same
        print(f"{np.dtype(dt).name} K={K} pair={out[0][3]} image used={out[0][2]}/{out[1][2]}: {'same bits' if same else 'DIFFERENT'}", flush=True)
print("MISMATCHES", bad)
for dt in (np.float64, np.float32):
    s = synth.make_system(14, 7, 50, seed=0)
    for rep in range(2):
        for ni in (1, 0):
            sol = Solver(14, 7, 50, dt)
            sol.set_option("no_image", ni)
            dev = sol.upload_system(s)
            lam, dz = sol.new(700), sol.new(sol.N)
            for _ in range(50): sol.linsys(*dev, 0.0, 100, s.rho, lam, dz)
            torch.cuda.synchronize()
            best = 1e9
            for blk in range(5):
                t0 = time.perf_counter()
                for _ in range(400): sol.linsys(*dev, 0.0, 100, s.rho, lam, dz)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 400)
            print(f"14/7/50 {np.dtype(dt).name} no_image={ni}: {best * 1e6:.2f} us per step", flush=True)
            sol.close()
