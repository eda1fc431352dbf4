"""Same-box A/B of the default recurrence against the single-reduction recurrence (option pcg_variant = 1), alternating: us per PCG
iteration of the launch and us per whole step (assembly + PCG of 100 iterations + dz).  At 14/7/50 fp64 variant 1 is the
mixed-rows kernel's own single-reduction form (pcg_single_f64m_kernel<..., CG1>).   python tools/variant_ab.py [rounds]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tune_pcg import run
from gato_python_amd import synth
from gato_python_amd.solver import Solver


def step_us(S, C, K, dt, variant, steps=300):
    sysm = synth.make_system(S, C, K, seed=0)
    sol = Solver(S, C, K, dt)
    sol.set_option("pcg_variant", variant)
    dev = sol.upload_system(sysm)
    lam, dz = sol.new(S * K), sol.new(sol.N)
    for _ in range(20):
        sol.linsys(*dev, 0.0, 100, sysm.rho, lam, dz)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sol.linsys(*dev, 0.0, 100, sysm.rho, lam, dz)
    torch.cuda.synchronize()
    us = 1e6 * (time.perf_counter() - t0) / steps
    sol.check_status()
    v = sol.get_option("last_variant"), sol.get_option("last_pair")
    sol.close()
    return us, v


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    for (S, C, K, dt) in ((14, 7, 50, np.float64), (14, 7, 50, np.float32), (14, 7, 37, np.float64)):
        acc = {0: [], 1: []}
        st = {0: [], 1: []}
        for _ in range(rounds):
            for v in (0, 1):
                acc[v].append(run(S, C, K, dt, reps=20, opts={"pcg_variant": v})["us_per_iter"])
                u, info = step_us(S, C, K, dt, v)
                st[v].append(u)
        print(f"{S}/{C}/{K} {np.dtype(dt).name}: default {np.median(acc[0]):.3f} us/iter ({min(acc[0]):.3f}..{max(acc[0]):.3f}), step {np.median(st[0]):.1f} us"
              f" = {1e8 / np.median(st[0]):.0f} it/s | single-reduction {np.median(acc[1]):.3f} us/iter ({min(acc[1]):.3f}..{max(acc[1]):.3f}), "
              f"step {np.median(st[1]):.1f} us = {1e8 / np.median(st[1]):.0f} it/s", flush=True)
