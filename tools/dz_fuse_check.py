"""Whole step of one 14/7/50 system with dz as a launch of its own (default for one system) and in the PCG launch's epilogue
(option no_fuse_dz = -1): us per step."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gato_python_amd import synth
from gato_python_amd.solver import Solver


def main():
    for dt in (np.float64, np.float32):
        S, C, K = 14, 7, 50
        s = synth.make_system(S, C, K, seed=0)
        out = {}
        for fuse in (0, -1):
            sol = Solver(S, C, K, dt)
            sol.set_option("no_fuse_dz", fuse)
            if dt == np.float32:
                sol.set_option("no_pair", 1)               # the fp32 pair kernel has no dz epilogue
            dev = sol.upload_system(s)
            lam, dz = sol.new(S * K), sol.new(sol.N)
            for _ in range(10):
                sol.linsys(*dev, 0.0, 100, s.rho, lam, dz)
            torch.cuda.synchronize()
            n = 1000
            t0 = time.perf_counter()
            for _ in range(n):
                sol.linsys(*dev, 0.0, 100, s.rho, lam, dz)
            torch.cuda.synchronize()
            out[fuse] = ((time.perf_counter() - t0) / n * 1e6, sol.get_option("last_dz_fused"), dz.clone())
            sol.close()
        print(f"{np.dtype(dt).name}: dz launch {out[0][0]:.1f} us per step (fused={out[0][1]}); dz in the PCG epilogue {out[-1][0]:.1f} us (fused={out[-1][1]}); "
              f"same bits: {torch.equal(out[0][2], out[-1][2])}", flush=True)


if __name__ == "__main__":
    main()
