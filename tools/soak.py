"""Soak: many back-to-back solves (resident + single-reduction kernels, several geometries; the one-workgroup kernels whose helper
blocks do dz), every result - lambda AND dz - compared bit for bit with the first one and the status word checked: the
inter-workgroup hand-off must never lose or reorder a granule, a helper block must never read lambda before it is complete.
      python tools/soak.py [launches per case] [launches of the one-workgroup cases]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gato_python_amd import synth
from gato_python_amd.solver import Solver

n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n1 = int(sys.argv[2]) if len(sys.argv) > 2 else 4 * n0
bad = 0
for (S, C, K, dt, opts) in [(14, 7, 512, np.float32, {}), (14, 7, 1024, np.float32, {}), (14, 7, 512, np.float64, {}),   # one XCD: workgroup-scope granules
                            (14, 7, 512, np.float32, dict(xcd_pack=2)), (14, 7, 4096, np.float32, {}), (14, 7, 4096, np.float64, {}),
                            (32, 16, 1024, np.float32, {}), (14, 7, 4096, np.float32, dict(pcg_variant=1)),
                            (14, 7, 700, np.float64, dict(pcg_threads=256)), (14, 7, 50, np.float64, {}), (14, 7, 41, np.float64, {}),
                            (14, 7, 50, np.float32, {}), (14, 7, 73, np.float32, {}), (14, 7, 19, np.float32, {}), (14, 7, 37, np.float32, {}),   # helper blocks do dz; fp32: wave-private windows, transposed images
                            (14, 7, 20000, np.float32, {}), (14, 7, 15000, np.float64, {}),     # semi-resident launches
                            # round 4: runtime co-residency, the smallest K of the mixed-rows fp64 kernel, fp64 LDS-DMA ring (auto since round 5)
                            (14, 7, 4096, np.float32, dict(coop_launch=1)), (14, 7, 33, np.float64, {}), (14, 7, 65536, np.float64, {})]:
    s = synth.make_system(S, C, K, seed=3)
    sol = Solver(S, C, K, dt)
    for k_, v in opts.items():
        sol.set_option(k_, v)
    dev = sol.upload_system(s)
    lam, dz = sol.new(S * K), sol.new(sol.N)
    sol.linsys(*dev, 0.0, 30, s.rho, lam, dz)
    sol.check_status()
    ref, ref_dz = lam.clone(), dz.clone()
    n = n1 if sol.get_option("last_dz_fused") == 2 else n0
    t0 = time.time()
    mism = 0
    for i in range(n):
        if i % 50 == 49:
            dz.fill_(float("nan"))                       # a dz that is not (re)written shows
        sol.linsys(*dev, 0.0, 30, s.rho, lam, dz)
        if i % 50 == 49:
            sol.check_status()
            if not (torch.equal(lam, ref) and torch.equal(dz, ref_dz)):
                mism += 1
    torch.cuda.synchronize()
    sol.check_status()
    mism += 0 if (torch.equal(lam, ref) and torch.equal(dz, ref_dz)) else 1
    bad += mism
    print(f"{S}/{C}/{K} {np.dtype(dt).name} {opts}: {n} solves x 30 iterations, groups={sol.get_option('last_groups')}, "
          f"dz fused {sol.get_option('last_dz_fused')}, {(time.time() - t0) / n * 1e6:.0f} us per solve, mismatching checks: {mism}", flush=True)
    sol.close()
print("SOAK", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
