"""Whole sharded step of ONE rank (gato_cluster_linsys: sharded assembly + the rank's persistent launch + dz, one call) against the
iteration loop it contains: what a rank of K_system / N knots spends outside its loop, stage kernels against the fused assembly
launch.  One rank alone (a cluster of one), so the launch never waits for a peer.  python tools/cluster_step_time.py [K ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gato_python_amd import synth
from gato_python_amd.dist import ClusterPCG
from gato_python_amd.solver import Solver

if __name__ == "__main__":
    Ks = [int(a) for a in sys.argv[1:]] or [512, 256, 1024]
    for K in Ks:
        for dt in (np.float32, np.float64):
            s = synth.make_system(14, 7, K, seed=0)
            for variant in (0, 1):
                for asm in (1, 2):
                    sol = Solver(14, 7, K, dt)
                    sol.set_option("pcg_variant", variant)
                    sol.set_option("asm_mode", asm)
                    cl = ClusterPCG(sol, 0, 1, inprocess_peers=True)
                    ClusterPCG.connect_inprocess([cl])
                    d = sol.upload_system(s)
                    lam, dz, it = sol.new(14 * K), sol.new(sol.N), sol.new(1, torch.int32)
                    for _ in range(10):
                        cl.linsys(d, 0.0, 100, s.rho, lam, dz, it)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    n = 200
                    for _ in range(n):
                        cl.linsys(d, 0.0, 100, s.rho, lam, dz, it)
                    torch.cuda.synchronize()
                    step = 1e6 * (time.perf_counter() - t0) / n
                    sol.set_option("time_pcg", 1)
                    ms = []
                    for _ in range(8):
                        cl.linsys(d, 0.0, 100, s.rho, lam, dz, it)
                        ms.append(sol.pcg_last_ms())
                    loop = 1e3 * float(np.median(ms[2:]))
                    sol.check_status()
                    print(f"14/7/{K} {np.dtype(dt).name} variant {sol.get_option('last_variant')} asm {'fused' if sol.get_option('last_asm_fused') else 'stages'}: "
                          f"step {step:.1f} us, launch {loop:.1f} us, outside the loop {step - loop:.1f} us", flush=True)
                    cl.close(); sol.close()
