"""Sweep the resident-PCG launch geometry on one GPU: us per PCG iteration per (workload, threads, groups)."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gato_python_amd import synth, _lib
from gato_python_amd.solver import Solver

def run(S, C, K, dt, threads=0, groups=0, mode=0, iters=100, reps=10, opts=None):
    sysm = synth.make_system(S, C, K, seed=0)
    sol = Solver(S, C, K, dt)
    sol.set_option("pcg_threads", threads); sol.set_option("pcg_groups", groups); sol.set_option("pcg_mode", mode)
    for k, v in (opts or {}).items():
        sol.set_option(k, v)
    dev = sol.upload_system(sysm)
    lam, dz = sol.new(S * K), sol.new(sol.N)
    try:
        sol.linsys(*dev, 0.0, iters, sysm.rho, lam, dz)
    except Exception as e:
        sol.close(); return None
    torch.cuda.synchronize(); sol.check_status()
    sol.set_option("time_pcg", 1)
    b = [sol.buffer_ptr(i) for i in (3, 4, 5)]
    ms = []
    for i in range(reps + 2):
        sol.pcg(b[0], b[1], b[2], 0.0, iters, lam=lam, check=False)
        v = sol.pcg_last_ms()
        if i >= 2: ms.append(v)
    r = dict(us_per_iter=1e3 * float(np.median(ms)) / iters, groups=sol.get_option("last_groups"),
             threads=sol.get_option("last_threads"), mode=sol.get_option("last_mode"))
    sol.close()
    return r

if __name__ == "__main__":
    cases = [(14, 7, 50, np.float32), (14, 7, 50, np.float64), (14, 7, 512, np.float32), (14, 7, 4096, np.float32),
             (14, 7, 4096, np.float64), (32, 16, 1024, np.float32)]
    for (S, C, K, dt) in cases:
        for t in (0, 64, 128, 256, 384, 512, 768):
            r = run(S, C, K, dt, threads=t)
            if r: print(S, C, K, np.dtype(dt).name, "req_threads", t, json.dumps(r), flush=True)
        r = run(S, C, K, dt, mode=2)
        print(S, C, K, np.dtype(dt).name, "streaming", json.dumps(r), flush=True)
    print("--- single-reduction variant (opt-in)")
    def run_v1(S, C, K, dt, threads=0, iters=100, reps=10):
        sysm = synth.make_system(S, C, K, seed=0)
        sol = Solver(S, C, K, dt); sol.set_option("pcg_variant", 1); sol.set_option("pcg_threads", threads)
        dev = sol.upload_system(sysm); lam, dz = sol.new(S * K), sol.new(sol.N)
        sol.linsys(*dev, 0.0, iters, sysm.rho, lam, dz); torch.cuda.synchronize(); sol.check_status()
        sol.set_option("time_pcg", 1); b = [sol.buffer_ptr(i) for i in (3, 4, 5)]; ms = []
        for i in range(reps + 2):
            sol.pcg(b[0], b[1], b[2], 0.0, iters, lam=lam, check=False); v = sol.pcg_last_ms()
            if i >= 2: ms.append(v)
        r = dict(us_per_iter=1e3 * float(np.median(ms)) / iters, groups=sol.get_option("last_groups"), threads=sol.get_option("last_threads"), variant=sol.get_option("last_variant"))
        sol.close(); return r
    for (S, C, K, dt) in cases:
        for t in (0, 256, 512, 768):
            try: print(S, C, K, np.dtype(dt).name, "v1 req_threads", t, json.dumps(run_v1(S, C, K, dt, t)), flush=True)
            except Exception as e: print("fail", S, C, K, t, e)
