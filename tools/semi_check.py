"""Semi-resident PCG (K beyond the register file, one persistent launch) against the streaming kernels: result and time."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gato_python_amd import synth
from gato_python_amd.solver import Solver

def run(S, C, K, dt, semi, iters=20, reps=5):
    sysm = synth.make_system(S, C, K, seed=0)
    sol = Solver(S, C, K, dt)
    sol.set_option("pcg_semi", semi)
    dev = sol.upload_system(sysm)
    lam, dz = sol.new(S * K), sol.new(sol.N)
    sol.linsys(*dev, 0.0, iters, sysm.rho, lam, dz)
    torch.cuda.synchronize(); sol.check_status()
    sol.set_option("time_pcg", 1)
    b = [sol.buffer_ptr(i) for i in (3, 4, 5)]
    ms = []
    for i in range(reps):
        sol.pcg(b[0], b[1], b[2], 0.0, iters, lam=lam, check=False); ms.append(sol.pcg_last_ms())
    torch.cuda.synchronize(); sol.check_status()
    out = dict(mode=sol.get_option("last_mode"), semi=sol.get_option("last_semi"), groups=sol.get_option("last_groups"),
               threads=sol.get_option("last_threads"), us_per_iter=1e3 * float(np.median(ms)) / iters)
    res = lam.cpu().numpy().copy()
    sol.close()
    return out, res

if __name__ == "__main__":
    cases = [(14, 7, 14000, np.float32), (14, 7, 16384, np.float32), (14, 7, 32768, np.float32), (14, 7, 65536, np.float32), (14, 7, 131072, np.float32),
             (14, 7, 16384, np.float64), (14, 7, 65536, np.float64), (32, 16, 8192, np.float32), (32, 16, 20480, np.float32), (32, 16, 32768, np.float32),
             (32, 16, 40960, np.float32), (32, 16, 8192, np.float64)]
    if len(sys.argv) > 1:
        cases = [c for c in cases if str(c[0]) == sys.argv[1]] if sys.argv[1] in ("14", "32") else cases[:int(sys.argv[1])]
    for (S, C, K, dt) in cases:
        b, lb = run(S, C, K, dt, 0)
        line = f"{S}/{C}/{K} {np.dtype(dt).name}: streaming {b['us_per_iter']:.1f}"
        for semi, name in ((1, "semi-resident"), (2, "no resident rows"), (3, "LDS-DMA ring")):
            a, la = run(S, C, K, dt, semi)
            if a["semi"] != semi:
                line += f" | {name}: n/a"
                continue
            err = np.abs(la - lb).max() / np.abs(lb).max()
            line += f" | {name} {a['groups']}x{a['threads']}: {a['us_per_iter']:.1f} us/iter (rel diff {err:.1e})"
        print(line, flush=True)
