"""One library (GATO_HIP_LIB), BASELINE configs[1] (14/7/50 fp64): oracle check of a whole solve (lambda, dz, iters), us per PCG
iteration (PCG-only launches, HIP events) and us per whole step (assembly + 100 iterations + dz).  Child of tools/ab_libs.py."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gato_python_amd import synth
from gato_python_amd.solver import Solver
from oracle import gato_oracle as o


def main():
    S, C, K = 14, 7, 50
    dt = np.float32 if "f32" in sys.argv[1:] else np.float64
    out = {}
    s = synth.make_system(S, C, K, seed=0)
    sol = Solver(S, C, K, dt)
    for kv in sys.argv[1:]:
        if "=" in kv:
            k, v = kv.split("="); sol.set_option(k, int(v))
    dev = sol.upload_system(s)
    lam, dz = sol.new(S * K), sol.new(sol.N)
    it = sol.new(1, torch.int32)
    # parity: to tolerance, and 12 fixed iterations
    sol.linsys(*dev, 1e-6, 100, s.rho, lam, dz); torch.cuda.synchronize(); sol.check_status()
    lam_o, dz_o, it_o = o.linsys_solve(*s.csr_args(), S, C, K, 1e-6, 100, s.rho, dtype=dt)
    out["rel_lam_tol"] = float(np.abs(lam.cpu().numpy() - lam_o).max() / np.abs(lam_o).max())
    out["abs_dz_tol"] = float(np.abs(dz.cpu().numpy() - dz_o).max())
    sol.linsys(*dev, 0.0, 12, s.rho, lam, dz); torch.cuda.synchronize()
    lam_o, dz_o, _ = o.linsys_solve(*s.csr_args(), S, C, K, 0.0, 12, s.rho, dtype=dt)
    out["rel_lam_12"] = float(np.abs(lam.cpu().numpy() - lam_o).max() / np.abs(lam_o).max())
    b = [sol.buffer_ptr(i) for i in (3, 4, 5)]
    _, itt = sol.pcg(b[0], b[1], b[2], 1e-6, 100, lam=lam, iters=it)
    out["iters"] = int(itt.item()); out["iters_oracle"] = int(it_o)
    # PCG-only launches
    sol.set_option("time_pcg", 1)
    ms = []
    for i in range(42):
        sol.pcg(b[0], b[1], b[2], 0.0, 100, lam=lam, check=False)
        if i >= 2: ms.append(sol.pcg_last_ms())
    out["us_per_iter"] = round(1e3 * float(np.median(ms)) / 100, 4)
    sol.set_option("time_pcg", 0)
    for _ in range(20): sol.linsys(*dev, 0.0, 100, s.rho, lam, dz)
    torch.cuda.synchronize()
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n): sol.linsys(*dev, 0.0, 100, s.rho, lam, dz)
    torch.cuda.synchronize()
    out["us_per_step"] = round((time.perf_counter() - t0) / n * 1e6, 2)
    sol.check_status()
    sol.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
