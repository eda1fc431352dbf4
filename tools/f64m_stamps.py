"""(scratch build: tools/devbuild.sh 14 7 -DGATO_F64M_STAMP, GATO_HIP_LIB=build/ab/...) where ONE iteration of the one-workgroup
fp64 kernel of BASELINE configs[1] spends its cycles, wave by wave: s_memtime at the phase boundaries of the last-but-one iteration."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gato_python_amd import synth
from gato_python_amd.solver import Solver

NAMES = ["top", "P1 done", "sum1: at barrier", "sum1: past barrier", "v known", "r put, at barrier", "past barrier", "P2 done",
         "sum2: at barrier", "sum2: past barrier", "eta' known", "p put, at barrier", "past barrier (next top)"]


def main():
    S, C, K = 14, 7, 50
    s = synth.make_system(S, C, K, seed=0)
    sol = Solver(S, C, K, np.float64)
    sol.set_option("record_eta", 1); sol.set_option("time_pcg", 1)
    for kv in sys.argv[1:]:
        k, v = kv.split("="); sol.set_option(k, int(v))
    dev = sol.upload_system(s)
    lam, dz = sol.new(S * K), sol.new(sol.N)
    for _ in range(3):
        sol.linsys(*dev, 0.0, 100, s.rho, lam, dz)
    torch.cuda.synchronize()
    h = sol.eta_history(1024 + 8 * 16 + 16)[1024:1024 + 8 * 16].reshape(8, 16)[:, :13]
    t0 = h[:, 0].min()
    print(f"launch {1e3 * sol.pcg_last_ms():.1f} us for 100 iterations; cycles since the first wave entered the iteration (s_memtime units)")
    print("wave " + " ".join(f"{n[:12]:>12s}" for n in NAMES))
    for w in range(8):
        print(f"{w:4d} " + " ".join(f"{h[w, i] - t0:12.0f}" for i in range(13)))
    d = h - t0
    print("slowest wave per point: " + " ".join(f"{d[:, i].max():.0f}" for i in range(13)))
    sol.close()


if __name__ == "__main__":
    main()
