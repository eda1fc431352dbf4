"""One-XCD launches (xcd_pack): us per PCG iteration with the working blocks on each of the eight XCDs (option xcd_sel)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tune_pcg import run


def main():
    for (S, C, K, dt) in ((14, 7, 512, np.float32), (14, 7, 1024, np.float32), (14, 7, 512, np.float64)):
        out = []
        for sel in range(8):
            r = run(S, C, K, dt, reps=20, opts={"xcd_sel": sel})
            out.append(round(r["us_per_iter"], 3))
        auto = run(S, C, K, dt, reps=20)
        print(f"{S}/{C}/{K} {np.dtype(dt).name} ({r['groups']} x {r['threads']}): us/iter by XCD {out}; auto (calibrated) {auto['us_per_iter']:.3f}", flush=True)


if __name__ == "__main__":
    main()
