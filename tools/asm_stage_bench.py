#!/usr/bin/env python3
"""Device time of each assembly stage kernel (gather+inversions, Schur, stair, dz) on batches and large K, with the bytes a
stage must move and the HBM rate that corresponds to (GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                            # noqa: E402
from gato_python_amd import synth                       # noqa: E402
from gato_python_amd.solver import Solver               # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    cases = [(14, 7, 50, np.float64, 512), (14, 7, 50, np.float32, 512), (14, 7, 4096, np.float32, 1), (14, 7, 4096, np.float64, 1),
             (32, 16, 1024, np.float32, 1)]
    if len(sys.argv) > 1:
        cases = cases[:int(sys.argv[1])]
    for S, C, K, dt, B in cases:
        w = np.dtype(dt).itemsize
        base = synth.make_system(S, C, K, seed=0)
        sol = Solver(S, C, K, dt, batch=B)
        if B > 1:
            d = sol.upload_batch([base] * B)
            sol.set_option("batch_nnz_G", len(base.G_val))
            sol.set_option("batch_nnz_C", len(base.C_val))
        else:
            d = sol.upload_system(base)
        sol.set_option("asm_mode", 1)
        n = S + C
        knots = K * B
        lam, dz = sol.new(B * S * K), sol.new(B * sol.N)
        # whole-solve path with stage timers
        sol.set_option("time_stages", 1)
        call = (lambda: sol.linsys_batched(*d, 0.0, 10, base.rho, lam, dz)) if B > 1 else (lambda: sol.linsys(*d, 0.0, 10, base.rho, lam, dz))
        for _ in range(5):
            call()
        st = []
        for _ in range(10):
            call()
            st.append(sol.last_stage_ms())
        asm = 1e3 * np.median([x["assembly"] for x in st])
        tdz = 1e3 * np.median([x["dz"] for x in st])
        # bytes per knot each stage must move at least
        nnz = (len(base.G_val) + len(base.C_val)) / K
        b_gather = nnz * (w + 4) + (2 * (S * S + C * C) + S * S + S * C) * w
        b_schur = (2 * (S * S + C * C) + S * S + S * C + 6 * S * S) * w            # Ginv(k-1,k) + C_dense in, S row + Pinv row out
        b_ss = (5 * S * S) * w
        b_dz = (S * S + C * C + S * S + S * C + 3 * n) * w
        tot = b_gather + b_schur + b_ss + b_dz
        print(f"{S}/{C}/{K} {np.dtype(dt).name} B={B}: assembly {asm:7.1f} us + dz {tdz:6.1f} us = {asm + tdz:7.1f} us for {knots} knots; "
              f"min bytes {tot * knots / 1e6:.0f} MB -> {tot * knots / (asm + tdz) / 1e6:.2f} TB/s", flush=True)
        sol.close()


if __name__ == "__main__":
    main()
