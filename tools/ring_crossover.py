"""where the LDS-DMA ring (pcg_semi = 3) overtakes the semi-resident launch (1) and the launch without
resident rows (2), per dtype and STATE_SIZE - the numbers behind the auto rule in plan_resident_k (gato_capi.hip).
usage: ring_crossover.py [f64|f32|s32|s32f64]    (us per PCG iteration, 20 iterations, median of 5)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from semi_check import run

SWEEPS = {
    "f64": [(14, 7, K, np.float64) for K in (12288, 16384, 20480, 24576, 32768, 40960, 49152, 65536)],
    "f32": [(14, 7, K, np.float32) for K in (49152, 65536, 81920, 98304, 131072)],
    "s32": [(32, 16, K, np.float32) for K in (16384, 24576, 32768)],
    "s32f64": [(32, 16, K, np.float64) for K in (4096, 8192, 16384)],
}

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "f64"
    for (S, C, K, dt) in SWEEPS[which]:
        mb = 2 * 3 * S * S * K * np.dtype(dt).itemsize / 1e6
        line = f"{S}/{C}/{K} {np.dtype(dt).name} ({mb:.0f} MB of S + Pinv):"
        base = None
        for semi, name in ((1, "semi"), (2, "nores"), (3, "ring"), (-1, "auto")):
            try:
                a, la = run(S, C, K, dt, semi)
            except Exception as e:      # noqa: BLE001
                line += f" | {name}: n/a ({str(e)[:40]})"
                continue
            if semi > 0 and a["semi"] != semi:
                line += f" | {name}: n/a"
                continue
            if base is None:
                base = la
            tag = f"{name}" + (f"->{a['semi']}" if semi < 0 else "")
            line += f" | {tag} {a['groups']}x{a['threads']}: {a['us_per_iter']:.1f} (diff {np.abs(la - base).max() / np.abs(base).max():.0e})"
        print(line, flush=True)
