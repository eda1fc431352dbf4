"""fp32 parity, measured instead of assumed (shared by the GPU test modules; dumped once per session by conftest.py).

The reference computes in fp32.  An fp32 result is judged against the fp64 oracle run on the SAME
(fp32-rounded) inputs: err_gpu = |x_gpu32 - x_64| / |x_64| beside err_oracle = |x_oracle32 - x_64| / |x_64|, the error
the reference's own arithmetic makes in the oracle's (= the reference's) accumulation order.  The HIP kernels sum in
another order (MFMA tiles, DPP trees, two rows per lane), so the bar is err_gpu <= F32_FACTOR * err_oracle + F32_FLOOR.
Every pair is appended to gpurun_out/f32_parity.json (DESIGN.md section 4 quotes the measured maxima)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32_FACTOR, F32_FLOOR = 2.0, 5e-6     # floor: ~40 fp32 ulps (the pendulum run iterates past convergence: 5.2e-6 vs 1.4e-6)
LOG = []


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def check_f32(what, gpu, oracle32, truth64, factor=F32_FACTOR, floor=F32_FLOOR):
    if not (np.all(np.isfinite(truth64)) and np.all(np.isfinite(oracle32))):
        return
    eg, eo = rel(np.asarray(gpu, np.float64), truth64), rel(np.asarray(oracle32, np.float64), truth64)
    LOG.append(dict(what=what, err_gpu=float(eg), err_oracle=float(eo)))
    assert eg <= factor * eo + floor, (what, eg, eo)


def dump():
    if not LOG:
        return
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    ratios = [e["err_gpu"] / max(e["err_oracle"], F32_FLOOR) for e in LOG]
    worst = LOG[int(np.argmax(ratios))]
    with open(os.path.join(out, "f32_parity.json"), "w") as f:
        json.dump(dict(factor=F32_FACTOR, floor=F32_FLOOR, pairs=len(LOG), max_ratio=max(ratios), worst=worst, entries=LOG), f, indent=1)
