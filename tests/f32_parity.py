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


def check_f32(what, gpu, oracle32, truth64, factor=F32_FACTOR, floor=F32_FLOOR, second_order32=None):
    """second_order32: the same fp32 recurrence summed in ANOTHER CPU order (the numpy restatement beside the C one), for
    the few cases that are order-chaotic by construction - a tiny system iterated past convergence, barely preconditioned
    CG past its loss of conjugacy (tools/past_convergence.py, tests/test_oracle.py::test_fp32_point_jacobi_...): there the
    two CPU orders themselves are up to 10x apart in either direction, and the GPU is held to the worse of the two."""
    if not (np.all(np.isfinite(truth64)) and np.all(np.isfinite(oracle32))):
        return
    eg, eo = rel(np.asarray(gpu, np.float64), truth64), rel(np.asarray(oracle32, np.float64), truth64)
    entry = dict(what=what, err_gpu=float(eg), err_oracle=float(eo))
    if second_order32 is not None:
        e2 = rel(np.asarray(second_order32, np.float64), truth64)
        entry["err_oracle_second_order"] = float(e2)
        eo = max(eo, e2)
        entry["err_oracle"] = float(eo)
    LOG.append(entry)
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
