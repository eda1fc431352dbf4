"""Where the planner changes the launch - kernel family, workgroup size, row layout, semi-resident form, streaming - as K grows,
and a whole solve against the oracle on both sides of every such boundary (K = b - 1, b, b + 1): the sizes where a kernel runs with
its last knot slot full, one knot in its last workgroup, the first / last K of a family.  Every K up to 300, then the changes of the
plan signature located by bisection up to KMAX.
      python tools/boundary_sweep.py [S C [f32|f64 [KMAX [option=value ...]]]]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GATO_NO_TUNE", "1")
import torch
from gato_python_amd import synth
from gato_python_amd.solver import Solver
from oracle import c_oracle as co
from oracle import gato_oracle as o

S, C = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (14, 7)
dt = np.float64 if len(sys.argv) > 3 and sys.argv[3] == "f64" else np.float32
KMAX = int(sys.argv[4]) if len(sys.argv) > 4 else 40000
opts = {k: int(v) for k, v in (kv.split("=") for kv in sys.argv[5:])}
variant = opts.get("pcg_variant", 0)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


_blocks = {}


def system(K):
    return synth.make_system(S, C, K, seed=77)


def run(K, mi, tol, check):
    s = system(K)
    sol = Solver(S, C, K, dt)
    for k_, v_ in opts.items():
        sol.set_option(k_, v_)
    d = sol.upload_system(s)
    lam = torch.full((S * K,), float("nan"), dtype=sol.dtype, device="cuda")
    dz = torch.full((sol.N,), float("nan"), dtype=sol.dtype, device="cuda")
    sol.linsys(*d, tol, mi, s.rho, lam=lam, dz=dz)
    torch.cuda.synchronize()
    sol.check_status()
    g = sol.get_option
    sig = (g("last_mode"), g("last_semi"), g("last_threads") if g("last_groups") > 1 else -g("last_threads"), g("last_pair"), g("last_dpp"), g("last_variant"),
           g("last_dz_fused"), g("last_image"))
    out = None
    if check:
        if g("last_variant"):
            Gd, Cd = co.convert(*s.csr_args()[:6], S, C, K, s.rho, dt)
            Sb, Pb, gam, Gi = co.form_schur(Gd, Cd, s.g.astype(dt), s.c.astype(dt), S, C, K)
            Pb = co.form_ss(Sb, Pb, S, K)
            lo, it_o = o.pcg_single_reduction(Sb, Pb, gam, S, K, tol, mi)
            lo = lo.reshape(-1)
            dzo = co.compute_dz(Gi, Cd, s.g.astype(dt), lo, S, C, K)
        else:
            lo, dzo, it_o = co.linsys_solve(*s.csr_args(), S, C, K, tol, mi, s.rho, dtype=dt)
        out = (rel(lam.cpu().numpy(), lo), rel(dz.cpu().numpy(), dzo), g("last_groups"))
    sol.close()
    return sig, out


def signature(K):
    return run(K, 1, 0.0, False)[0]


# 1. boundaries
bounds, prev = [], signature(1)
sigs = {1: prev}
for K in range(2, min(300, KMAX) + 1):
    sg = signature(K)
    sigs[K] = sg
    if sg != prev:
        bounds.append(K)
    prev = sg
K = 300
step = 256
while K < KMAX:
    K2 = min(K + step, KMAX)
    sg2 = signature(K2)
    if sg2 != prev:                           # a change in (K, K2]: bisect to the first K with another signature (one change per step assumed,
        lo_, hi_ = K, K2                      # later ones are found by the following steps)
        while hi_ - lo_ > 1:
            mid = (lo_ + hi_) // 2
            if signature(mid) != prev:
                hi_ = mid
            else:
                lo_ = mid
        bounds.append(hi_)
        prev = signature(hi_)
        K = hi_
    else:
        K = K2
print(f"{S}/{C} {np.dtype(dt).name} {opts}: {len(bounds)} plan boundaries up to K = {KMAX}: {bounds}", flush=True)

# 2. whole solves on both sides of every boundary (fixed 20 iterations: the same iterate in every summation order to rounding)
bad, done = 0, set()
for b in bounds + [KMAX]:
    for K in (b - 1, b, b + 1):
        if K < 1 or K in done:
            continue
        done.add(K)
        sig, (el, ed, groups) = run(K, 20, 0.0, True)
        tiny = K <= 16
        bar = (1e-5 if tiny else 1e-9 if not variant else 1e-7) if dt == np.float64 else (5e-2 if tiny else 2e-3)
        if K == 1:
            ed = 0.0
        ok = el < bar and ed < 10 * bar
        bad += not ok
        print(("ok   " if ok else "FAIL ") + f"K = {K}: {sig} groups {groups} lam {el:.1e} dz {ed:.1e}", flush=True)
print("BOUNDARIES", "FAILED" if bad else "ok", bad)
sys.exit(1 if bad else 0)
