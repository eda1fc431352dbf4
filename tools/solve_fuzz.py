"""Randomised one-GPU solves against the oracle: shape, K (1 .. tens of thousands: every launch variant the planner knows), type,
entry (the reference surface linsys_solve with its cached solver, the device entry, the stage entries, the block entry, a batch),
recurrence, launch options, tolerance, iteration cap, rho, dense / diagonal Q.  fp64 cases must match the oracle to rounding with
the same iteration count (default recurrence); fp32 cases are measured like the parity suite (tests/f32_parity.py).  Solvers are
created and destroyed case after case in one process.  Prints the failing cases, if any.
      python tools/solve_fuzz.py [cases] [seed]          FUZZ_ONLY=i,j,...: draw every case, run only these"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GATO_VERBOSE", "0")
import torch
from gato_python_amd import _lib, synth, linsys as host
from gato_python_amd.solver import Solver
from oracle import c_oracle as co
from oracle import gato_oracle as o
from f32_parity import F32_FACTOR, F32_FLOOR


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


SHAPES = [(14, 7), (14, 7), (14, 7), (2, 1), (12, 6), (32, 16), (4, 2), (6, 3)]
OPTION_DRAWS = dict(pcg_threads=[0, 0, 0, 64, 128, 256, 512, 1024], pcg_semi=[-1, -1, -1, 0, 1, 2, 3], dpp_rows=[-1, -1, 0, 1],
                    xcd_pack=[-1, -1, 0, 1], max_workgroups=[0, 0, 0, 1, 7, 33, 100, 256], no_pair=[0, 0, 1], no_image=[0, 0, 1],
                    no_single_lds=[0, 0, 1], wave_pub=[1, 1, 0], coop_launch=[0, 0, 1], pcg_mode=[0, 0, 0, 1, 2],
                    no_fuse_dz=[0, 0, 1], asm_mode=[0, 0, 1, 2])


def compiled(S, C):
    n = _lib.lib().gato_num_shapes()
    import ctypes as ct
    for i in range(n):
        a, b = ct.c_int(), ct.c_int()
        _lib.lib().gato_shape(i, ct.byref(a), ct.byref(b))
        if (a.value, b.value) == (S, C):
            return True
    return False


def draw(rng, i):
    S, C = SHAPES[int(rng.integers(0, len(SHAPES)))]
    band = int(rng.integers(0, 10))
    kcap = {2: 60000, 4: 40000, 6: 30000, 12: 16000, 14: 24000, 32: 5000}[S]
    K = int(rng.integers(1, 12)) if band == 0 else int(rng.integers(12, 200)) if band < 4 else int(rng.integers(200, 3000)) if band < 8 \
        else int(rng.integers(3000, kcap))
    dt = np.float64 if rng.integers(0, 3) else np.float32
    entry = ["host", "device", "stages", "blocks", "batch"][int(rng.integers(0, 5))]
    variant = int(rng.integers(0, 4) == 0)
    opts = {}
    if entry != "host":
        for name, vals in OPTION_DRAWS.items():
            if rng.integers(0, 4) == 0:
                opts[name] = int(rng.choice(vals))
        if variant:
            opts["pcg_variant"] = 1
    tol = float(rng.choice([1e-6, 1e-9, 1e-12])) if dt == np.float64 else float(rng.choice([1e-4, 1e-6]))
    mi = int(rng.choice([200, 200, 200, 1, 3, 17]))
    rho = float(rng.choice([1e-3, 1e-3, 0.0, 0.1]))
    dense_q = bool(rng.integers(0, 2))
    B = int(rng.integers(2, 9)) if entry == "batch" else 1
    if entry == "batch":
        K = min(K, 600)
    # extensions beside the reference's behaviour: a true warm start from a random lambda0 (N2), the reference's other two
    # preconditioner builds (block-Jacobi, point-Jacobi) - device and stage entries, default recurrence
    extra = int(rng.integers(0, 6))
    warm = entry in ("device", "stages") and extra == 0
    precon = int(rng.integers(1, 3)) if entry == "device" and extra == 1 else 0
    if warm or precon:
        opts.pop("pcg_variant", None)
        K = min(K, 6000 if warm else 1500)                # the numpy restatement checks these
    return dict(i=i, S=S, C=C, K=K, dt=dt, entry=entry, opts=opts, tol=tol, mi=mi, rho=rho, dense_q=dense_q, B=B, warm=warm, precon=precon)


def oracle_solve(s, p, variant):
    """(lam, dz, iters) of the restatement in p['dt'], and of fp64 on the same (type-rounded) inputs."""
    S, C, K, dt = p["S"], p["C"], p["K"], p["dt"]
    out = []
    for t in ([dt] if dt == np.float64 else [dt, np.float64]):
        a = [np.asarray(x, dt).astype(t) if x.dtype.kind == "f" else x for x in s.csr_args()]
        if p["warm"]:
            Gd, Cd = co.convert(*a[:6], S, C, K, dt(p["rho"]).astype(t), t)
            Sb, Pb, gam, Gi = co.form_schur(Gd, Cd, a[6], a[7], S, C, K)
            Pb = co.form_ss(Sb, Pb, S, K)
            lam, it = o.pcg(Sb, Pb, gam, S, K, p["tol"], p["mi"], lam0=np.asarray(p["lam0"], dt).astype(t))
            out.append((lam.reshape(-1), co.compute_dz(Gi, Cd, a[6], lam.reshape(-1), S, C, K), it))
        elif p["precon"]:
            out.append(o.linsys_solve(*a, S, C, K, p["tol"], p["mi"], dt(p["rho"]).astype(t), dtype=t, precon_mode=p["precon"])[:3])
        elif not variant:
            out.append(co.linsys_solve(*a, S, C, K, p["tol"], p["mi"], dt(p["rho"]).astype(t), dtype=t))
        else:
            Gd, Cd = co.convert(*a[:6], S, C, K, dt(p["rho"]).astype(t), t)
            Sb, Pb, gam, Gi = co.form_schur(Gd, Cd, a[6], a[7], S, C, K)
            Pb = co.form_ss(Sb, Pb, S, K)
            lam, it = o.pcg_single_reduction(Sb, Pb, gam, S, K, p["tol"], p["mi"])
            out.append((lam, co.compute_dz(Gi, Cd, a[6], lam, S, C, K), it))
    return out


def run_gpu(s, p, systems):
    """-> (lam, dz, iters or None, what ran)."""
    S, C, K, dt, B = p["S"], p["C"], p["K"], p["dt"], p["B"]
    if p["entry"] == "host":
        host.set_precision("f64" if dt == np.float64 else "f32")
        lam, dz = host.linsys_solve(s.G_row, s.G_col, s.G_val, s.C_row, s.C_col, s.C_val, s.g, s.c, np.zeros(S * K), 1, p["tol"],
                                    p["mi"], False, p["rho"])
        st = host.last_stats()
        return np.asarray(lam), np.asarray(dz), st["iters"], "host"
    sol = Solver(S, C, K, dt, batch=B)
    try:
        for k, v in p["opts"].items():
            sol.set_option(k, v)
        if p["warm"]:
            sol.set_option("true_warm_start", 1)
        if p["precon"]:
            sol.set_option("precon_mode", p["precon"])
        new = lambda n: torch.full((n,), float("nan"), dtype=torch.float64 if dt == np.float64 else torch.float32, device="cuda")
        its = None
        if p["entry"] == "device":
            d = sol.upload_system(s)
            lam, dz = (sol.to_device(p["lam0"]) if p["warm"] else new(S * K)), new(sol.N)
            sol.linsys(*d, p["tol"], p["mi"], p["rho"], lam=lam, dz=dz)
        elif p["entry"] == "blocks":
            Gd, Cd = co.convert(*s.csr_args()[:6], S, C, K, 0.0, dt)                 # the block layouts, rho not yet added
            lam, dz = new(S * K), new(sol.N)
            sol.linsys_blocks(sol.to_device(Gd), sol.to_device(Cd) if Cd.size else sol.new(1), sol.to_device(s.g), sol.to_device(s.c),
                              p["tol"], p["mi"], p["rho"], lam=lam, dz=dz)
        elif p["entry"] == "stages":
            d = sol.upload_system(s)
            Gd, Cd = sol.convert(*d[:6], p["rho"])
            Sb, Pb, gam, Gi = sol.form_schur(Gd, Cd, d[6], d[7])
            Pb = sol.form_ss(Sb, Pb)
            lam, its = sol.pcg(Sb, Pb, gam, p["tol"], p["mi"], lam=sol.to_device(p["lam0"]) if p["warm"] else None)
            dz = sol.compute_dz(Gi, Cd, d[6], lam)
        else:
            d = sol.upload_batch(systems)
            lam, dz = new(S * K * B), new(sol.N * B)
            its = torch.zeros(B, dtype=torch.int32, device="cuda")
            sol.linsys_batched(*d, p["tol"], p["mi"], p["rho"], lam, dz, its)
        torch.cuda.synchronize()
        sol.check_status()
        what = f"mode {sol.get_option('last_mode')} groups {sol.get_option('last_groups')} threads {sol.get_option('last_threads')} " \
               f"semi {sol.get_option('last_semi')} variant {sol.get_option('last_variant')} fallback {sol.get_option('last_fallback')}"
        ran_variant = sol.get_option("last_variant")
        return lam.cpu().numpy(), dz.cpu().numpy(), (its.cpu().numpy() if its is not None else None), what, ran_variant
    finally:
        sol.close()


def case(rng, i, only=None):
    p = draw(rng, i)
    tag = f"case {i}: {p['S']}/{p['C']}/{p['K']} {np.dtype(p['dt']).name} {p['entry']}" + (f" x{p['B']}" if p["B"] > 1 else "") + \
          f" tol {p['tol']:g} max_iters {p['mi']} rho {p['rho']:g} dense_q {int(p['dense_q'])} {p['opts']}" + \
          (" warm start" if p["warm"] else "") + (f" precon_mode {p['precon']}" if p["precon"] else "")
    if only is not None and i not in only:
        return tag + " not run", True
    if not compiled(p["S"], p["C"]):
        return tag + " not run", True
    S, C, K, B = p["S"], p["C"], p["K"], p["B"]
    systems = [synth.make_system(S, C, K, seed=5000 + 16 * i + b, dense_q=p["dense_q"], rho=p["rho"]) for b in range(B)]
    s = systems[0]
    if p["warm"]:
        p["lam0"] = np.random.default_rng(9000 + i).standard_normal(S * K).astype(p["dt"])
    try:
        got = run_gpu(s, p, systems)
    except Exception as e:                                  # an option combination the library refuses: reported, not a failure
        msg = str(e)
        refused = "EINVAL" in msg or "ESHAPE" in msg or "do not fit" in msg or "not available" in msg or "needs" in msg
        return tag + (" refused: " if refused else " RAISED: ") + msg[:160], refused
    lam, dz, its, what = got[:4]
    variant = got[4] if len(got) > 4 else 0
    ok, notes = True, []
    n, sk = (S + C) * K - C, S * K
    for b in range(B):
        ref = oracle_solve(systems[b], p, variant)
        lam_b, dz_b = lam[b * sk:(b + 1) * sk], dz[b * n:(b + 1) * n]
        it_b = None if its is None else int(np.asarray(its).reshape(-1)[b])
        if not (np.isfinite(lam_b).all() and np.isfinite(dz_b).all()):
            ok = False
            notes.append(f"sys {b}: non-finite output")
            continue
        if p["dt"] == np.float64:
            lam_o, dz_o, it_o = ref[0]
            # fixed-iteration runs and converged runs alike: the same iterate to rounding (the launch variants sum in other orders;
            # near the tolerance one more / fewer iteration can happen only if eta sits within rounding of it - not seen so far).
            # Systems of a few knots run CG to its finite termination, where the iterate depends on the summation order at the
            # level of cond(S) * eps: there the bar is what the two CPU restatements (C and numpy) differ by between themselves.
            bar_l = bar_d = 1e-6 if variant else 1e-8
            if K <= 16 and not variant:
                lam_n, dz_n = o.linsys_solve(*systems[b].csr_args(), S, C, K, p["tol"], p["mi"], p["rho"], dtype=np.float64)[:2]
                bar_l, bar_d = max(bar_l, 16 * rel(lam_n, lam_o)), max(bar_d, 16 * rel(dz_n, dz_o))
            elif K <= 16:
                bar_l, bar_d = 1e-4, 1e-3
            if p["precon"] == 2:         # point-Jacobi: barely preconditioned CG loses conjugacy, hundreds of iterations amplify the
                bar_l, bar_d = max(bar_l, 1e-5), max(bar_d, 1e-4)     # summation order (tests/test_oracle.py::test_fp32_point_jacobi_...)
            el, ed = rel(lam_b, lam_o), rel(dz_b, dz_o)
            if K == 1:
                ed = 0.0                 # dz of a one-knot system is a difference of equal numbers: zero to rounding, no relative error
            same_it = it_b is None or variant or it_b == it_o or (K <= 16 and abs(it_b - it_o) <= 1)
            if not (el < bar_l and ed < bar_d and same_it):
                ok = False
            notes.append(f"sys {b}: lam {el:.1e} dz {ed:.1e} iters {it_b} (oracle {it_o})")
        else:
            (lam_o, dz_o, it_o), (lam_t, dz_t, it_t) = ref
            f = 16.0 if p["precon"] == 2 else F32_FACTOR if K > 16 else 8.0     # a few knots, point-Jacobi: order-chaotic (see the fp64 branch)
            bar_l = f * rel(lam_o, lam_t) + F32_FLOOR
            bar_d = f * rel(dz_o, dz_t) + F32_FLOOR
            el, ed = rel(lam_b, lam_t), rel(dz_b, dz_t)
            if K == 1:
                ed = bar_d = 0.0
            if p["mi"] >= 200 and not variant and it_t == it_o:           # measured only where fp32 and fp64 iterate alike (converged)
                if not (el <= bar_l and ed <= bar_d):
                    ok = False
            elif not (el < 0.5 and ed < 0.5):                             # fixed iterations in fp32: another order, another iterate
                ok = False
            notes.append(f"sys {b}: lam {el:.1e} (oracle32 {rel(lam_o, lam_t):.1e}) dz {ed:.1e} (oracle32 {rel(dz_o, dz_t):.1e}) iters {it_b} "
                         f"(oracle {it_o}, fp64 {it_t})")
    return tag + " | " + what + " | " + "; ".join(notes if not ok else notes[:3]), ok


if __name__ == "__main__":
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    only = {int(v) for v in os.environ["FUZZ_ONLY"].split(",")} if os.environ.get("FUZZ_ONLY") else None
    bad = 0
    for i in range(ncases):
        msg, ok = case(rng, i, only)
        if msg.endswith("not run"):
            continue
        print(("ok   " if ok else "FAIL ") + msg, flush=True)
        bad += 0 if ok else 1
    print("FUZZ", "FAILED" if bad else "ok", bad)
    sys.exit(1 if bad else 0)
