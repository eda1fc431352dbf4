"""GPU suite (-m gpu): the HIP path, called through the C ABI, against the oracle on the same seeded
inputs, against the committed golden fixtures, and - at BASELINE's full sizes - through
size-independent properties (KKT residuals, determinism, resident == streaming).

Tolerances: CSR scatter is bit-exact.  fp64 stages: 1e-11 relative (same formulas, different
summation order only in the wave-parallel dots).  fp32: measured against the fp64 oracle on the same
inputs, beside the fp32 oracle's own error (check_f32 below: GPU error <= 2 x oracle error), PCG at a
fixed iteration count and at the exit test.  BASELINE's bar, ||dz - dz_ref||inf < 1e-6 and lambda within 1e-6 relative, is
asserted in fp64 against the dense KKT solve.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from gato_python_amd import _lib, synth          # noqa: E402
from oracle import c_oracle as co                # noqa: E402
from oracle import gato_oracle as o              # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU suite needs a GPU"
    _lib.lib()                                    # the HIP library must be the thing that runs


from f32_parity import check_f32                # noqa: E402  (tests/f32_parity.py: the measured fp32 bar and its log)


def make_solver(S, C, K, dt):
    from gato_python_amd.solver import Solver
    return Solver(S, C, K, dt)


def _f32_truth_inputs(s):
    """The fp32-rounded inputs of a system as fp64 arrays + the fp32-rounded rho: what an fp32 run really solves."""
    return s.astype(np.float32).astype(np.float64), float(np.float32(s.rho))


def check_solve(tag, s, S, C, K, dt, tol, mi, lam, dz, it=None, it_slack=2, rerun=None, f64_tol=1e-8, two_orders=False):
    """A whole solve (CSR in, lambda / dz out) against the C oracle's whole solve on the same inputs.
    fp64: the oracle's iteration count, lambda and dz to f64_tol.
    fp32: iteration count within it_slack of the fp32 oracle's; lambda and dz judged by check_f32 against the CONVERGED
    fp64 solution of the fp32-rounded system, beside the fp32 oracle's own error (both were stopped by the same test).
    If the exit iterations differ and `rerun(tol, mi) -> (lam, dz)` is given, the comparison is repeated at the FIXED
    iteration count both reached (exit_tol = 0), against the fp64 iterates after that many iterations.
    Returns the oracle's iteration count."""
    lam, dz = np.asarray(lam), np.asarray(dz)
    lam_o, dz_o, it_o = co.linsys_solve(*s.csr_args(), S, C, K, tol, mi, s.rho, dtype=dt)
    if np.dtype(dt) == np.float64:
        if it is not None:
            assert it == it_o, (tag, it, it_o)
        assert rel(lam, lam_o) < f64_tol and rel(dz, dz_o) < f64_tol, (tag, rel(lam, lam_o), rel(dz, dz_o))
        return it_o
    if it is not None:
        assert abs(it - it_o) <= it_slack, (tag, it, it_o)
    s64, rho32 = _f32_truth_inputs(s)
    lam_t, dz_t, _ = co.linsys_solve(*s64.csr_args(), S, C, K, 1e-14, max(600, 2 * mi), rho32, dtype=np.float64)
    lam_n = dz_n = None
    if two_orders:      # order-chaotic by construction (see f32_parity.check_f32): the numpy restatement as the second CPU order
        lam_n, dz_n = o.linsys_solve(*s.csr_args(), S, C, K, tol, mi, s.rho, dtype=np.float32)[:2]
    check_f32(f"solve lambda {tag}", lam, lam_o, lam_t, second_order32=lam_n)
    check_f32(f"solve dz {tag}", dz, dz_o, dz_t, second_order32=dz_n)
    if it is not None and it != it_o and rerun is not None:
        n = min(it, it_o) + 1                                  # iterations both runs completed (iters = 0-based exit index)
        n = min(n, mi)
        lam_f, dz_f = rerun(0.0, n)
        lam_of, dz_of, _ = co.linsys_solve(*s.csr_args(), S, C, K, 0.0, n, s.rho, dtype=np.float32)
        lam_tf, dz_tf, _ = co.linsys_solve(*s64.csr_args(), S, C, K, 0.0, n, rho32, dtype=np.float64)
        check_f32(f"solve lambda after {n} fixed iterations {tag}", lam_f, lam_of, lam_tf)
        check_f32(f"solve dz after {n} fixed iterations {tag}", dz_f, dz_of, dz_tf)
    return it_o


def check_pcg(tag, Sb, Pb, gam, S, K, tol, mi, lam, it=None, it_slack=2, f64_tol=1e-9):
    """A PCG run on given (oracle-assembled) matrices against the C oracle's PCG on the same matrices: fp64 to f64_tol with
    the oracle's iteration count; fp32 by check_f32 against the fp64 iteration on the same fp32 matrices - at the same
    fixed count when exit_tol = 0, else against the converged fp64 solution."""
    lam = np.asarray(lam)
    lam_o, it_o = co.pcg(Sb, Pb, gam, S, K, tol, mi)
    if Sb.dtype == np.float64:
        if it is not None:
            assert it == it_o, (tag, it, it_o)
        assert rel(lam, lam_o) < f64_tol, (tag, rel(lam, lam_o))
        return it_o
    if it is not None:
        assert abs(it - it_o) <= it_slack, (tag, it, it_o)
    S64, P64, g64 = Sb.astype(np.float64), Pb.astype(np.float64), gam.astype(np.float64)
    truth = co.pcg(S64, P64, g64, S, K, 0.0, mi)[0] if tol == 0.0 else co.pcg(S64, P64, g64, S, K, 1e-14, max(600, 2 * mi))[0]
    check_f32(f"pcg {tag}", lam, lam_o, truth)
    return it_o


def system(S, C, K, seed=0, dq=False):
    if (S, C, K) == (2, 1, 5) and seed == -1:
        return synth.pendulum_system()
    return synth.make_system(S, C, K, seed=seed, dense_q=dq)


CASES = [(2, 1, 5, -1, False), (2, 1, 2, 0, False), (2, 1, 1, 0, False), (14, 7, 50, 0, False), (14, 7, 7, 1, True),
         (32, 16, 12, 5, True), (2, 1, 700, 2, True)]


@pytest.mark.parametrize("S,C,K,seed,dq", CASES)
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_every_stage_against_oracle(S, C, K, seed, dq, dt):
    s = system(S, C, K, seed, dq) if K > 1 else synth.blocks_to_csr(*synth.make_blocks(S, C, 1, seed, dq))
    sol = make_solver(S, C, K, dt)
    dev = sol.upload_system(s)
    f64 = dt == np.float64
    tol_blk = 1e-11 if f64 else 3e-4
    # A1 convert: bit-exact
    Gd, Cd = sol.convert(*dev[:6], s.rho)
    Gd_o, Cd_o = co.convert(*s.csr_args()[:6], S, C, K, s.rho, dt)
    assert np.array_equal(host(Gd), Gd_o) and np.array_equal(host(Cd), Cd_o)
    # A2 Schur / block-Jacobi / gamma / inverses
    Sb, Pb, gam, Gi = sol.form_schur(Gd, Cd, dev[6], dev[7])
    Sb_o, Pb_o, gam_o, Gi_o = co.form_schur(Gd_o, Cd_o, s.g, s.c, S, C, K)
    tag = f"{S}/{C}/{K} seed {seed}"
    if f64:
        assert rel(host(Gi), Gi_o) < tol_blk and rel(host(Sb), Sb_o) < tol_blk
        assert rel(host(Pb), Pb_o) < tol_blk and rel(host(gam), gam_o) < tol_blk
    else:                                     # fp32: against the fp64 oracle on the same fp32-rounded inputs
        g32, c32 = s.g.astype(np.float32).astype(np.float64), s.c.astype(np.float32).astype(np.float64)
        Sb_t, Pb_t, gam_t, Gi_t = co.form_schur(Gd_o.astype(np.float64), Cd_o.astype(np.float64), g32, c32, S, C, K)
        for nm, a, b, t in (("Ginv", Gi, Gi_o, Gi_t), ("S", Sb, Sb_o, Sb_t), ("Pinv.main", Pb, Pb_o, Pb_t), ("gamma", gam, gam_o, gam_t)):
            check_f32(f"schur {nm} {tag}", host(a), b, t)
        assert rel(host(Sb), Sb_o) < tol_blk                      # and still close to the fp32 oracle itself
    # A3 stair off-diagonals (fed with the oracle's inputs so the stage is isolated)
    Pb2 = sol.form_ss(sol.to_device(Sb_o), sol.to_device(Pb_o))
    Pb2_o = co.form_ss(Sb_o, Pb_o, S, K)
    if f64:
        assert rel(host(Pb2), Pb2_o) < 1e-12
    else:
        check_f32(f"stair Pinv {tag}", host(Pb2), Pb2_o, co.form_ss(Sb_o.astype(np.float64), Pb_o.astype(np.float64), S, K))
    # A5 PCG on the oracle's S, Pinv, gamma: fixed iteration count
    n_it = 8
    lam, it = sol.pcg(sol.to_device(Sb_o), sol.to_device(Pb2_o), sol.to_device(gam_o), 0.0, n_it)
    lam_o, it_o = co.pcg(Sb_o, Pb2_o, gam_o, S, K, 0.0, n_it)
    assert int(host(it)[0]) == it_o == n_it
    if np.all(np.isfinite(lam_o)):
        if f64:
            assert rel(host(lam), lam_o) < 1e-10
        else:
            lam_t, _ = co.pcg(Sb_o.astype(np.float64), Pb2_o.astype(np.float64), gam_o.astype(np.float64), S, K, 0.0, n_it)
            check_f32(f"pcg {n_it} iterations {tag}", host(lam), lam_o, lam_t)
    # A9 dz on the oracle's lambda
    lam_c, _ = co.pcg(Sb_o, Pb2_o, gam_o, S, K, 1e-8, 200)
    dz = sol.compute_dz(sol.to_device(Gi_o), sol.to_device(Cd_o) if K > 1 else sol.new(1), dev[6], sol.to_device(lam_c))
    dz_o = co.compute_dz(Gi_o, Cd_o, s.g, lam_c, S, C, K)
    if f64:
        assert rel(host(dz), dz_o) < 1e-12
    else:
        dz_t = co.compute_dz(Gi_o.astype(np.float64), Cd_o.astype(np.float64), s.g.astype(np.float32).astype(np.float64),
                             lam_c.astype(np.float64), S, C, K)
        check_f32(f"dz {tag}", host(dz), dz_o, dz_t)
    sol.close()


def test_pendulum_golden_through_the_dropin(golden_dir):
    """The reference's own test (test_pendulum_5.py) run against this build, with its assertion."""
    import gpu_library
    gold = json.load(open(os.path.join(golden_dir, "pendulum.json")))
    i, e = gold["inputs"], gold["expected"]
    gpu_library.clear_problem_size()
    args = (i["G_row"], i["G_col"], i["G_val"], i["C_row"], i["C_col"], i["C_val"], i["g_val"], i["c_val"],
            i["input_lambda"], i["testiters"], i["exit_tol"], i["max_iters"], i["warm_start"], i["rho"])
    l, dz = gpu_library.linsys_solve(*args)
    assert isinstance(l, list) and isinstance(dz, list) and len(l) == 10 and len(dz) == 14
    st = gpu_library.last_stats()
    # fp32: the exit test |eta| < 1e-6 falls on iteration 4 or 5 depending on rounding (numpy restatement 5,
    # C restatement 4; in fp64 it is 4)
    assert st["iters"] in (e["iters_f64"], e["iters_f32"]) and (st["S"], st["C"], st["K"]) == (2, 1, 5) and len(st["ms"]) == 10
    x = np.concatenate([e["dense_kkt_norho_dz"], e["dense_kkt_norho_lam"]])
    assert np.allclose(np.concatenate([dz, l]), x, rtol=1, atol=0.01)          # test_pendulum_5.py:37
    assert rel(l, e["lam"]) < 5e-5 and np.abs(np.asarray(dz) - e["dz"]).max() < 5e-3
    gpu_library.set_precision("f64")
    try:
        l, dz = gpu_library.linsys_solve(*args)
        assert gpu_library.last_stats()["iters"] == e["iters_f64"]
        assert rel(l, e["lam"]) < 1e-11 and np.abs(np.asarray(dz) - e["dz"]).max() < 1e-9
        assert rel(l, e["dense_kkt_lam"]) < 1e-9
    finally:
        gpu_library.set_precision("f32")


@pytest.mark.parametrize("name,S,C,K,seed,dq,tol,mi", [("iiwa_14_7_50_seed0.npz", 14, 7, 50, 0, False, 1e-6, 100),
                                                       ("s32_c16_k12_seed5_denseq.npz", 32, 16, 12, 5, True, 1e-12, 500)])
def test_golden_fp64_whole_solve(golden_dir, name, S, C, K, seed, dq, tol, mi):
    gold = np.load(os.path.join(golden_dir, name))
    s = synth.make_system(S, C, K, seed=seed, dense_q=dq)
    sol = make_solver(S, C, K, np.float64)
    dev = sol.upload_system(s)
    lam, dz = sol.new(S * K), sol.new(sol.N)
    sol.linsys(*dev, tol, mi, s.rho, lam, dz)
    torch.cuda.synchronize()
    sol.check_status()
    assert rel(host(lam), gold["lam"]) < 1e-9 and rel(host(dz), gold["dz"]) < 1e-9
    if "lam_tight" in gold:                     # BASELINE bar against the dense KKT solve
        sol.linsys(*dev, 1e-15, 1000, s.rho, lam, dz)
        torch.cuda.synchronize()
        assert rel(host(lam), gold["dense_lam"]) < 1e-6
        assert np.abs(host(dz) - gold["dense_dz"]).max() < 1e-6
    sol.close()


@pytest.mark.parametrize("S,C,K,dt,opts", [
    (14, 7, 50, np.float32, {}),                                  # one workgroup, two rows per lane (packed FMA)
    (14, 7, 50, np.float32, dict(no_pair=1)),                     # one workgroup, one row per lane
    (14, 7, 73, np.float32, {}),                                  # largest two-rows-per-lane system
    (2, 1, 400, np.float32, {}),
    (14, 7, 50, np.float64, {}),                                  # one workgroup, two rows per lane in 3 of 8 waves (mixed kernel)
    (14, 7, 50, np.float64, dict(no_pair=1)),                     # one workgroup, one row per lane, Pinv row tails in LDS (what batches run)
    (14, 7, 50, np.float64, dict(no_single_lds=1)),               # two workgroups (register budget)
    (14, 7, 37, np.float64, {}),                                  # mixed kernel with idle lanes: the last two-row lane's knot continues in one-row lanes
    (14, 7, 41, np.float64, {}),
    (14, 7, 37, np.float64, dict(no_pair=1)),                     # LDS-tail variant with idle lanes
    (14, 7, 50, np.float32, dict(pcg_groups=7)),                  # forced ragged split
    (14, 7, 50, np.float64, dict(pcg_threads=64)),                # 4 knots per workgroup, 13 groups
    (14, 7, 512, np.float32, {}),
    (14, 7, 512, np.float64, {}),
    (14, 7, 4096, np.float32, {}),
    (14, 7, 4096, np.float64, {}),
    (32, 16, 1024, np.float32, {}),
    (32, 16, 256, np.float64, {}),
    (12, 6, 700, np.float32, {}),
    (6, 3, 900, np.float64, {}),
    (14, 7, 777, np.float64, dict(pcg_threads=128)),              # ragged last workgroup, 87 groups across the XCDs
    (2, 1, 3000, np.float32, dict(pcg_threads=128)),
    # row layout of the plain launches (option dpp_rows): the one the default does not take for the shape, and one-workgroup launches
    (14, 7, 512, np.float32, dict(dpp_rows=1)),                   # DPP rows in fp32 at S = 14 (auto keeps the packed-FMA LDS form)
    (14, 7, 4096, np.float32, dict(dpp_rows=1)),                  # 128 workgroups across the XCDs
    (14, 7, 512, np.float64, dict(dpp_rows=0)),                   # LDS operand windows in fp64 (auto takes DPP rows)
    (14, 7, 4096, np.float64, dict(dpp_rows=0)),
    (32, 16, 1024, np.float32, dict(dpp_rows=1)),
    (32, 16, 256, np.float64, dict(dpp_rows=0)),
    (12, 6, 700, np.float32, dict(dpp_rows=1)),
    (12, 6, 700, np.float64, {}),
    (14, 7, 30, np.float64, {}),                                  # one workgroup in the DPP-row layout
    (14, 7, 30, np.float64, dict(dpp_rows=0)),
    (14, 7, 50, np.float64, dict(dpp_rows=1)),                    # forced: two DPP-row workgroups instead of the mixed-rows kernel
    (14, 7, 45, np.float32, dict(dpp_rows=1, pcg_threads=768)),   # forced: one DPP-row workgroup of 12 waves instead of two rows per lane
    (14, 7, 777, np.float64, dict(pcg_threads=128, dpp_rows=0)),
    (14, 7, 512, np.float32, dict(pcg_mode=_lib.PCG_STREAMING)),
    (14, 7, 777, np.float64, dict(pcg_mode=_lib.PCG_STREAMING)),
    (32, 16, 100, np.float32, dict(pcg_mode=_lib.PCG_STREAMING)),
    (2, 1, 5000, np.float64, dict(pcg_mode=_lib.PCG_STREAMING)),
])
def test_pcg_variants_against_oracle(S, C, K, dt, opts):
    """Resident (1..W workgroups, in-launch hand-offs) and streaming PCG vs the C oracle on the same
    S, Pinv, gamma: same iteration count at exit_tol, lambda equal to rounding."""
    s = synth.make_system(S, C, K, seed=11)
    Gd, Cd = co.convert(*s.csr_args()[:6], S, C, K, s.rho, dt)
    Sb, Pb, gam, Gi = co.form_schur(Gd, Cd, s.g, s.c, S, C, K)
    Pb = co.form_ss(Sb, Pb, S, K)
    sol = make_solver(S, C, K, dt)
    for k, v in opts.items():
        sol.set_option(k, v)
    f64 = dt == np.float64
    dS, dP, dg = sol.to_device(Sb), sol.to_device(Pb), sol.to_device(gam)
    # fixed number of iterations
    lam, it = sol.pcg(dS, dP, dg, 0.0, 12)
    lam_o, it_o = co.pcg(Sb, Pb, gam, S, K, 0.0, 12)
    assert int(host(it)[0]) == 12
    tag = f"{S}/{C}/{K} {opts}"
    S64, P64, g64 = Sb.astype(np.float64), Pb.astype(np.float64), gam.astype(np.float64)
    if f64:
        assert rel(host(lam), lam_o) < 1e-10
    else:                                      # fp32: error against the fp64 iteration on the same matrices
        check_f32(f"pcg 12 iterations {tag}", host(lam), lam_o, co.pcg(S64, P64, g64, S, K, 0.0, 12)[0])
    # run to tolerance: iteration count of the reference's exit test
    tol = 1e-8 if f64 else 1e-4
    lam, it = sol.pcg(dS, dP, dg, tol, 300)
    lam_o, it_o = co.pcg(Sb, Pb, gam, S, K, tol, 300)
    assert abs(int(host(it)[0]) - it_o) <= (0 if f64 else 2), (int(host(it)[0]), it_o)
    if f64:
        assert rel(host(lam), lam_o) < 1e-9
    else:                                      # both stopped by the same test: error against the converged fp64 solution
        check_f32(f"pcg to {tol} {tag}", host(lam), lam_o, co.pcg(S64, P64, g64, S, K, 1e-14, 600)[0])
    # deterministic: bitwise identical on a second run
    lam2, it2 = sol.pcg(dS, dP, dg, tol, 300)
    assert torch.equal(lam, lam2) and torch.equal(it, it2)
    mode = sol.get_option("last_mode")
    assert mode == opts.get("pcg_mode", _lib.PCG_RESIDENT)
    if (S, K, f64) in ((14, 50, True), (14, 37, True), (14, 41, True)) and not set(opts) - {"no_pair"}:
        assert sol.get_option("last_pair") == (0 if opts.get("no_pair") else 2)      # the fp64 mixed-rows kernel is what ran
        assert sol.get_option("last_groups") == 1 and sol.get_option("last_threads") == (512 if not opts else 64 * ((K * S + 63) // 64))
    if "pcg_groups" in opts:
        assert sol.get_option("last_groups") >= opts["pcg_groups"]
    if "dpp_rows" in opts:
        assert sol.get_option("last_dpp") == opts["dpp_rows"]
    elif mode == _lib.PCG_RESIDENT and S in (12, 14, 32) and sol.get_option("last_pair") == 0 and not opts:
        assert sol.get_option("last_dpp") == (1 if f64 else 0)                         # what auto takes: fp64
    sol.close()


def test_unused_boundary_blocks_are_never_read():
    """S[0].left and S[K-1].right are unwritten garbage in the reference (gpu_library.cu:40-41): the PCG
    must not touch them (gato_utils.cuh:157-174)."""
    S, C, K = 14, 7, 40
    s = synth.make_system(S, C, K, seed=4)
    Gd, Cd = co.convert(*s.csr_args()[:6], S, C, K, s.rho, np.float64)
    Sb, Pb, gam, _ = co.form_schur(Gd, Cd, s.g, s.c, S, C, K)
    Pb = co.form_ss(Sb, Pb, S, K)
    lam_o, it_o = co.pcg(Sb, Pb, gam, S, K, 1e-10, 200)
    Sb[:S * S] = np.nan
    Pb[:S * S] = np.nan
    Sb[-S * S:] = np.nan
    Pb[-S * S:] = np.nan
    for mode in (_lib.PCG_RESIDENT, _lib.PCG_STREAMING):
        sol = make_solver(S, C, K, np.float64)
        sol.set_option("pcg_mode", mode)
        lam, it = sol.pcg(sol.to_device(Sb), sol.to_device(Pb), sol.to_device(gam), 1e-10, 200)
        assert int(host(it)[0]) == it_o and rel(host(lam), lam_o) < 1e-9
        sol.close()


@pytest.mark.parametrize("S,C,K,dt", [(14, 7, 4096, np.float64), (14, 7, 4096, np.float32), (32, 16, 1024, np.float64),
                                      (32, 16, 1024, np.float32)])
def test_full_size_properties(S, C, K, dt):
    """BASELINE sizes: the returned (lambda, dz) satisfy the KKT equations
         (G + rho I) dz + C^T lambda = g,   C dz = c
    to solver tolerance (sparse fp64 residuals on the host), resident == streaming, and the iteration
    count equals the oracle's."""
    from scipy import sparse
    s = synth.make_system(S, C, K, seed=21)
    sol = make_solver(S, C, K, dt)
    dev = sol.upload_system(s)
    f64 = dt == np.float64
    tol, mi = (1e-12, 400) if f64 else (1e-5, 200)
    lam, dz = sol.new(S * K), sol.new(sol.N)
    sol.linsys(*dev, tol, mi, s.rho, lam, dz)
    torch.cuda.synchronize()
    sol.check_status()
    G = sparse.csr_matrix((s.G_val, s.G_col, s.G_row), shape=(s.N, s.N)) + s.rho * sparse.identity(s.N)
    Cm = sparse.csr_matrix((s.C_val, s.C_col, s.C_row), shape=(S * K, s.N))
    l, d = host(lam).astype(np.float64), host(dz).astype(np.float64)
    r1 = G @ d + Cm.T @ l - s.g
    r2 = Cm @ d - s.c
    scale = max(np.abs(s.g).max(), np.abs(l).max())
    assert np.abs(r1).max() / scale < (1e-10 if f64 else 1e-4)
    # the constraint residual is the Schur-system residual left by the exit test |r.Pinv r| < tol
    assert np.abs(r2).max() / max(np.abs(d).max(), 1.0) < (5e-6 if f64 else 1e-3)
    # against the oracle's whole solve: its iteration count; BASELINE's bar ||dz - dz_ref||inf < 1e-6 in fp64 (dz = O(1..10)
    # here, so the absolute bar is the tighter one); fp32 measured beside the fp32 oracle's own error (check_f32)
    it = int(np.frombuffer(_read_iters(sol), np.int32)[0])

    def rerun_on(solver):
        def rerun(tol_, mi_):
            a, b = solver.new(S * K), solver.new(solver.N)
            solver.linsys(*dev, tol_, mi_, s.rho, a, b)
            torch.cuda.synchronize()
            solver.check_status()
            return host(a), host(b)
        return rerun
    it_o = check_solve(f"full size {S}/{C}/{K}", s, S, C, K, dt, tol, mi, l, d, it, rerun=rerun_on(sol))
    if f64:
        _, dz_o, _ = co.linsys_solve(*s.csr_args(), S, C, K, tol, mi, s.rho, dtype=dt)
        assert np.abs(d - dz_o).max() < 1e-6, np.abs(d - dz_o).max()
    # streaming variant: the same checks against the oracle (not only against the resident launch)
    sol2 = make_solver(S, C, K, dt)
    sol2.set_option("pcg_mode", _lib.PCG_STREAMING)
    lam2, dz2 = sol2.new(S * K), sol2.new(sol2.N)
    sol2.linsys(*dev, tol, mi, s.rho, lam2, dz2)
    torch.cuda.synchronize()
    sol2.check_status()
    it2 = int(np.frombuffer(_read_iters(sol2), np.int32)[0])
    check_solve(f"full size {S}/{C}/{K} streaming", s, S, C, K, dt, tol, mi, host(lam2), host(dz2), it2, rerun=rerun_on(sol2))
    if f64:
        assert it2 == it_o and np.abs(host(dz2) - dz_o).max() < 1e-6
    sol.close()
    sol2.close()


def test_dropin_accepts_numpy_and_env_shape(monkeypatch):
    import gpu_library
    s = synth.make_system(14, 7, 50, seed=0)
    gpu_library.set_problem_size(14, 7, 50)
    try:
        l, dz = gpu_library.linsys_solve(s.G_row, s.G_col, s.G_val, s.C_row, s.C_col, s.C_val, s.g, s.c,
                                         np.zeros(700), 2, 1e-6, 100, False, s.rho)
    finally:
        gpu_library.clear_problem_size()
    check_solve("drop-in 14/7/50", s, 14, 7, 50, np.float32, 1e-6, 100, np.asarray(l), np.asarray(dz), gpu_library.last_stats()["iters"])
    with pytest.raises(ValueError):
        gpu_library.set_problem_size(14, 7, 49)
        try:
            gpu_library.linsys_solve(s.G_row, s.G_col, s.G_val, s.C_row, s.C_col, s.C_val, s.g, s.c,
                                     np.zeros(700), 1, 1e-6, 100, False, s.rho)
        finally:
            gpu_library.clear_problem_size()


def test_pybind11_module_runs_the_reference_test(golden_dir):
    """bindings/pybind11: the reference's own script flow (test_pendulum_5.py:25-37) against the pybind11
    module built over the C ABI, in a fresh interpreter that cannot see the ctypes drop-in."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import json, numpy as np, gpu_library
assert gpu_library.__file__.endswith('.so')
gold = json.load(open({os.path.join(golden_dir, 'pendulum.json')!r}))
i, e = gold['inputs'], gold['expected']
l, dz = gpu_library.linsys_solve(i['G_row'], i['G_col'], i['G_val'], i['C_row'], i['C_col'], i['C_val'], i['g_val'],
                                 i['c_val'], i['input_lambda'], i['testiters'], i['exit_tol'], i['max_iters'],
                                 i['warm_start'], i['rho'])
x = np.concatenate([e['dense_kkt_norho_dz'], e['dense_kkt_norho_lam']])
assert np.allclose(np.concatenate([dz, l]), x, rtol=1, atol=0.01)
assert np.abs(np.asarray(l) - e['lam']).max() / np.abs(e['lam']).max() < 5e-5
print('Test passed')
"""
    env = dict(os.environ, PYTHONPATH=os.path.join(root, "bindings", "pybind11", "build"), GATO_VERBOSE="1")
    r = subprocess.run([sys.executable, "-c", code], cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "Test passed" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    assert "first run PCG terminated in " in r.stdout and "avg time:" in r.stdout


def test_pybind11_module_at_iiwa_size(tmp_path):
    """the SHIPPED pybind11 module (bindings/pybind11, what a maintainer would install in place of gpu_library.cu)
    at the reference's own shape, in a fresh interpreter that sees the .so only: IIWA 14/7/50 (test_IIWA50.py:15-18) as Python
    lists, as numpy float64 / float32 arrays, with c_val as Python ints (test_pendulum_5.py:18 passes ints) and the CSR index
    arrays as int64, testiters = 3 - lambda, dz and the iteration count of every form against the C oracle; an index that does
    not fit 32 bits must raise instead of wrapping."""
    import subprocess
    import sys
    S, C, K = 14, 7, 50
    s = synth.make_system(S, C, K, seed=0)
    s.c = np.round(10.0 * s.c)                                 # integer-valued c: passed as Python ints below
    np.savez(tmp_path / "in.npz", **{k: getattr(s, k) for k in ("G_row", "G_col", "G_val", "C_row", "C_col", "C_val", "g", "c")}, rho=s.rho)
    code = f"""
import numpy as np, gpu_library
assert gpu_library.__file__.endswith('.so')
d = np.load({str(tmp_path / 'in.npz')!r})
names = ['G_row', 'G_col', 'G_val', 'C_row', 'C_col', 'C_val', 'g', 'c']
base = [d[n] for n in names]
lam0 = np.zeros(14 * 50)
forms = {{
    'lists': [a.tolist() for a in base] + [lam0.tolist()],
    'f64': [np.asarray(a, np.float64) if a.dtype.kind == 'f' else a for a in base] + [lam0],
    'f32': [np.asarray(a, np.float32) if a.dtype.kind == 'f' else np.asarray(a, np.int32) for a in base] + [lam0.astype(np.float32)],
    'ints_and_int64': [np.asarray(a, np.int64) if a.dtype.kind in 'iu' else a.tolist() for a in base[:7]] + [[int(v) for v in base[7]], [0] * 700],
}}
out = {{}}
for name, args in forms.items():
    l, dz = gpu_library.linsys_solve(*args, 3, 1e-6, 100, False, float(d['rho']))
    assert type(l) is list and type(dz) is list and type(l[0]) is float and len(l) == 700 and len(dz) == 21 * 50 - 7
    st = gpu_library.last_stats()
    assert len(st['ms']) == 3 and (st['S'], st['C'], st['K']) == (14, 7, 50)
    out[name + '_lam'], out[name + '_dz'], out[name + '_iters'] = np.asarray(l), np.asarray(dz), st['iters']
bad = [np.asarray(a, np.int64) if a.dtype.kind in 'iu' else a for a in base] + [lam0]
bad[1] = bad[1].copy(); bad[1][5] = 2 ** 40
try:
    gpu_library.linsys_solve(*bad, 1, 1e-6, 100, False, float(d['rho']))
    raise SystemExit('an over-range int64 index was accepted')
except OverflowError:
    pass
np.savez({str(tmp_path / 'out.npz')!r}, **out)
print('Test passed')
"""
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "bindings", "pybind11", "build"), GATO_VERBOSE="1")
    for k in ("GATO_STATE_SIZE", "GATO_CONTROL_SIZE", "GATO_KNOT_POINTS"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "Test passed" in r.stdout, r.stdout[-1500:] + r.stderr[-2500:]
    assert r.stdout.count("first run PCG terminated in ") == 4 and r.stdout.count("avg time:") == 4
    out = np.load(tmp_path / "out.npz")
    first = None
    for name in ("lists", "f64", "f32", "ints_and_int64"):
        lam, dz, it = out[name + "_lam"], out[name + "_dz"], int(out[name + "_iters"])
        check_solve(f"pybind11 module 14/7/50 as {name}", s, S, C, K, np.float32, 1e-6, 100, lam, dz, it)
        first = (lam, dz, it) if first is None else first
        assert np.array_equal(lam, first[0]) and np.array_equal(dz, first[1]) and it == first[2], name     # the same floats reach the device


@pytest.mark.parametrize("S,C,K,dt,opts", [(14, 7, 50, np.float64, {}), (14, 7, 50, np.float32, {}),
                                           (14, 7, 300, np.float64, {}), (2, 1, 40, np.float64, dict(pcg_threads=64)),
                                           (14, 7, 900, np.float32, {}),
                                           (14, 7, 300, np.float64, dict(dpp_rows=0)), (14, 7, 900, np.float32, dict(dpp_rows=1)),
                                           (32, 16, 200, np.float32, dict(dpp_rows=1)),
                                           (14, 7, 300, np.float64, dict(pcg_mode=_lib.PCG_STREAMING)),
                                           (32, 16, 40, np.float32, dict(pcg_mode=_lib.PCG_STREAMING)),
                                           (14, 7, 14500, np.float64, {}),                  # semi-resident launch
                                           (14, 7, 14500, np.float64, dict(pcg_semi=2))])   # ... without resident rows
def test_true_warm_start(S, C, K, dt, opts):
    """opt-in real warm start r0 = gamma - S lambda0 (the default stays the reference's
    no-op, D5).  Checked against the numpy restatement with the same lambda0."""
    s = synth.make_system(S, C, K, seed=17)
    out = o.linsys_solve(*s.csr_args(), S, C, K, 1e-30, 6, s.rho, dtype=dt, return_all=True)    # 6 iterations in
    lam0 = out["lam"]
    f64 = dt == np.float64
    tol = 1e-9 if f64 else 1e-4
    lam_w, it_w = o.pcg(out["S"], out["Pinv"], out["gamma"], S, K, tol, 300, lam0=lam0)
    lam_c, it_c = o.pcg(out["S"], out["Pinv"], out["gamma"], S, K, tol, 300)
    assert it_w < it_c                                              # the guess helps
    sol = make_solver(S, C, K, dt)
    for k, v in opts.items():
        sol.set_option(k, v)
    dS, dP, dg = sol.to_device(out["S"]), sol.to_device(out["Pinv"]), sol.to_device(out["gamma"])
    # default: reference behaviour, the initial guess is ignored
    lam = sol.to_device(lam0)
    lam, it = sol.pcg(dS, dP, dg, tol, 300, lam=lam)
    check_pcg(f"warm start ignored {S}/{C}/{K} {opts}", out["S"], out["Pinv"], out["gamma"], S, K, tol, 300, host(lam), int(host(it)[0]))
    # opt-in: true warm start
    sol.set_option("true_warm_start", 1)
    lam = sol.to_device(lam0)
    lam, it = sol.pcg(dS, dP, dg, tol, 300, lam=lam)
    assert abs(int(host(it)[0]) - it_w) <= (0 if f64 else 2), (int(host(it)[0]), it_w)
    if f64:
        assert rel(host(lam), lam_w) < 1e-9
    else:           # fp32: against the converged fp64 solution of the same fp32 matrices, beside the numpy restatement's error
        conv = co.pcg(out["S"].astype(np.float64), out["Pinv"].astype(np.float64), out["gamma"].astype(np.float64), S, K, 1e-14, 600)[0]
        check_f32(f"true warm start {S}/{C}/{K} {opts}", host(lam), lam_w, conv)
    # starting from the converged solution: exits at once
    lam = sol.to_device(lam_w)
    lam2, it = sol.pcg(dS, dP, dg, tol * 1e4, 300, lam=lam)
    assert int(host(it)[0]) <= 1                                                       # one more step is taken before the test
    if f64:
        assert rel(host(lam2), lam_w) < 1e-5
    else:
        check_f32(f"warm start from the solution {S}/{C}/{K} {opts}", host(lam2), lam_w, conv)
    sol.close()


@pytest.mark.parametrize("S,C,K,B,dt", [(14, 7, 50, 5, np.float64), (14, 7, 50, 7, np.float32), (2, 1, 5, 33, np.float64),
                                        (14, 7, 120, 3, np.float64), (32, 16, 6, 4, np.float32), (14, 7, 50, 300, np.float32),
                                        (14, 7, 20, 9, np.float64), (14, 7, 34, 4, np.float64), (32, 16, 16, 5, np.float64)])   # one DPP-row workgroup per system; 34: not in that layout
def test_batched_solves(S, C, K, B, dt):
    """B independent systems (shared sparsity, own values) in one call - every stage one
    launch for the whole batch, the PCG one workgroup per system.  Each system must equal its own oracle solve."""
    from gato_python_amd.solver import Solver
    systems = [synth.make_system(S, C, K, seed=100 + b) for b in range(B)]
    sol = Solver(S, C, K, dt, batch=B)
    dev = sol.upload_batch(systems)
    f64 = dt == np.float64
    tol, mi = (1e-10, 300) if f64 else (1e-5, 100)
    lam, dz = sol.new(B * S * K), sol.new(B * sol.N)
    iters = sol.new(B, torch.int32)
    sol.linsys_batched(*dev, tol, mi, systems[0].rho, lam, dz, iters)
    torch.cuda.synchronize()
    sol.check_status()
    lam_h, dz_h, it_h = host(lam).reshape(B, -1), host(dz).reshape(B, -1), host(iters)
    for b in (range(B) if B <= 40 else list(range(0, B, 37)) + [B - 1]):
        s = systems[b]
        check_solve(f"batch {S}/{C}/{K} system {b} of {B}", s, S, C, K, dt, tol, mi, lam_h[b], dz_h[b], int(it_h[b]))
    assert sol.get_option("batch") == B
    sol.close()


def test_random_resident_geometries():
    """Randomised launch geometries of the resident kernel (ragged knot splits, 1-knot workgroups, every thread
    count) against the oracle in fp64: same iteration count, lambda to 1e-9."""
    rng = np.random.default_rng(2024)
    S, C = 14, 7
    cache = {}
    for trial in range(36):
        K = int(rng.choice([3, 17, 50, 51, 97, 200, 333]))
        if K not in cache:
            s = synth.make_system(S, C, K, seed=K)
            Gd, Cd = co.convert(*s.csr_args()[:6], S, C, K, s.rho, np.float64)
            Sb, Pb, gam, _ = co.form_schur(Gd, Cd, s.g, s.c, S, C, K)
            Pb = co.form_ss(Sb, Pb, S, K)
            cache[K] = (Sb, Pb, gam) + co.pcg(Sb, Pb, gam, S, K, 1e-9, 300)
        Sb, Pb, gam, lam_o, it_o = cache[K]
        threads = int(rng.choice([64, 128, 192, 256, 320, 448, 512]))
        groups = int(rng.integers(0, min(K, 40) + 1))
        sol = make_solver(S, C, K, np.float64)
        sol.set_option("pcg_mode", _lib.PCG_RESIDENT)
        sol.set_option("no_single_lds", int(rng.integers(0, 2)))
        sol.set_option("pcg_threads", threads if groups == 0 else 0)
        sol.set_option("pcg_groups", groups)
        sol.set_option("dpp_rows", int(rng.integers(0, 2)))          # either row layout
        try:
            lam, it = sol.pcg(sol.to_device(Sb), sol.to_device(Pb), sol.to_device(gam), 1e-9, 300)
        except _lib.GatoError as e:
            assert e.code == -1, e          # geometry that does not fit is refused, never mis-run
            sol.close()
            continue
        assert int(host(it)[0]) == it_o, (K, threads, groups, int(host(it)[0]), it_o)
        assert rel(host(lam), lam_o) < 1e-8, (K, threads, groups)   # summation order differs with the split
        sol.close()


@pytest.mark.parametrize("S,C,K,dt", [(32, 16, 256, np.float32), (32, 16, 256, np.float64), (32, 16, 1024, np.float32), (32, 16, 9, np.float64)])
def test_dpp_rows_same_bits_as_the_lds_windows_where_the_geometry_is_the_same(S, C, K, dt):
    """At S = 32 a knot has 32 lanes in either row layout, so both take the same launch geometry and the same dot-product
    grouping: the DPP-row products (v_fmac_*_dpp row_newbcast chains) must then give the very bits of the LDS-window products
    (fp64: columns left to right; fp32: even + odd columns) - lambda, dz and iters."""
    s = synth.make_system(S, C, K, seed=5)
    out = []
    for dpp in (0, 1):
        sol = make_solver(S, C, K, dt)
        sol.set_option("dpp_rows", dpp)
        dev = sol.upload_system(s)
        lam, dz = sol.new(S * K), sol.new(sol.N)
        for tol, mi in ((1e-9 if dt == np.float64 else 1e-5, 200), (0.0, 17)):
            sol.linsys(*dev, tol, mi, s.rho, lam, dz)
            out.append((dpp, host(lam).copy(), host(dz).copy(), _read_iters(sol), sol.get_option("last_groups"), sol.get_option("last_threads")))
            assert sol.get_option("last_dpp") == dpp
        sol.close()
    for a, b in ((out[0], out[2]), (out[1], out[3])):
        assert a[4:] == b[4:], (a[4:], b[4:])
        assert a[3] == b[3] and a[1].tobytes() == b[1].tobytes() and a[2].tobytes() == b[2].tobytes()


def test_mixed_rows_kernel_every_size_it_serves():
    """pcg_single_f64m_kernel (fp64, one workgroup, two rows per lane in four of its eight waves, DPP rows in the others):
    every K it is selected for at 14/7 - 33 (the first size whose DPP rows no longer fit one 512-thread workgroup) to 50 -
    against the oracle in fp64, with and without a true warm start, and against the one-row-per-lane kernels on the same system."""
    S, C = 14, 7
    for K in range(33, 51):
        s = synth.make_system(S, C, K, seed=100 + K)
        Gd, Cd = co.convert(*s.csr_args()[:6], S, C, K, s.rho, np.float64)
        Sb, Pb, gam, _ = co.form_schur(Gd, Cd, s.g, s.c, S, C, K)
        Pb = co.form_ss(Sb, Pb, S, K)
        lam_o, it_o = co.pcg(Sb, Pb, gam, S, K, 1e-9, 300)
        res = {}
        for no_pair in (0, 1):
            sol = make_solver(S, C, K, np.float64)
            sol.set_option("no_pair", no_pair)
            lam, it = sol.pcg(sol.to_device(Sb), sol.to_device(Pb), sol.to_device(gam), 1e-9, 300)
            assert sol.get_option("last_pair") == (0 if no_pair else 2), (K, no_pair)
            assert sol.get_option("last_groups") == (1 if (not no_pair or K >= 37) else 2), (K, no_pair)    # 33..36 one row per lane: two workgroups
            assert int(host(it)[0]) == it_o, (K, no_pair)
            assert rel(host(lam), lam_o) < 1e-9, (K, no_pair)
            res[no_pair] = host(lam).copy()
            if not no_pair and K % 4 == 1:                      # true warm start: the numpy restatement with the same lambda0
                sol.set_option("true_warm_start", 1)
                l0 = lam_o * (1 + 1e-3 * np.cos(np.arange(S * K)))
                lam_w, it_w = o.pcg(Sb, Pb, gam, S, K, 1e-9, 300, lam0=l0)
                lam2, it2 = sol.pcg(sol.to_device(Sb), sol.to_device(Pb), sol.to_device(gam), 1e-9, 300, lam=sol.to_device(l0))
                assert int(host(it2)[0]) == it_w < it_o and rel(host(lam2), lam_w) < 1e-9, (K, int(host(it2)[0]), it_w, it_o)
            sol.close()
        assert rel(res[0], res[1]) < 1e-10, K


def test_one_xcd_placement_is_a_hint_only():
    """Launches of up to 32 workgroups are placed on ONE XCD (xcd_pack), and which of the eight hosts them is measured once
    per solver and geometry (xcd_sel = -1) or fixed by option.  Whatever the placement, the bits are the same."""
    S, C, K = 14, 7, 512
    s = synth.make_system(S, C, K, seed=5)
    ref = None
    # xcd_pack = 2, 3: the working blocks are dealt to two / three XCDs - the in-kernel placement check then finds different XCC
    # ids and keeps the agent-scope stores (the workgroup-scope stores of the one-XCD fast path would never be seen there)
    for opts in ({}, dict(xcd_sel=5), dict(xcd_sel=2), dict(xcd_pack=0), dict(xcd_pack=2), dict(xcd_pack=3)):
        sol = make_solver(S, C, K, np.float32)
        for k, v in opts.items():
            sol.set_option(k, v)
        dev = sol.upload_system(s)
        lam, dz = sol.new(S * K), sol.new(sol.N)
        for _ in range(2):                                   # (the solver measured its placement at creation: gato_solver_tune)
            sol.linsys(*dev, 1e-5, 60, s.rho, lam, dz)
            sol.check_status()
        sel = sol.get_option("last_xcd_sel")
        if "xcd_sel" in opts:
            assert sel == opts["xcd_sel"]
        elif opts.get("xcd_pack") == 0:
            assert sel == -1
        elif "xcd_pack" in opts:
            assert 0 <= sel <= 7
        else:
            assert 0 <= sel <= 7
        if ref is None:
            ref = (host(lam).copy(), host(dz).copy())
        assert np.array_equal(host(lam), ref[0]) and np.array_equal(host(dz), ref[1]), opts
        sol.close()


def test_pcg_entry_is_enqueue_only():
    """gato_pcg / gato_linsys_device are documented as asynchronous (include/gato_hip.h): enqueue only.  The XCD placement
    of the one-XCD launches is measured in gato_solver_create / gato_solver_tune (blocking, solver-owned scratch), never in
    the entry: with ~60 ms of earlier work queued on the stream the call must return while that work is still running -
    for the default geometry (measured at creation) and for a geometry nobody measured (runs on XCD 0) - and an explicit
    tune() afterwards changes the placement only, not a bit of the result, the caller's buffers or the sticky status."""
    import time
    S, C, K = 14, 7, 512
    s = synth.make_system(S, C, K, seed=3)
    sol = make_solver(S, C, K, np.float32)
    dev = sol.upload_system(s)
    lam, dz = sol.new(S * K), sol.new(sol.N)
    sol.linsys(*dev, 0.0, 40, s.rho, lam, dz)                     # warm the code objects up
    torch.cuda.synchronize()
    assert 0 <= sol.get_option("last_xcd_sel") <= 7                  # measured at creation, no launch of the caller needed
    ref = (host(lam).copy(), host(dz).copy())
    stream = torch.cuda.current_stream()
    t0 = time.perf_counter()                                         # what one tick of torch.cuda._sleep lasts on this box
    torch.cuda._sleep(2_000_000)
    torch.cuda.synchronize()
    per_tick = (time.perf_counter() - t0) / 2_000_000
    ticks = int(0.06 / per_tick)
    for geometry in ("default", "unmeasured"):
        if geometry == "unmeasured":
            sol.set_option("pcg_threads", 448)                       # another geometry: 16 workgroups, no measurement for it
        lam.zero_()
        dz.zero_()
        torch.cuda.synchronize()
        torch.cuda._sleep(ticks)
        t0 = time.perf_counter()
        sol.linsys(*dev, 0.0, 40, s.rho, lam, dz)
        dt_call = time.perf_counter() - t0
        busy = not stream.query()
        torch.cuda.synchronize()
        sol.check_status()
        assert busy and dt_call < 0.03, (geometry, busy, dt_call)     # returned while the queued work was still running
        if geometry == "default":
            assert np.array_equal(host(lam), ref[0]) and np.array_equal(host(dz), ref[1])
        else:
            assert sol.get_option("last_xcd_sel") == 0 and sol.get_option("last_groups") > 15
            unmeasured = (host(lam).copy(), host(dz).copy())
    # explicit tune for the new geometry: blocking, touches neither the caller's buffers nor the status
    lam.fill_(7.0)
    dz.fill_(7.0)
    sol.tune()
    assert float(lam.min()) == 7.0 and float(dz.max()) == 7.0
    sol.check_status()
    sol.linsys(*dev, 0.0, 40, s.rho, lam, dz)
    torch.cuda.synchronize()
    assert 0 <= sol.get_option("last_xcd_sel") <= 7
    assert np.array_equal(host(lam), unmeasured[0]) and np.array_equal(host(dz), unmeasured[1])
    sol.close()


def test_c_host_example():
    """examples/solve_pendulum.c: the reference's test case from plain C over the C ABI (no Python in the path)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "examples"), "-s"])
    r = subprocess.run([os.path.join(root, "examples", "solve_pendulum")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "Test passed" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("S,C,K,dt,dq", [(14, 7, 50, np.float64, False), (14, 7, 9, np.float32, True), (32, 16, 5, np.float64, True)])
def test_direct_block_input(S, C, K, dt, dq):
    """per-knot blocks handed over in the dense layouts (no CSR scatter) give the same
    (lambda, dz) as the CSR path."""
    s = synth.make_system(S, C, K, seed=31, dense_q=dq)
    Gd0, Cd = co.convert(*s.csr_args()[:6], S, C, K, 0.0, dt)            # blocks without rho
    sol = make_solver(S, C, K, dt)
    f64 = dt == np.float64
    tol, mi = (1e-10, 300) if f64 else (1e-5, 100)
    lam, dz = sol.new(S * K), sol.new(sol.N)
    sol.linsys_blocks(sol.to_device(Gd0), sol.to_device(Cd), sol.to_device(s.g), sol.to_device(s.c), tol, mi, s.rho, lam, dz)
    torch.cuda.synchronize()
    check_solve(f"block input {S}/{C}/{K}", s, S, C, K, dt, tol, mi, host(lam), host(dz), int(np.frombuffer(_read_iters(sol), np.int32)[0]))
    sol.close()


@pytest.mark.parametrize("S,C,K,dt,opts", [(14, 7, 50, np.float64, {}), (14, 7, 50, np.float32, {}),
                                           (14, 7, 50, np.float64, dict(pcg_threads=192)),       # 10 knots per workgroup
                                           (14, 7, 512, np.float64, {}), (14, 7, 512, np.float32, {}),
                                           (14, 7, 4096, np.float32, {}), (14, 7, 4096, np.float64, {}),
                                           (32, 16, 300, np.float64, {}), (2, 1, 900, np.float64, dict(pcg_threads=64))])
def test_single_reduction_variant(S, C, K, dt, opts):
    """Opt-in Chronopoulos-Gear variant (one hand-off per iteration): equals its own numpy restatement to rounding,
    and the reference recurrence's solution to solver tolerance; iteration counts within one of each other."""
    s = synth.make_system(S, C, K, seed=23)
    Gd, Cd = co.convert(*s.csr_args()[:6], S, C, K, s.rho, dt)
    Sb, Pb, gam, _ = co.form_schur(Gd, Cd, s.g, s.c, S, C, K)
    Pb = co.form_ss(Sb, Pb, S, K)
    f64 = dt == np.float64
    tol = 1e-9 if f64 else 1e-4
    lam_cg, it_cg = o.pcg_single_reduction(Sb, Pb, gam, S, K, tol, 300)
    lam_ref, it_ref = co.pcg(Sb, Pb, gam, S, K, tol, 300)
    sol = make_solver(S, C, K, dt)
    sol.set_option("pcg_variant", 1)
    for k, v in opts.items():
        sol.set_option(k, v)
    dS, dP, dg = sol.to_device(Sb), sol.to_device(Pb), sol.to_device(gam)
    lam, it = sol.pcg(dS, dP, dg, tol, 300)
    assert sol.get_option("last_variant") == 1
    assert abs(int(host(it)[0]) - it_cg) <= (0 if f64 else 2), (int(host(it)[0]), it_cg, it_ref)
    assert abs(int(host(it)[0]) - it_ref) <= 2
    if f64:
        assert rel(host(lam), lam_cg) < 1e-8 and rel(host(lam), lam_ref) < 1e-6
    else:           # fp32: the converged fp64 solution of the same matrices is the truth for both recurrences
        conv = co.pcg(Sb.astype(np.float64), Pb.astype(np.float64), gam.astype(np.float64), S, K, 1e-14, 600)[0]
        check_f32(f"single-reduction variant vs its restatement {S}/{C}/{K}", host(lam), lam_cg, conv)
        check_f32(f"single-reduction variant vs the reference recurrence {S}/{C}/{K}", host(lam), lam_ref, conv)
    lam2, it2 = sol.pcg(dS, dP, dg, tol, 300)
    assert torch.equal(lam, lam2) and torch.equal(it, it2)        # deterministic
    sol.close()


@pytest.mark.parametrize("S,C,dt,threads,ks", [(32, 16, np.float64, 0, range(17, 130)), (14, 7, np.float64, 128, range(11, 100)),
                                               (6, 3, np.float64, 64, range(9, 80))])
def test_single_reduction_variant_every_k_of_a_range(S, C, dt, threads, ks):
    """Every K of a range through the single-reduction kernel, several workgroups: the K whose even split would leave the last
    workgroup a single knot (31, 37, ... 61 at 32/16 in workgroups of 6) run the BALANCED split (sizes differing by one) instead
    of falling back to the default recurrence; each solve against the numpy restatement of the recurrence."""
    balanced = 0
    for K in ks:
        s = synth.make_system(S, C, K, seed=500 + K)
        Gd, Cd = co.convert(*s.csr_args()[:6], S, C, K, s.rho, dt)
        Sb, Pb, gam, _ = co.form_schur(Gd, Cd, s.g, s.c, S, C, K)
        Pb = co.form_ss(Sb, Pb, S, K)
        lam_cg, it_cg = o.pcg_single_reduction(Sb, Pb, gam, S, K, 1e-9, 300)
        sol = make_solver(S, C, K, dt)
        sol.set_option("pcg_variant", 1)
        if threads:
            sol.set_option("pcg_threads", threads)
        lam, it = sol.pcg(sol.to_device(Sb), sol.to_device(Pb), sol.to_device(gam), 1e-9, 300)
        W, T = sol.get_option("last_groups"), sol.get_option("last_threads")
        assert sol.get_option("last_variant") == 1, (K, W, T)
        assert int(host(it)[0]) == it_cg and rel(host(lam), lam_cg) < 1e-8, (K, W, int(host(it)[0]), it_cg)
        if W > 1:
            kpw = -(-K // W)
            balanced += K - (W - 1) * kpw == 1
        sol.close()
    assert balanced >= 3, balanced                          # the range does hold such K


def test_extra_shape_library():
    """a shape added at build time (EXTRA_SHAPES) gets the GENERIC launch bounds - S = 16: the fp32 two-rows-per-lane
    kernel at 256 threads, i.e. FOUR waves under block sums that read eight waves' partials.  build() compiles a one-shape
    (16, 8) library beside the product (build/ab/libgato_hip_s16.so); whole solves through it (one workgroup, several, batches,
    both types) against the C oracle, in a process of its own (the library is chosen before the package loads)."""
    import subprocess
    import sys
    lib = os.path.join(ROOT, "build", "ab", "libgato_hip_s16.so")
    assert os.path.exists(lib), "build() makes it: python -c 'import __graft_entry__ as g; g.build()'"
    env = dict(os.environ, GATO_HIP_LIB=lib, GATO_NO_TUNE="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "extra_shape_worker.py"), "16", "8"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "extra shape ok 16 8" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    assert "('float32', 1, 1, 256)" in r.stdout, r.stdout[-1500:]      # the four-wave two-rows-per-lane launch did run


@pytest.mark.parametrize("S,C,K", [(4, 2, 30), (6, 3, 100), (12, 6, 64), (12, 6, 700)])
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_other_compiled_shapes(S, C, K, dt):
    """The extra (STATE_SIZE, CONTROL_SIZE) instantiations of the default build: whole solve vs the oracle."""
    s = synth.make_system(S, C, K, seed=41, dense_q=True)
    sol = make_solver(S, C, K, dt)
    dev = sol.upload_system(s)
    f64 = dt == np.float64
    tol, mi = (1e-10, 400) if f64 else (1e-5, 150)
    lam, dz = sol.new(S * K), sol.new(sol.N)
    sol.linsys(*dev, tol, mi, s.rho, lam, dz)
    torch.cuda.synchronize()
    sol.check_status()
    check_solve(f"shape {S}/{C}/{K}", s, S, C, K, dt, tol, mi, host(lam), host(dz), int(np.frombuffer(_read_iters(sol), np.int32)[0]))
    sol.close()


@pytest.mark.parametrize("opts", [{}, dict(no_single_lds=1), dict(pcg_mode=_lib.PCG_STREAMING), dict(pcg_variant=1),
                                  dict(pcg_groups=3)])
def test_eta_history(opts):
    """record_eta: the residual measure eta = r.Pinv r per iteration (what the reference prints under DEBUG_MODE,
    gato_pcg.cuh:397-400) equals the oracle's history."""
    S, C, K = 14, 7, 50
    s = synth.make_system(S, C, K, seed=3)
    out = o.linsys_solve(*s.csr_args(), S, C, K, 1e-9, 200, s.rho, dtype=np.float64, return_all=True)
    sol = make_solver(S, C, K, np.float64)
    sol.set_option("record_eta", 1)
    for k, v in opts.items():
        sol.set_option(k, v)
    lam, it = sol.pcg(sol.to_device(out["S"]), sol.to_device(out["Pinv"]), sol.to_device(out["gamma"]), 1e-9, 200)
    n = int(host(it)[0])
    hist = sol.eta_history(n + 1)
    ref = np.asarray(out["eta"])
    assert abs(n - out["iters"]) <= (2 if opts.get("pcg_variant") else 0)
    m = min(len(ref), len(hist)) - 1
    tol = 1e-5 if opts.get("pcg_variant") else 1e-8
    assert np.allclose(hist[:m], ref[:m], rtol=tol, atol=1e-12), (hist[:5], ref[:5])
    sol.close()


@pytest.mark.parametrize("S,C,K,seed,dq", CASES + [(14, 7, 600, 3, False), (12, 6, 33, 4, True)])
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_fused_assembly_is_bit_identical_to_the_stage_kernels(S, C, K, seed, dq, dt):
    """gato_linsys_device assembles small problems in ONE launch (CSR gather + inversions + Schur + stair); the stage
    entries (gato_convert / gato_form_schur / gato_form_ss, checked against the oracle above) are separate launches.
    Same arithmetic per block, so every work buffer must match bit for bit."""
    s = system(S, C, K, seed, dq) if K > 1 else synth.blocks_to_csr(*synth.make_blocks(S, C, 1, seed, dq))
    names = ["G_dense", "C_dense", "Ginv", "S", "Pinv", "gamma", "lam", "dz"]
    ref = None
    for opts in (dict(asm_mode=1), dict(asm_mode=2)):
        sol = make_solver(S, C, K, dt)
        for k, v in opts.items():
            sol.set_option(k, v)
        dev = sol.upload_system(s)
        for _ in range(2):
            sol.linsys(*dev, 1e-8, 50, s.rho)
            sol.check_status()
        got = {n: sol.read_buffer(n) for n in names}
        assert sol.get_option("last_asm_fused") == {1: 0, 2: 1}[opts["asm_mode"]]
        sol.close()
        if ref is None:
            ref = got
            continue
        for n in names:
            assert np.array_equal(got[n], ref[n], equal_nan=True), (opts, n)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_fused_assembly_batched_and_blocks(dt):
    S, C, K, B = 14, 7, 20, 9
    systems = [synth.make_system(S, C, K, seed=10 + b, dense_q=False) for b in range(B)]
    names = ["G_dense", "Ginv", "S", "Pinv", "gamma", "lam", "dz"]
    from gato_python_amd.solver import Solver
    ref = None
    for mode in (1, 2):
        sol = Solver(S, C, K, dt, batch=B)
        sol.set_option("asm_mode", mode)
        dev = sol.upload_batch(systems)
        lam, dz = sol.new(B * S * K), sol.new(B * sol.N)
        sol.linsys_batched(*dev, 1e-8, 60, systems[0].rho, lam, dz)
        sol.check_status()
        got = {n: sol.read_buffer(n) for n in names if n not in ("lam", "dz")}
        got["lam"], got["dz"] = host(lam), host(dz)
        sol.close()
        if ref is None:
            ref = got
            continue
        for n in names:
            assert np.array_equal(got[n], ref[n]), n
    # direct block input (mode 2 of the fused launch) against the stage kernels
    s = systems[0]
    Gd_o, Cd_o = co.convert(*s.csr_args()[:6], S, C, K, 0.0, dt)
    ref = None
    for mode in (1, 2, 3):
        sol = make_solver(S, C, K, dt)
        sol.set_option("asm_mode", mode)
        sol.linsys_blocks(sol.to_device(Gd_o), sol.to_device(Cd_o), sol.to_device(s.g), sol.to_device(s.c), 1e-8, 60, s.rho)
        sol.check_status()
        got = {n: sol.read_buffer(n) for n in names}
        sol.close()
        if ref is None:
            ref = got
            continue
        for n in names:
            assert np.array_equal(got[n], ref[n]), n


@pytest.mark.parametrize("variant", [0, 1])
def test_handoff_epochs_across_launches_and_counter_wrap(variant):
    """Multi-workgroup launches never re-zero the hand-off granules: each launch gets a fresh epoch range from the
    solver's counter.  Repeated solves, a change of launch geometry in between, and the counter's wrap (test hook
    pcg_epoch) must all give the first launch's result bit for bit."""
    S, C, K, dt = 14, 7, 600, np.float32
    s = system(S, C, K, 21)
    sol = make_solver(S, C, K, dt)
    sol.set_option("pcg_variant", variant)
    dev = sol.upload_system(s)
    lam, dz = sol.new(S * K), sol.new(sol.N)

    def solve():
        sol.linsys(*dev, 1e-7, 80, s.rho, lam, dz)
        sol.check_status()
        return host(lam).copy()

    ref = solve()
    assert sol.get_option("last_groups") > 1
    for _ in range(3):
        assert np.array_equal(solve(), ref)
    sol.set_option("pcg_threads", 256)                 # other geometry: other slot layout over the same granules
    other = solve()
    assert rel(other, ref) < 1e-3
    sol.set_option("pcg_threads", 0)
    assert np.array_equal(solve(), ref)
    sol.set_option("pcg_epoch", -300)                  # 300 epochs before 2^32: the next launch must start over
    for _ in range(3):
        assert np.array_equal(solve(), ref)
    sol.close()


@pytest.mark.parametrize("K", [5, 64, 700])
def test_kkt_producer_through_the_hip_path(K):
    """the upstream producer (gato_python_amd/kkt.py: one linearisation of a pendulum OCP around a rolled-out
    trajectory, the problem family of the reference's own fixture) solved by the HIP path in f64 against the dense solve."""
    from gato_python_amd import kkt
    rng = np.random.default_rng(K)
    plant, dt = kkt.PendulumPlant(), 0.02
    u = 0.8 * rng.standard_normal((K - 1, 1))
    x = kkt.rollout(plant, (0.2, 0.0), u, dt)
    p = kkt.get_kkt(plant, x, u, (0.25, 0.05), (np.pi, 0.0), dt, np.diag([1.0, 0.3]), np.array([[0.1]]), np.diag([100.0, 30.0]))
    sol = make_solver(2, 1, K, np.float64)
    dev = sol.upload_system(p)
    lam, dz = sol.new(2 * K), sol.new(sol.N)
    sol.linsys(*dev, 1e-18, 2000, p.rho, lam, dz)
    sol.check_status()
    dz_d, lam_d = synth.dense_kkt_solve(p)
    assert np.abs(host(dz) - dz_d).max() < 1e-6 * max(1.0, np.abs(dz_d).max())
    assert rel(host(lam), lam_d) < 1e-6
    lam_o, dz_o, it_o = o.linsys_solve(*p.csr_args(), 2, 1, K, 1e-18, 2000, p.rho, np.float64)[:3]
    assert rel(host(lam), lam_o) < 1e-8
    sol.close()


@pytest.mark.parametrize("K,want", [(58000, 1), (59000, 3)])
def test_fp64_ring_is_auto_selected_past_its_measured_crossover(K, want):
    """in fp64 the LDS-DMA ring is what auto (pcg_semi = -1) takes once S + Pinv of a 14/7 system pass 550 MB
    (K = 58 461; measured cross-over against the semi-resident launch, profiles/r05_ring_crossover.log) - below it the
    semi-resident launch.  Both sides of the rule against the C oracle: exit iteration, lambda, dz."""
    S, C, dt = 14, 7, np.float64
    s = system(S, C, K, 23)
    sol = make_solver(S, C, K, dt)
    dev = sol.upload_system(s)
    lam, dz = sol.new(S * K), sol.new(sol.N)
    sol.linsys(*dev, 1e-9, 40, s.rho, lam, dz)
    sol.check_status()
    assert sol.get_option("last_mode") == _lib.PCG_RESIDENT and sol.get_option("last_semi") == want, sol.get_option("last_semi")
    it = int(np.frombuffer(_read_iters(sol), np.int32)[0])
    check_solve(f"auto launch (variant {want}) at {S}/{C}/{K} fp64", s, S, C, K, dt, 1e-9, 40, host(lam), host(dz), it, f64_tol=1e-9)
    sol.close()


@pytest.mark.parametrize("S,C,K,dt,which", [(14, 7, 14000, np.float32, 1), (14, 7, 20011, np.float64, 1), (32, 16, 5003, np.float32, 1),
                                            (12, 6, 30000, np.float32, 1), (14, 7, 60000, np.float32, 1),
                                            (14, 7, 14000, np.float32, 2), (14, 7, 20011, np.float64, 2), (32, 16, 3001, np.float64, 2),
                                            (2, 1, 140000, np.float32, 2),
                                            (14, 7, 14000, np.float32, 3), (14, 7, 20011, np.float64, 3), (14, 7, 131072, np.float32, 3),
                                            (14, 7, 13901, np.float32, 3), (14, 7, 65531, np.float64, 3),
                                            (32, 16, 5003, np.float32, 3), (32, 16, 30011, np.float32, 3)])   # the ring at S = 32: 2-block-row tiles, 12 rows per lane
def test_semi_resident_kernel_matches_the_streaming_kernels(S, C, K, dt, which):
    """K beyond the register file: one persistent launch whose workgroups keep part of their knots' matrix rows in
    registers and re-read the rest from memory every product (gato_pcg_resident.hip, XR > 0), against the streaming
    kernels (two launches per iteration) on the same assembled system: same iterates, same exit iteration.
    which = 1: boundary and leading knots register-resident; 2: the variant without resident rows (what fp64 at S = 32 gets);
    3: the LDS-DMA ring (gato_pcg_dma.hip: vectors in registers, block rows streamed through LDS), including ragged last
    workgroups and tiles (K = 13901, K = 65531)."""
    s = system(S, C, K, 17)
    f64 = dt == np.float64
    tol = 1e-9 if f64 else 1e-4
    res = {}
    for semi in (which, 0):
        sol = make_solver(S, C, K, dt)
        sol.set_option("pcg_semi", semi)
        sol.set_option("record_eta", 1)
        dev = sol.upload_system(s)
        lam, dz = sol.new(S * K), sol.new(sol.N)
        sol.linsys(*dev, tol, 40, s.rho, lam, dz)
        sol.check_status()
        it = int(np.frombuffer(_read_iters(sol), np.int32)[0])
        assert sol.get_option("last_semi") == semi
        assert sol.get_option("last_mode") == (_lib.PCG_RESIDENT if semi else _lib.PCG_STREAMING)
        res[semi] = (host(lam).copy(), host(dz).copy(), it, sol.eta_history(min(it + 1, 40)))
        if semi:                                              # a second solve on the same solver: fresh epochs, same bits
            lam2, dz2 = sol.new(S * K), sol.new(sol.N)
            sol.linsys(*dev, tol, 40, s.rho, lam2, dz2)
            sol.check_status()
            assert np.array_equal(host(lam2), res[semi][0])

            # DIRECTLY against the C oracle (these launches used to meet only the streaming kernels):
            # the oracle's exit iteration and solution, and 12 fixed iterations against the oracle's iterates
            def rerun(tol_, mi_):
                a, b = sol.new(S * K), sol.new(sol.N)
                sol.linsys(*dev, tol_, mi_, s.rho, a, b)
                sol.check_status()
                assert sol.get_option("last_semi") == semi
                return host(a), host(b)
            tag = f"persistent launch variant {which} at {S}/{C}/{K}"
            check_solve(tag, s, S, C, K, dt, tol, 40, res[semi][0], res[semi][1], it, it_slack=1, rerun=rerun, f64_tol=1e-9)
            lam12, dz12 = rerun(0.0, 12)
            lam_o, dz_o, it_o = co.linsys_solve(*s.csr_args(), S, C, K, 0.0, 12, s.rho, dtype=dt)
            if f64:
                assert rel(lam12, lam_o) < 1e-10 and rel(dz12, dz_o) < 1e-10
            else:
                s64, rho32 = _f32_truth_inputs(s)
                lam_t, dz_t, _ = co.linsys_solve(*s64.csr_args(), S, C, K, 0.0, 12, rho32, dtype=np.float64)
                check_f32(f"lambda after 12 iterations, {tag}", lam12, lam_o, lam_t)
                check_f32(f"dz after 12 iterations, {tag}", dz12, dz_o, dz_t)
        sol.close()
    # and against the streaming kernels on the same assembled system: same iterates, same exit iteration
    (la, da, ia, ea), (lb, db, ib, eb) = res[which], res[0]
    assert abs(ia - ib) <= (0 if f64 else 1), (ia, ib)
    n = min(len(ea), len(eb), 8)
    assert np.allclose(ea[:n], eb[:n], rtol=1e-9 if f64 else 2e-3)          # eta = r.Pinv r: a cancelling sum in fp32
    if f64:
        assert ia == ib and rel(la, lb) < 1e-10 and rel(da, db) < 1e-10


def _read_iters(sol):
    import ctypes as ct
    buf = (ct.c_int * 1)()
    torch.cuda.synchronize()
    rc = ct.CDLL("libamdhip64.so").hipMemcpy(buf, ct.c_void_p(sol.buffer_ptr(8)), 4, 2)
    assert rc == 0
    return bytes(buf)


@pytest.mark.parametrize("S,C,K,dt,tol,mi,opts", [
    (14, 7, 50, np.float64, 1e-9, 0, {}), (14, 7, 50, np.float64, 1e-9, 1, {}), (14, 7, 50, np.float64, 1e30, 50, {}),
    (14, 7, 300, np.float64, 1e30, 50, {}), (14, 7, 300, np.float64, 1e-9, 0, {}),
    (14, 7, 300, np.float64, 1e-9, 0, dict(pcg_mode=_lib.PCG_STREAMING)), (14, 7, 300, np.float64, 1e30, 5, dict(pcg_mode=_lib.PCG_STREAMING)),
    (2, 1, 2, np.float64, 1e-12, 20, dict(pcg_groups=2)), (2, 1, 3, np.float64, 1e-12, 20, dict(pcg_groups=3)),
    (2, 1, 1, np.float64, 1e-12, 20, {}), (14, 7, 2, np.float32, 1e-6, 20, dict(pcg_groups=2)),
    (14, 7, 300, np.float64, 1e-9, 0, dict(pcg_variant=1)), (14, 7, 300, np.float64, 1e30, 5, dict(pcg_variant=1)),
    (14, 7, 20000, np.float32, 1e30, 5, {}), (14, 7, 20000, np.float32, 1e-4, 0, {}), (14, 7, 20000, np.float32, 1e-4, 0, dict(pcg_semi=2))])
def test_pcg_degenerate_iteration_counts_and_geometries(S, C, K, dt, tol, mi, opts):
    """max_iters = 0 and 1, an exit test that is true at once (returned iters = 0 after ONE step, gato_pcg.cuh:404-411), one
    knot per workgroup, one-knot systems - in every kernel family, against the oracle."""
    s = synth.make_system(S, C, K, seed=1) if K > 1 else synth.blocks_to_csr(*synth.make_blocks(S, C, 1, 1, False))
    Gd, Cd = co.convert(*s.csr_args()[:6], S, C, K, s.rho, dt)
    Sb, Pb, gam, _ = co.form_schur(Gd, Cd, s.g, s.c, S, C, K)
    Pb = co.form_ss(Sb, Pb, S, K)
    lam_o, it_o = co.pcg(Sb, Pb, gam, S, K, tol, mi)
    sol = make_solver(S, C, K, dt)
    for k, v in opts.items():
        sol.set_option(k, v)
    lam, it = sol.pcg(sol.to_device(Sb), sol.to_device(Pb), sol.to_device(gam), tol, mi)
    assert int(host(it)[0]) == it_o
    scale = max(float(np.abs(lam_o).max()), 1e-30)
    if dt == np.float64:
        assert float(np.abs(host(lam) - lam_o).max()) / scale < 1e-10
    elif scale > 1e-30:       # fp32: same number of steps as the oracle (asserted above)
        S64, P64, g64 = Sb.astype(np.float64), Pb.astype(np.float64), gam.astype(np.float64)
        if it_o == mi or tol >= 1.0:      # a fixed number of steps (no exit / exit at once): the fp64 iterates after as many
            truth = co.pcg(S64, P64, g64, S, K, 0.0, mi if it_o == mi else it_o + 1)[0]
        else:                             # stopped by the exit test: the converged fp64 solution (the stopping error dominates)
            truth = co.pcg(S64, P64, g64, S, K, 1e-14, 600)[0]
        # 14/7/2 in fp32 is 28 unknowns run for 20 iterations: from iteration 10 on the fp32 iterates of ANY summation order are
        # 1e-3..1e-2 away from the fp64 iterate of the same index (tools/past_convergence.py: at n = 20 C order 7e-5, numpy order
        # 8e-4, GPU 4e-4..2e-3 depending on the launch geometry) while all of them are equally far from the CONVERGED solution
        # (1.7e-3 / 1.8e-3 / 2.3e-3): that is the meaningful reference there
        second = o.pcg(Sb, Pb, gam, S, K, tol, mi)[0] if K <= 2 else None
        if K <= 2:
            truth = co.pcg(S64, P64, g64, S, K, 1e-14, 600)[0]
        check_f32(f"degenerate {S}/{C}/{K} tol {tol} max_iters {mi} {opts}", host(lam), lam_o, truth, second_order32=second)
    else:
        assert not np.any(host(lam))
    sol.close()


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_csr_with_unsorted_columns_and_explicit_zeros(dt):
    """The scatter takes CSR rows in any column order and with structurally present zeros (what a caller that builds CSR
    by hand may pass; scipy's csr_matrix(dense) gives sorted, zero-free rows): both the per-knot gather kernel and the
    fused assembly launch must give the oracle's dense blocks bit for bit."""
    S, C, K = 14, 7, 12
    s = synth.make_system(S, C, K, seed=31, dense_q=True)
    rng = np.random.default_rng(5)

    def shuffle(indptr, indices, data):
        idx, dat, ptr = [], [], [0]
        for r in range(len(indptr) - 1):
            cols = list(indices[indptr[r]:indptr[r + 1]]); vals = list(data[indptr[r]:indptr[r + 1]])
            perm = rng.permutation(len(cols))
            idx += [cols[i] for i in perm]; dat += [vals[i] for i in perm]
            ptr.append(len(idx))
        return np.asarray(ptr, np.int32), np.asarray(idx, np.int32), np.asarray(dat, np.float64)

    G_row, G_col, G_val = shuffle(s.G_row, s.G_col, s.G_val)
    C_row, C_col, C_val = shuffle(s.C_row, s.C_col, s.C_val)
    G_val[rng.random(len(G_val)) < 0.1] = 0.0                  # explicit zeros off the diagonal are just values
    s2 = synth.KKTSystem(S, C, K, G_row, G_col, G_val, C_row, C_col, C_val, s.g, s.c, s.rho)
    Gd_o, Cd_o = co.convert(*s2.csr_args()[:6], S, C, K, s2.rho, dt)
    sol = make_solver(S, C, K, dt)
    dev = sol.upload_system(s2)
    Gd, Cd = sol.convert(*dev[:6], s2.rho)
    assert np.array_equal(host(Gd), Gd_o) and np.array_equal(host(Cd), Cd_o)
    for mode in (1, 2):
        sol.set_option("asm_mode", mode)
        sol.linsys(*dev, 1e-8, 5, s2.rho)
        assert np.array_equal(sol.read_buffer("G_dense"), Gd_o) and np.array_equal(sol.read_buffer("C_dense"), Cd_o)
    sol.close()


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_csr_with_structurally_empty_rows(dt):
    """RAGGED / EMPTY rows: rows of C with no dynamics entries at all (only the identity on x_{k+1} stays, or nothing), rows of G
    reduced to their diagonal, whole knots whose B block is structurally empty - row pointers repeat, entry counts per knot
    differ.  Dense blocks bit for bit (gather kernel, stage path, fused launch) and the whole solve against the oracle."""
    S, C, K = 14, 7, 11
    s = synth.make_system(S, C, K, seed=43, dense_q=True)
    rng = np.random.default_rng(11)
    n = S + C

    def thin(indptr, indices, data, keep):
        idx, dat, ptr = [], [], [0]
        for r in range(len(indptr) - 1):
            for e in range(indptr[r], indptr[r + 1]):
                if keep(r, int(indices[e])):
                    idx.append(indices[e]); dat.append(data[e])
            ptr.append(len(idx))
        return np.asarray(ptr, np.int32), np.asarray(idx, np.int32), np.asarray(dat, np.float64)

    diag_only = set(int(r) for r in rng.choice(n * K - C, 40, replace=False))
    G_row, G_col, G_val = thin(s.G_row, s.G_col, s.G_val, lambda r, c: r == c or (r not in diag_only and c not in diag_only))   # (symmetric)
    bare = set(int(r) for r in rng.choice(np.arange(S, S * K), 25, replace=False))        # C rows that keep only their identity entry
    gone = set(int(r) for r in rng.choice(np.arange(S, S * K), 6, replace=False))          # ... and rows with NO entry at all
    noB = {3, 7}                                                                            # knots whose B block is structurally empty

    def keep_c(r, c):
        if r in gone:
            return False
        br = r // S - 1
        ident = c == (br + 1) * n + r % S
        if r in bare:
            return ident
        if r >= S and br in noB and br * n + S <= c < (br + 1) * n:
            return False
        return True
    C_row, C_col, C_val = thin(s.C_row, s.C_col, s.C_val, keep_c)
    assert np.any(np.diff(C_row) == 0) and np.any(np.diff(C_row) == 1) and len(C_val) < len(s.C_val)
    s2 = synth.KKTSystem(S, C, K, G_row, G_col, G_val, C_row, C_col, C_val, s.g, s.c, s.rho)
    Gd_o, Cd_o = co.convert(*s2.csr_args()[:6], S, C, K, s2.rho, dt)
    sol = make_solver(S, C, K, dt)
    dev = sol.upload_system(s2)
    Gd, Cd = sol.convert(*dev[:6], s2.rho)
    assert np.array_equal(host(Gd), Gd_o) and np.array_equal(host(Cd), Cd_o)
    f64 = dt == np.float64
    tol = 1e-10 if f64 else 1e-5
    for mode in (1, 2):
        sol.set_option("asm_mode", mode)
        lam, dz = sol.new(S * K), sol.new(sol.N)
        sol.linsys(*dev, tol, 200, s2.rho, lam, dz)
        sol.check_status()
        assert np.array_equal(sol.read_buffer("G_dense"), Gd_o) and np.array_equal(sol.read_buffer("C_dense"), Cd_o)
        check_solve(f"structurally empty rows 14/7/{K} asm_mode {mode}", s2, S, C, K, dt, tol, 200, host(lam), host(dz),
                    int(np.frombuffer(_read_iters(sol), np.int32)[0]), f64_tol=1e-9)
    sol.close()


@pytest.mark.parametrize("S,C,K", [(14, 7, 12), (2, 1, 9), (32, 16, 5)])
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_csr_rows_with_duplicate_columns_keep_the_last_entry(S, C, K, dt):
    """COLLISIONS: a CSR row that holds the same column twice (legal CSR; scipy only removes such entries on sum_duplicates()).
    The reference walks a row with one thread, so the LAST entry in storage order wins - its value, and its own rho if it is
    the diagonal (gato_schur.cuh:689-702, :733-741).  The scatter here is a thread per entry: it must give the same blocks bit
    for bit - duplicates next to each other and far apart, in shuffled rows, in G (diagonal and off-diagonal) and in C, through
    the per-knot gather kernel, the stage path of a whole solve and the fused assembly launch; a batch shares the pattern."""
    s = synth.make_system(S, C, K, seed=41, dense_q=True)
    rng = np.random.default_rng(9)

    def with_duplicates(indptr, indices, data, frac):
        idx, dat, ptr = [], [], [0]
        for r in range(len(indptr) - 1):
            cols = list(indices[indptr[r]:indptr[r + 1]]); vals = list(data[indptr[r]:indptr[r + 1]])
            if cols and rng.random() < frac:
                for _ in range(int(rng.integers(1, 4))):           # 1..3 extra entries: copies of columns already in the row ...
                    j = int(rng.integers(0, len(cols)))
                    at = int(rng.integers(0, len(cols) + 1))       # ... anywhere in the row: before or after the original, adjacent or not
                    cols.insert(at, cols[j]); vals.insert(at, float(rng.standard_normal()))
            idx += cols; dat += vals
            ptr.append(len(idx))
        return np.asarray(ptr, np.int32), np.asarray(idx, np.int32), np.asarray(dat, np.float64)

    G_row, G_col, G_val = with_duplicates(s.G_row, s.G_col, s.G_val, 0.3)
    C_row, C_col, C_val = with_duplicates(s.C_row, s.C_col, s.C_val, 0.3)
    assert len(G_val) > len(s.G_val) and len(C_val) > len(s.C_val)
    s2 = synth.KKTSystem(S, C, K, G_row, G_col, G_val, C_row, C_col, C_val, s.g, s.c, s.rho)
    Gd_o, Cd_o = co.convert(*s2.csr_args()[:6], S, C, K, s2.rho, dt)
    Gd_n, Cd_n = o.convert(*s2.csr_args()[:6], S, C, K, s2.rho, dt)              # both restatements walk a row in order
    assert np.array_equal(Gd_o, Gd_n) and np.array_equal(Cd_o, Cd_n)
    sol = make_solver(S, C, K, dt)
    dev = sol.upload_system(s2)
    for rep in range(3):                                           # the same answer every time: no race decides the winner
        Gd, Cd = sol.convert(*dev[:6], s2.rho)
        assert np.array_equal(host(Gd), Gd_o) and np.array_equal(host(Cd), Cd_o), rep
    for mode in (1, 2):
        sol.set_option("asm_mode", mode)
        for rep in range(2):
            sol.linsys(*dev, 1e-8, 5, s2.rho)
            assert np.array_equal(sol.read_buffer("G_dense"), Gd_o) and np.array_equal(sol.read_buffer("C_dense"), Cd_o), (mode, rep)
    sol.close()
    # a duplicate-free system afterwards on a fresh solver: the fast path (no owner resolution) still gives the oracle's blocks
    Gd_c, Cd_c = co.convert(*s.csr_args()[:6], S, C, K, s.rho, dt)
    sol = make_solver(S, C, K, dt)
    Gd, Cd = sol.convert(*sol.upload_system(s)[:6], s.rho)
    assert np.array_equal(host(Gd), Gd_c) and np.array_equal(host(Cd), Cd_c)
    sol.close()


def test_thirteen_workgroups_of_64_threads_1000_launches():
    """The geometry of the one hand-off time-out ever seen on the GPU box (round 1, an uncommitted layout that packed the
    granules of several workgroups into one 128-B line): 13 workgroups x 64 threads at 14/7/50 f32.  The committed
    layout gives every workgroup its own lines (static_assert in pcg_resident_kernel + the checks below); 1000 launches,
    every result bitwise equal to the first, no time-out.  One run - not a loop hunting for a fault."""
    from gato_python_amd.csrc_layout import slot_granules
    for S_, esz in ((2, 4), (14, 4), (14, 8), (32, 4), (32, 8), (12, 8)):
        g = slot_granules(S_, esz)
        assert g % 16 == 0 and g >= 16 + 2 * S_ * (esz // 4)          # whole 128-B lines, partial line + both halo blocks
    S, C, K, dt = 14, 7, 50, np.float32
    s = system(S, C, K, seed=2)
    Gd, Cd = co.convert(*s.csr_args()[:6], S, C, K, s.rho, dt)
    Sb, Pb, gam, _ = co.form_schur(Gd, Cd, s.g, s.c, S, C, K)
    Pb = co.form_ss(Sb, Pb, S, K)
    lam_o, _ = co.pcg(Sb, Pb, gam, S, K, 0.0, 20)
    sol = make_solver(S, C, K, dt)
    sol.set_option("pcg_threads", 64)
    sol.set_option("pcg_groups", 13)
    dS, dP, dg = sol.to_device(Sb), sol.to_device(Pb), sol.to_device(gam)
    lam = sol.new(S * K)
    it = sol.new(1, torch.int32)
    first = None
    for i in range(1000):
        sol.pcg(dS, dP, dg, 0.0, 20, lam=lam, iters=it, check=False)
        if i % 250 == 0 or i == 999:
            got = host(lam)
            first = got if first is None else first
            assert np.array_equal(got, first) and int(host(it)[0]) == 20
    sol.check_status()
    assert sol.get_option("last_groups") == 13 and sol.get_option("last_threads") == 64
    check_f32("13 x 64 threads, 20 iterations", first, lam_o,
              co.pcg(Sb.astype(np.float64), Pb.astype(np.float64), gam.astype(np.float64), S, K, 0.0, 20)[0])
    sol.close()


def test_three_concurrent_multi_workgroup_solves_on_three_streams():
    """A12: three 114-workgroup persistent launches of one process on three streams cannot all be co-resident (342
    workgroups, 256 CUs).  The per-device CU budget makes the third wait for the first two; three correct answers."""
    S, C, K, dt = 14, 7, 4096, np.float32
    s = system(S, C, K, seed=4)
    Gd, Cd = co.convert(*s.csr_args()[:6], S, C, K, s.rho, dt)
    Sb, Pb, gam, _ = co.form_schur(Gd, Cd, s.g, s.c, S, C, K)
    Pb = co.form_ss(Sb, Pb, S, K)
    lam_o, it_o = co.pcg(Sb, Pb, gam, S, K, 1e-4, 60)
    sols = [make_solver(S, C, K, dt) for _ in range(3)]
    dS, dP, dg = sols[0].to_device(Sb), sols[0].to_device(Pb), sols[0].to_device(gam)
    from gato_python_amd.dist import lockstep_streams
    streams = lockstep_streams(3)
    lams = [sols[0].new(S * K) for _ in range(3)]
    its = [sols[0].new(1, torch.int32) for _ in range(3)]
    torch.cuda.synchronize()
    for rep in range(10):
        for i in range(3):
            with torch.cuda.stream(streams[i]):
                sols[i].pcg(dS, dP, dg, 1e-4, 60, lam=lams[i], iters=its[i], check=False)
    torch.cuda.synchronize()
    for i in range(3):
        sols[i].check_status()
        assert sols[i].get_option("last_groups") > 100
        check_pcg(f"three concurrent launches, stream {i}", Sb, Pb, gam, S, K, 1e-4, 60, host(lams[i]), int(host(its[i])[0]))
        sols[i].close()


def test_handoff_timeout_is_reported_in_band_and_recovered():
    """A workgroup of a persistent launch that never shows up (test hook of the diagnostic build: the last workgroup
    returns at once, as if it had not been scheduled) makes the others give up after the time-out: iters = -1 in-band,
    gato_pcg_status reports it once, and gato_solver_recover re-runs the solve through the streaming kernels."""
    S, C, K, dt = 14, 7, 512, np.float64
    s = system(S, C, K, seed=6)
    lam_o, dz_o, it_o = co.linsys_solve(*s.csr_args(), S, C, K, 1e-9, 150, s.rho, dtype=dt)
    sol = make_solver(S, C, K, dt)
    dev = sol.upload_system(s)
    lam, dz = sol.new(S * K), sol.new(sol.N)
    sol.set_option("timeout_ms", 20)
    sol.set_option("stamp_pcg", 1)
    sol.set_option("ablate", 16)
    sol.linsys(*dev, 1e-9, 150, s.rho, lam, dz)
    assert np.frombuffer(_read_iters(sol), np.int32)[0] == -1
    with pytest.raises(_lib.GatoError, match="ETIMEOUT"):
        sol.check_status()
    sol.check_status()                                    # reported once
    sol.linsys(*dev, 1e-9, 150, s.rho, lam, dz)           # times out again ...
    assert sol.recover() is True                          # ... and is re-run through the streaming kernels
    assert sol.get_option("last_fallback") == 1 and sol.get_option("last_mode") == _lib.PCG_STREAMING
    assert np.frombuffer(_read_iters(sol), np.int32)[0] == it_o
    assert rel(host(lam), lam_o) < 1e-9 and rel(host(dz), dz_o) < 1e-9
    sol.set_option("ablate", 0)
    sol.set_option("stamp_pcg", 0)
    sol.linsys(*dev, 1e-9, 150, s.rho, lam, dz)
    assert sol.recover() is False and sol.get_option("last_mode") == _lib.PCG_RESIDENT
    assert np.frombuffer(_read_iters(sol), np.int32)[0] == it_o and rel(host(lam), lam_o) < 1e-9
    sol.close()


@pytest.mark.parametrize("mode", [_lib.PRECON_BLOCK_JACOBI, _lib.PRECON_POINT_JACOBI])
@pytest.mark.parametrize("S,C,K,dt", [(14, 7, 50, np.float64), (14, 7, 600, np.float64), (2, 1, 5, np.float64), (32, 16, 9, np.float32)])
def test_preconditioner_modes(S, C, K, dt, mode):
    """The reference's compile switches BLOCK_J_PRECON / SS_PRECON (gato_defines.h:9-10) as the runtime option precon_mode:
    Pinv and the whole solve against the oracle run in the same mode (same iteration count in fp64)."""
    s = system(S, C, K, seed=8) if (S, C, K) != (2, 1, 5) else synth.pendulum_system()
    f64 = dt == np.float64
    sol = make_solver(S, C, K, dt)
    sol.set_option("precon_mode", mode)
    sol.set_option("time_stages", 1)
    dev = sol.upload_system(s)
    lam, dz = sol.new(S * K), sol.new(sol.N)
    # fixed iteration count: the same iterates as the oracle in that mode
    n_it = 4 if K == 5 else 15
    out = o.linsys_solve(*s.csr_args(), S, C, K, 0.0, n_it, s.rho, dtype=dt, return_all=True, precon_mode=mode)
    sol.linsys(*dev, 0.0, n_it, s.rho, lam, dz)
    sol.check_status()
    assert sol.get_option("last_asm_fused") == 0
    assert rel(sol.read_buffer("Pinv"), out["Pinv"]) < (1e-11 if f64 else 3e-4)
    assert int(np.frombuffer(_read_iters(sol), np.int32)[0]) == n_it
    if f64:
        assert rel(host(lam), out["lam"]) < 1e-9 and rel(host(dz), out["dz"]) < 1e-9
    else:
        # fp32: judged against the fp64 iterates after the same number of iterations, beside the fp32 oracle's error.
        # Block-Jacobi: at n_it.  Point-Jacobi at 32/16/9 is barely preconditioned CG on 288 unknowns and loses its
        # conjugacy in single precision between iterations 10 and 15: the error of the SAME recurrence against fp64 is
        # 4e-6 after 10 iterations, 8e-5 after 12 and 6e-2 (C oracle's summation order) / 8e-3 (numpy oracle's order) after
        # 15 - two CPU orders 8x apart (tests/test_oracle.py::test_fp32_point_jacobi_is_order_chaotic_past_ten_iterations).
        # So the GPU is held to the 2x bar where the iteration is still deterministic to rounding (5 and 10 iterations),
        # and at 15 to the worse of the two CPU orders.
        s64, rho32 = _f32_truth_inputs(s)

        def after(n):
            sol.linsys(*dev, 0.0, n, s.rho, lam, dz)
            sol.check_status()
            t = o.linsys_solve(*s64.csr_args(), S, C, K, 0.0, n, rho32, dtype=np.float64, return_all=True, precon_mode=mode)
            a = o.linsys_solve(*s.csr_args(), S, C, K, 0.0, n, s.rho, dtype=np.float32, return_all=True, precon_mode=mode)
            c = co.pcg(a["S"], a["Pinv"], a["gamma"], S, K, 0.0, n)[0]          # the C oracle's order on the same fp32 matrices
            return host(lam).copy(), a["lam"], c, t["lam"]
        if mode == _lib.PRECON_BLOCK_JACOBI:
            g, a, c, t = after(n_it)
            check_f32(f"precon mode {mode} {S}/{C}/{K} lambda after {n_it}", g, a, t, floor=1e-4)
        else:
            for n in (5, 10):
                g, a, c, t = after(n)
                worse = a if rel(a, t) >= rel(c, t) else c
                check_f32(f"precon mode {mode} {S}/{C}/{K} lambda after {n}", g, worse, t)
            g, a, c, t = after(15)
            worse = a if rel(a, t) >= rel(c, t) else c
            check_f32(f"precon mode {mode} {S}/{C}/{K} lambda after 15 (order-chaotic)", g, worse, t)
    # to tolerance: a weaker preconditioner needs more iterations and reaches the same solution.  Their number depends on
    # rounding when convergence is slow: the two CPU restatements themselves differ by one there (point-Jacobi fp64 14/7/50:
    # numpy order 99, C order 100; 14/7/600: 161 / 160), so point-Jacobi gets max(2, 2 %) and block-Jacobi in fp64 none
    tol, mi = (1e-9, 3000) if f64 else (1e-4, 400)
    out = o.linsys_solve(*s.csr_args(), S, C, K, tol, mi, s.rho, dtype=dt, return_all=True, precon_mode=mode)
    sol.linsys(*dev, tol, mi, s.rho, lam, dz)
    sol.check_status()
    it = int(np.frombuffer(_read_iters(sol), np.int32)[0])
    slack = max(2, out["iters"] // 50) if (mode == _lib.PRECON_POINT_JACOBI or not f64) else 0
    assert abs(it - out["iters"]) <= slack, (it, out["iters"])
    if f64:
        assert rel(host(lam), out["lam"]) < 1e-5
    else:
        s64, rho32 = _f32_truth_inputs(s)
        conv = o.linsys_solve(*s64.csr_args(), S, C, K, 1e-14, 3000, rho32, dtype=np.float64, return_all=True, precon_mode=mode)
        check_f32(f"precon mode {mode} {S}/{C}/{K} lambda at the exit test", host(lam), out["lam"], conv["lam"])
    ms = sol.last_stage_ms()
    assert ms["assembly"] > 0 and ms["pcg"] > 0 and ms["dz"] > 0
    sol.close()


def test_stage_times_as_data():
    """gato_last_stage_ms (the reference prints its Schur and solve times, gato_schur.cuh:907-913,972-982)."""
    S, C, K = 14, 7, 50
    s = system(S, C, K, seed=0)
    sol = make_solver(S, C, K, np.float64)
    dev = sol.upload_system(s)
    lam, dz = sol.new(S * K), sol.new(sol.N)
    with pytest.raises(_lib.GatoError):
        sol.last_stage_ms()
    sol.set_option("time_stages", 1)
    for _ in range(3):
        sol.linsys(*dev, 0.0, 100, s.rho, lam, dz)
    ms = sol.last_stage_ms()
    assert 0.001 < ms["assembly"] < 1.0 and 0.05 < ms["pcg"] < 5.0 and 0.0005 < ms["dz"] < 1.0, ms
    sol.close()


@pytest.mark.parametrize("K,dt", [(50, np.float64), (41, np.float64), (37, np.float64), (50, np.float32), (73, np.float32), (9, np.float32)])
def test_dz_by_the_helper_blocks_is_bit_identical_to_the_dz_launch(K, dt):
    """BASELINE configs[1] (one system, fp64, the mixed-rows kernel) and the fp32 two-rows-per-lane kernel: the launch's helper
    blocks - there to warm the L2 - also do the dz back-substitution as soon as the solving workgroup has published lambda
    (default; no_fuse_dz = 1 keeps the dz launch).  Same formulas and order as dz_kernel: the same bits, run after run."""
    from gato_python_amd.solver import Solver
    S, C = 14, 7
    s = system(S, C, K, seed=7)
    res = {}
    f64 = dt == np.float64
    tol = 1e-9 if f64 else 1e-5
    for nofuse in (0, 1):
        sol = Solver(S, C, K, dt)
        sol.set_option("no_fuse_dz", nofuse)
        dev = sol.upload_system(s)
        lam, dz = sol.new(S * K), sol.new(sol.N)
        runs = []
        for rep in range(6):
            dz.fill_(float("nan"))
            sol.linsys(*dev, tol, 80, s.rho, lam, dz)
            sol.check_status()
            runs.append((host(lam).copy(), host(dz).copy()))
        assert sol.get_option("last_dz_fused") == (0 if nofuse else 2) and sol.get_option("last_pair") == (2 if f64 else 1)
        assert all(np.array_equal(r[0], runs[0][0]) and np.array_equal(r[1], runs[0][1]) for r in runs)
        res[nofuse] = runs[0]
        sol.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    check_solve(f"dz by the helper blocks 14/7/{K} {np.dtype(dt).name}", s, S, C, K, dt, tol, 80, res[0][0], res[0][1], f64_tol=1e-9)


@pytest.mark.parametrize("K,dt", [(50, np.float64), (37, np.float64), (50, np.float32), (73, np.float32), (9, np.float32), (2, np.float32), (1, np.float32)])
def test_transposed_images_give_the_bits_of_the_block_rows(K, dt):
    """Whole solves of one system that a one-workgroup two-rows-per-lane kernel serves: the fused assembly launch also writes S
    and Pinv transposed (column c of all rows contiguous) and the PCG launch loads its rows from there with unit stride
    (default); no_image = 1 loads from the block rows as every other kernel does.  Same lambda and dz bit for bit, also through
    the block-input entry; a stage-level gato_pcg call on the caller's own arrays and a solve in another preconditioner mode
    never see the images."""
    from gato_python_amd.solver import Solver
    S, C = 14, 7
    s = system(S, C, K, seed=21) if K > 1 else synth.blocks_to_csr(*synth.make_blocks(S, C, 1, 21, False))
    f64 = dt == np.float64
    tol = 1e-9 if f64 else 1e-5
    res = {}
    for noimg in (0, 1):
        sol = Solver(S, C, K, dt)
        sol.set_option("no_image", noimg)
        dev = sol.upload_system(s)
        lam, dz = sol.new(S * K), sol.new(sol.N)
        for rep in range(3):
            dz.fill_(float("nan"))
            sol.linsys(*dev, tol, 80, s.rho, lam, dz)
            sol.check_status()
        assert sol.get_option("last_image") == (0 if noimg else 1) and sol.get_option("last_pair") == (2 if f64 else 1)
        res[noimg] = (host(lam).copy(), host(dz).copy())
        if not noimg:
            # stage-level PCG on the solver's own S / Pinv / gamma (what the whole solve left there): block rows, not images
            l2, _ = sol.pcg(sol.buffer_ptr(3), sol.buffer_ptr(4), sol.buffer_ptr(5), tol, 80)
            assert sol.get_option("last_image") == 0
            assert np.array_equal(host(l2), res[0][0])
            if K > 2:
                sol.set_option("precon_mode", 1)                  # block-Jacobi: stage kernels, no images
                sol.linsys(*dev, tol, 200, s.rho, lam, dz)
                sol.check_status()
                assert sol.get_option("last_image") == 0
                sol.set_option("precon_mode", 0)
                sol.linsys(*dev, tol, 80, s.rho, lam, dz)         # and back: images rewritten by this solve's assembly
                sol.check_status()
                assert sol.get_option("last_image") == 1
                assert np.array_equal(host(lam), res[0][0]) and np.array_equal(host(dz), res[0][1])
        sol.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    check_solve(f"transposed images 14/7/{K} {np.dtype(dt).name}", s, S, C, K, dt, tol, 80, res[0][0], res[0][1], f64_tol=1e-9, two_orders=K <= 2)


@pytest.mark.parametrize("K,B,warm", [(50, 1, 0), (73, 1, 0), (49, 1, 1), (19, 1, 0), (10, 1, 1), (9, 1, 0), (2, 1, 0), (1, 1, 0), (50, 5, 0), (23, 7, 0)])
def test_private_windows_kernel_over_wave_boundaries_batches_and_warm_start(K, B, warm):
    """The fp32 two-rows-per-lane kernel (the reference's precision at its own shape: what the drop-in runs at 14/7/50) keeps
    WAVE-PRIVATE operand windows: every wave advances the halo rows of r and p itself with the owner's FMA, so an iteration has
    the two barriers of its block sums only (the shared-window form it was checked against bit for bit in rounds 3-4 is gone).
    Wave boundaries inside a knot (K = 19, 73), one- and two-knot systems, true warm start, batches: 12 fixed iterations against
    the oracle's iterates, the tolerance exit against the oracle's solve, the same bits run after run, and - the halo rows are
    where this kernel could go wrong - the general one-row-per-lane kernel on the same system beside it."""
    from gato_python_amd.solver import Solver
    S, C, dt = 14, 7, np.float32
    systems = [system(S, C, K, seed=11 + b) if K > 1 else synth.blocks_to_csr(*synth.make_blocks(S, C, 1, 11 + b, False)) for b in range(B)]
    lam0 = np.random.default_rng(K).standard_normal(B * S * K).astype(dt)
    res = {}
    for no_pair in (0, 1):
        if no_pair and B > 1 and K * S > 768:
            continue                                    # (a batch needs one workgroup per system: the one-row kernel serves K S <= 768)
        sol = Solver(S, C, K, dt, batch=B)
        sol.set_option("no_pair", no_pair)
        if warm:
            sol.set_option("true_warm_start", 1)
        sol.set_option("record_eta", 1)
        lam, dz, it = sol.new(B * S * K), sol.new(B * sol.N), sol.new(B, torch.int32)
        dev = sol.upload_batch(systems) if B > 1 else sol.upload_system(systems[0])
        runs = []
        for (tol, mi) in ((0.0, 12 if K > 2 else K), (1e-5, 80), (1e-5, 80)):     # (a one-knot system is solved exactly by its first step: eta = 0 afterwards)
            if warm:
                lam.copy_(torch.from_numpy(lam0))
            if B > 1:
                sol.linsys_batched(*dev, tol, mi, systems[0].rho, lam, dz, it)
            else:
                sol.linsys(*dev, tol, mi, systems[0].rho, lam, dz)
            sol.check_status()
            runs.append((host(lam).copy(), host(dz).copy(), host(it).copy() if B > 1 else sol.eta_history(min(mi, 12))))
        if not no_pair:
            assert sol.get_option("last_pair") == 1 and sol.get_option("last_groups") == 1
        res[no_pair] = runs
        sol.close()
    fixed, tol1, tol2 = res[0]
    assert np.isfinite(fixed[0]).all() and np.isfinite(fixed[1]).all()
    assert np.array_equal(tol1[0], tol2[0]) and np.array_equal(tol1[1], tol2[1]) and np.array_equal(tol1[2], tol2[2])     # deterministic
    n_dz = (S + C) * K - C
    if 1 in res and K > 2:          # the one-row kernel on the same inputs: another summation grouping of the same recurrence
        assert rel(fixed[0], res[1][0][0]) < 2e-4 and rel(tol1[0], res[1][1][0]) < 2e-3, (rel(fixed[0], res[1][0][0]), rel(tol1[0], res[1][1][0]))
    if K > 2 and not warm:          # (a warm start from a random lambda0 has no oracle run to stand beside: the one-row kernel above is its check)
        check_solve(f"private windows 14/7/{K} x{B} float32", systems[0], S, C, K, dt, 1e-5, 80, tol1[0][:S * K], tol1[1][:n_dz])
        if True:
            s0 = systems[0]
            lam_o, dz_o, _ = co.linsys_solve(*s0.csr_args(), S, C, K, 0.0, 12, s0.rho, dtype=dt)
            s64, rho32 = _f32_truth_inputs(s0)
            lam_t, dz_t, _ = co.linsys_solve(*s64.csr_args(), S, C, K, 0.0, 12, rho32, dtype=np.float64)
            check_f32(f"private windows, lambda after 12 iterations 14/7/{K} x{B}", fixed[0][:S * K], lam_o, lam_t)
            check_f32(f"private windows, dz after 12 iterations 14/7/{K} x{B}", fixed[1][:n_dz], dz_o, dz_t)


@pytest.mark.parametrize("S,C,K,dt,B", [(14, 7, 50, np.float64, 1), (14, 7, 37, np.float64, 1), (14, 7, 50, np.float64, 6), (2, 1, 5, np.float64, 1),
                                        (32, 16, 7, np.float32, 1), (14, 7, 1, np.float64, 1), (14, 7, 2, np.float32, 3)])
def test_dz_in_the_pcg_epilogue_is_bit_identical_to_the_dz_launch(S, C, K, dt, B):
    """One-workgroup PCG launches (and one workgroup per system of a batch) also do the dz back-substitution (compute_dz,
    gato_schur.cuh:758-867): same formulas and order as dz_kernel, so the results are the same bits."""
    from gato_python_amd.solver import Solver
    s = system(S, C, K, seed=5) if K > 1 else synth.blocks_to_csr(*synth.make_blocks(S, C, 1, 5, False))
    res = {}
    tol = 1e-9 if dt == np.float64 else 1e-5          # fp32: stop at convergence (60 iterations on 28 unknowns would run on rounding noise)
    for nofuse in (0, 1):
        sol = Solver(S, C, K, dt, batch=B)
        sol.set_option("no_fuse_dz", 1 if nofuse else -1)    # -1: also for one system (default: batches only)
        sol.set_option("no_pair", 1)                      # the two-rows-per-lane fp32 kernel has no dz epilogue
        if B > 1:
            dev = sol.upload_batch([s] * B)
            lam, dz = sol.new(B * S * K), sol.new(B * sol.N)
            sol.linsys_batched(*dev, tol, 60, s.rho, lam, dz)
        else:
            dev = sol.upload_system(s)
            lam, dz = sol.new(S * K), sol.new(sol.N)
            sol.linsys(*dev, tol, 60, s.rho, lam, dz)
        sol.check_status()
        assert sol.get_option("last_dz_fused") == (0 if nofuse else 1)
        res[nofuse] = (host(lam).copy(), host(dz).copy())
        sol.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    n_dz = (S + C) * K - C
    # (14/7/2 in fp32: a 28-unknown system whose last iterations run on rounding noise - order-chaotic, tools/past_convergence.py)
    check_solve(f"dz in the PCG epilogue {S}/{C}/{K} x{B}", s, S, C, K, dt, tol, 60, res[0][0][:S * K], res[0][1][:n_dz], f64_tol=1e-9,
                two_orders=K <= 2)


@pytest.mark.parametrize("K,B", [(50, 6), (37, 3), (73, 2), (3, 4), (1, 2)])
def test_dz_in_the_fp32_two_row_epilogue_of_a_batch_is_bit_identical_to_the_dz_launch(K, B):
    """Batches through pcg_single_f32x2_kernel (one workgroup per system, two rows per lane, wave-private windows): the
    dz back-substitution rides in the launch's epilogue as in the fp64 kernel - the bits of dz_kernel, no dz launch."""
    from gato_python_amd.solver import Solver
    S, C, dt = 14, 7, np.float32
    systems = [system(S, C, K, seed=20 + i) if K > 1 else synth.blocks_to_csr(*synth.make_blocks(S, C, 1, 20 + i, False)) for i in range(B)]
    res = {}
    for nofuse in (0, 1):
        sol = Solver(S, C, K, dt, batch=B)
        sol.set_option("no_fuse_dz", nofuse)
        dev = sol.upload_batch(systems)
        lam, dz = sol.new(B * S * K), sol.new(B * sol.N)
        dz.fill_(float("nan"))
        sol.linsys_batched(*dev, 1e-5, 60, systems[0].rho, lam, dz)
        sol.check_status()
        assert sol.get_option("last_pair") == 1 and sol.get_option("last_dz_fused") == (0 if nofuse else 1)
        res[nofuse] = (host(lam).copy(), host(dz).copy())
        sol.close()
    assert np.isfinite(res[0][1]).all()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    if K > 2:
        n_dz = (S + C) * K - C
        # K <= 4 (42 - 56 unknowns): exit_tol = 1e-5 stops these solves in the middle of the steep final descent (the error
        # falls 5x per iteration there and the three summation orders - GPU, C oracle, numpy oracle - swap places at every
        # iteration: profiles/r05_f32_pair_cause.log; round 4's worst pair, 1.55, was system 0 of K = 3 caught at such a
        # point with EQUAL exit iterations).  That snapshot is judged against both CPU orders, and the solve is judged again
        # four iterations later, where all three have reached the floor the arithmetic allows.
        tiny = K <= 4
        for b in (0, B - 1):
            it_o = check_solve(f"fp32 two-row epilogue dz 14/7/{K} system {b} of {B}", systems[b], S, C, K, dt, 1e-5, 60,
                               res[0][0][b * S * K:(b + 1) * S * K], res[0][1][b * n_dz:(b + 1) * n_dz], two_orders=tiny)
            if tiny:
                n = min(it_o + 5, 60)
                sol = Solver(S, C, K, dt, batch=B)
                dev = sol.upload_batch(systems)
                lam, dz = sol.new(B * S * K), sol.new(B * sol.N)
                sol.linsys_batched(*dev, 0.0, n, systems[0].rho, lam, dz)
                sol.check_status()
                s64, rho32 = _f32_truth_inputs(systems[b])
                lam_of = co.linsys_solve(*systems[b].csr_args(), S, C, K, 0.0, n, systems[b].rho, dtype=dt)[0]
                lam_tf = co.linsys_solve(*s64.csr_args(), S, C, K, 0.0, n, rho32, dtype=np.float64)[0]
                check_f32(f"solve lambda after {n} fixed iterations (past the descent) fp32 two-row epilogue 14/7/{K} system {b} of {B}",
                          host(lam)[b * S * K:(b + 1) * S * K], lam_of, lam_tf)
                sol.close()


def test_mixed_rows_kernel_against_the_general_launch():
    """pcg_single_f64m_kernel (BASELINE configs[1]: four two-row waves + four waves of 16-lane DPP rows; round 2's dense layout it
    was A/B-ed against for two rounds is gone): every K it serves stops at the oracle's iteration with the oracle's solution, and
    the general launch (option no_pair) on the same matrices agrees to rounding (another grouping of the dot products)."""
    S, C = 14, 7
    for K in (37, 42, 49, 50):
        s = synth.make_system(S, C, K, seed=300 + K)
        Gd, Cd = co.convert(*s.csr_args()[:6], S, C, K, s.rho, np.float64)
        Sb, Pb, gam, _ = co.form_schur(Gd, Cd, s.g, s.c, S, C, K)
        Pb = co.form_ss(Sb, Pb, S, K)
        lam_o, it_o = co.pcg(Sb, Pb, gam, S, K, 1e-10, 300)
        res = {}
        for no_pair in (0, 1):
            sol = make_solver(S, C, K, np.float64)
            sol.set_option("no_pair", no_pair)
            lam, it = sol.pcg(sol.to_device(Sb), sol.to_device(Pb), sol.to_device(gam), 1e-10, 300)
            assert sol.get_option("last_pair") == (0 if no_pair else 2) and int(host(it)[0]) == it_o, (K, no_pair, int(host(it)[0]), it_o)
            assert rel(host(lam), lam_o) < 1e-10, (K, no_pair)
            res[no_pair] = host(lam).copy()
            sol.close()
        assert rel(res[0], res[1]) < 1e-11, K


def test_captured_whole_solve_replays_with_new_inputs():
    """a whole solve captured into a graph and REPLAYED after the inputs changed in place.  A replay repeats the
    launch arguments of the capture, so while a stream is being captured the helper blocks of the one-workgroup kernels do not
    do dz (their flag value would be stale on replay) and a multi-workgroup persistent launch (hand-off epochs) is refused;
    lambda and dz of every replay must be the oracle's for the inputs of THAT replay."""
    from gato_python_amd.solver import Solver
    S, C, K = 14, 7, 50
    for dt, tol8 in ((np.float64, 1e-9), (np.float32, None)):
        s1, s2 = system(S, C, K, seed=11), system(S, C, K, seed=12)
        sol = Solver(S, C, K, dt)
        dev = list(sol.upload_system(s1))
        lam, dz = sol.new(S * K), sol.new(sol.N)
        st = torch.cuda.Stream()
        tol, mi = (1e-9, 100) if dt == np.float64 else (1e-5, 100)
        with torch.cuda.stream(st):
            sol.linsys(*dev, tol, mi, s1.rho, lam, dz)
            st.synchronize()
            assert sol.get_option("last_dz_fused") == 2                  # the plain stream: helper blocks do dz
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                sol.linsys(*dev, tol, mi, s1.rho, lam, dz)
            assert sol.get_option("last_dz_fused") == 0                  # captured: the dz launch of its own
            for rep, sysm in enumerate((s1, s2, s1, s2, s2)):
                new = sol.upload_system(sysm)
                for d, n in zip(dev[2:], new[2:]):                       # values only: the sparsity pattern is the capture's
                    d.copy_(n)
                lam.fill_(float("nan")); dz.fill_(float("nan"))
                g.replay()
                st.synchronize()
                sol.check_status()
                check_solve(f"graph replay {rep} 14/7/{K} {np.dtype(dt).name}", sysm, S, C, K, dt, tol, mi, host(lam), host(dz), f64_tol=1e-9)
        sol.close()
    # a launch that needs hand-off epochs cannot be captured: refused with an error, nothing enqueued, capture still valid
    S, C, K = 14, 7, 512
    s = system(S, C, K, seed=3)
    sol = Solver(S, C, K, np.float32)
    dev = sol.upload_system(s)
    lam, dz = sol.new(S * K), sol.new(sol.N)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        sol.linsys(*dev, 1e-5, 100, s.rho, lam, dz)
        st.synchronize()
        ref = host(lam).copy()
        g = torch.cuda.CUDAGraph()
        with pytest.raises(_lib.GatoError):
            with torch.cuda.graph(g, stream=st):
                sol.linsys(*dev, 1e-5, 100, s.rho, lam, dz)
    torch.cuda.synchronize()
    sol.set_option("pcg_mode", 2)                                        # the streaming kernels replay correctly
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        sol.linsys(*dev, 1e-5, 100, s.rho, lam, dz)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            sol.linsys(*dev, 1e-5, 100, s.rho, lam, dz)
        for _ in range(3):
            lam.fill_(float("nan"))
            g.replay()
            st.synchronize()
            assert rel(host(lam), ref) < 1e-4
    sol.close()


@pytest.mark.parametrize("K", [512, 4096])
def test_multi_workgroup_solve_beside_a_long_foreign_kernel(K):
    """VERDICT r3 #4 (A12: check_sms + cudaLaunchCooperativeKernel, gato_utils.cuh:829-854, gato_pcg.cuh:502-526): a persistent
    multi-workgroup solve enqueued while a long kernel of ANOTHER library (a torch matmul on another stream, invisible to this
    library's own co-residency gate) occupies the chip.  Both launch paths - the plain launch (default) and
    hipLaunchCooperativeKernel (option coop_launch) - must come back complete (no hand-off time-out) with the bits of the
    solve that ran alone."""
    from gato_python_amd.solver import Solver
    S, C = 14, 7
    s = system(S, C, K, seed=21)
    sol = Solver(S, C, K, np.float32)
    dev = sol.upload_system(s)
    lam, dz = sol.new(S * K), sol.new(sol.N)
    sol.linsys(*dev, 1e-5, 100, s.rho, lam, dz)
    torch.cuda.synchronize(); sol.check_status()
    assert sol.get_option("last_groups") > 1
    ref = (host(lam).copy(), host(dz).copy())
    A = torch.randn(12288, 12288, device="cuda", dtype=torch.float32)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    for coop in (0, 1, 0):
        sol.set_option("coop_launch", coop)
        lam.fill_(float("nan")); dz.fill_(float("nan"))
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(4):
                B = A @ A                                    # tens of milliseconds each, every CU busy
        for _ in range(3):
            sol.linsys(*dev, 1e-5, 100, s.rho, lam, dz)      # enqueued while the matmuls run
        torch.cuda.synchronize()
        sol.check_status()                                   # raises on a hand-off time-out
        assert np.array_equal(host(lam), ref[0]) and np.array_equal(host(dz), ref[1]), coop
    del B
    sol.close()




def test_largest_batch_one_grid_row_per_system():
    """65 535 systems in one call - the most a batch may hold (one grid row per system; more is refused when the solver is created,
    tests/test_capi_cpu.py): IIWA 14/7/50 fp32, a shared matrix with the right-hand side of system b scaled by 1 + b / B.  The first,
    middle and last systems against the oracle; every system against the first one scaled (the solve is linear in g and c)."""
    from gato_python_amd.solver import Solver
    S, C, K, B, dt = 14, 7, 50, 65535, np.float32
    s = synth.make_system(S, C, K, seed=3)
    sol = Solver(S, C, K, dt, batch=B)
    i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.int32)).cuda()
    scale = (1.0 + np.arange(B) / B).astype(np.float32)
    Gv = torch.from_numpy(np.tile(s.G_val.astype(dt), B)).cuda()
    Cv = torch.from_numpy(np.tile(s.C_val.astype(dt), B)).cuda()
    g = torch.from_numpy((scale[:, None] * s.g.astype(dt)[None, :]).reshape(-1)).cuda()
    c = torch.from_numpy((scale[:, None] * s.c.astype(dt)[None, :]).reshape(-1)).cuda()
    lam = torch.full((S * K * B,), float("nan"), dtype=torch.float32, device="cuda")
    dz = torch.full((sol.N * B,), float("nan"), dtype=torch.float32, device="cuda")
    its = torch.zeros(B, dtype=torch.int32, device="cuda")
    sol.linsys_batched(i32(s.G_row), i32(s.G_col), Gv, i32(s.C_row), i32(s.C_col), Cv, g, c, 0.0, 25, s.rho, lam, dz, its)
    torch.cuda.synchronize()
    sol.check_status()
    lam_h, dz_h = lam.cpu().numpy().reshape(B, -1), dz.cpu().numpy().reshape(B, -1)
    assert np.isfinite(lam_h).all() and np.isfinite(dz_h).all()
    assert int(its.min()) == 25 and int(its.max()) == 25
    for b in (0, B // 2, B - 1):
        lo, dzo, _ = co.linsys_solve(s.G_row, s.G_col, s.G_val, s.C_row, s.C_col, s.C_val, scale[b] * s.g.astype(dt), scale[b] * s.c.astype(dt),
                                     S, C, K, 0.0, 25, s.rho, dtype=dt)
        assert rel(lam_h[b], lo) < 1e-4 and rel(dz_h[b], dzo) < 1e-4, b
    assert np.abs(lam_h / scale[:, None] - lam_h[0][None, :]).max() / np.abs(lam_h[0]).max() < 1e-4
    assert np.abs(dz_h / scale[:, None] - dz_h[0][None, :]).max() / np.abs(dz_h[0]).max() < 1e-4
    sol.close()


def test_solvers_driven_from_four_host_threads():
    """Four host threads, each with a solver and a stream of its own, 150 whole solves each (multi-workgroup persistent launches:
    four of them do not fit the chip together, so the co-residency gate - process-wide state under a mutex - makes launches wait for
    each other), plus a fifth thread calling the reference surface (its cached solver, another mutex).  Every result equal to the
    thread's first one bit for bit and to the oracle; no time-out, no dead-lock."""
    import threading
    from gato_python_amd import linsys as host_entry
    S, C, dt = 14, 7, np.float32
    Ks = [4096, 3000, 4096, 700]
    sysms = [system(S, C, K, seed=60 + i) for i, K in enumerate(Ks)]
    answers = [co.linsys_solve(*s.csr_args(), S, C, s.K, 0.0, 25, s.rho, dtype=dt) for s in sysms]
    errors = []

    def worker(i):
        try:
            torch.cuda.set_device(0)
            s, K = sysms[i], Ks[i]
            sol = make_solver(S, C, K, dt)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                dev = sol.upload_system(s)
                lam, dz = sol.new(S * K), sol.new(sol.N)
                first = None
                for n in range(150):
                    sol.linsys(*dev, 0.0, 25, s.rho, lam, dz)
                    if n % 50 == 0 or n == 149:
                        st.synchronize()
                        got = (host(lam).copy(), host(dz).copy())
                        first = got if first is None else first
                        assert np.array_equal(got[0], first[0]) and np.array_equal(got[1], first[1]), (i, n)
                st.synchronize()
                sol.check_status()
            assert rel(first[0], answers[i][0]) < 2e-3 and rel(first[1], answers[i][1]) < 2e-3, i
            sol.close()
        except Exception as e:                                   # noqa: BLE001
            errors.append((i, repr(e)[:300]))

    def surface():
        try:
            P = synth.PENDULUM
            want = None
            for n in range(150):
                lam, dz = host_entry.linsys_solve(P["G_row"], P["G_col"], P["G_val"], P["C_row"], P["C_col"], P["C_val"], P["g_val"], P["c_val"],
                                                  P["input_lambda"], 1, 1e-6, 10, False, 1e-3)
                want = lam if want is None else want
                assert lam == want
        except Exception as e:                                   # noqa: BLE001
            errors.append(("surface", repr(e)[:300]))

    os.environ["GATO_VERBOSE"] = "0"
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)] + [threading.Thread(target=surface)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a thread is stuck"
    assert not errors, errors
