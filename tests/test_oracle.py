"""CPU suite: the oracle (numpy + C restatements) against the golden vectors and the dense KKT solve."""
import json
import os

import numpy as np
import pytest

from gato_python_amd import synth
from oracle import c_oracle as co
from oracle import gato_oracle as o


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


@pytest.fixture(scope="module")
def pend(golden_dir):
    with open(os.path.join(golden_dir, "pendulum.json")) as f:
        return json.load(f)


def test_pendulum_inputs_are_the_reference_literals(pend):
    # test_pendulum_5.py:9-24
    i = pend["inputs"]
    assert i["G_row"] == list(range(15)) and i["G_col"] == list(range(14))
    assert i["G_val"] == [1., 1., 0.1] * 4 + [100., 100.]
    assert i["C_row"] == [0, 1, 2, 5, 9, 12, 16, 19, 23, 26, 30]
    assert len(i["C_col"]) == 30 and len(i["C_val"]) == 30
    assert i["g_val"][12] == -314.159 and i["rho"] == .001 and i["max_iters"] == 10
    assert synth.PENDULUM["C_val"] == i["C_val"] and synth.PENDULUM["C_col"] == i["C_col"]


@pytest.mark.parametrize("impl", ["numpy", "c"])
def test_pendulum_every_intermediate(pend, impl):
    p = synth.pendulum_system()
    e = pend["expected"]
    m = o if impl == "numpy" else co
    Gd, Cd = m.convert(*p.csr_args()[:6], p.S, p.C, p.K, p.rho, np.float64)
    assert np.array_equal(Gd, np.asarray(e["G_dense"])) and np.array_equal(Cd, np.asarray(e["C_dense"]))
    Sb, Pb, gam, Gi = m.form_schur(Gd, Cd, p.g, p.c, p.S, p.C, p.K)
    Pb = m.form_ss(Sb, Pb, p.S, p.K)
    np.testing.assert_allclose(Sb, e["S"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(Pb, e["Pinv"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(gam, e["gamma"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(Gi, e["Ginv"], rtol=1e-13, atol=1e-15)
    lam, iters, hist = m.pcg(Sb, Pb, gam, p.S, p.K, 1e-6, 10, return_history=True)
    assert iters == e["iters_f64"] == 4
    np.testing.assert_allclose(np.asarray(hist)[:5], e["eta"][:5], rtol=1e-10)
    dz = m.compute_dz(Gi, Cd, p.g, lam, p.S, p.C, p.K)
    np.testing.assert_allclose(lam, e["lam"], rtol=1e-11)
    np.testing.assert_allclose(dz, e["dz"], rtol=1e-9, atol=1e-10)


def test_pendulum_matches_dense_kkt_with_rho(pend):
    """The reference test's own oracle (test_pendulum_5.py:28-37), with rho as the solver adds it (D6)."""
    e = pend["expected"]
    p = synth.pendulum_system()
    dz, lam = synth.dense_kkt_solve(p, with_rho=True)
    np.testing.assert_allclose(lam, e["dense_kkt_lam"], rtol=1e-12)
    assert np.abs(np.asarray(e["lam"]) - lam).max() < 1e-9
    assert np.abs(np.asarray(e["dz"]) - dz).max() < 1e-9
    # the reference's assertion, verbatim tolerance, against the rho-free system it builds
    x = np.concatenate([e["dense_kkt_norho_dz"], e["dense_kkt_norho_lam"]])
    x_gato = np.concatenate([e["dz"], e["lam"]])
    assert np.allclose(x_gato, x, rtol=1, atol=0.01)
    # known answers quoted in SURVEY.md section 8c
    assert abs(e["lam"][0] - (-203.147040572)) < 1e-8 and abs(e["dz"][2] - (-32.17244716)) < 1e-7


def test_pendulum_fp32_iteration_count(pend):
    p = synth.pendulum_system()
    _, _, it = o.linsys_solve(*p.csr_args(), p.S, p.C, p.K, 1e-6, 10, p.rho, dtype=np.float32)
    assert it == pend["expected"]["iters_f32"] == 5


@pytest.mark.parametrize("name,S,C,K,seed,dq", [("iiwa_14_7_50_seed0.npz", 14, 7, 50, 0, False),
                                                ("s32_c16_k12_seed5_denseq.npz", 32, 16, 12, 5, True)])
def test_synthetic_golden(golden_dir, name, S, C, K, seed, dq):
    gold = np.load(os.path.join(golden_dir, name))
    s = synth.make_system(S, C, K, seed=seed, dense_q=dq)
    tol, mi = (1e-6, 100) if S == 14 else (1e-12, 500)
    for m in (o, co):
        lam, dz, it = m.linsys_solve(*s.csr_args(), S, C, K, tol, mi, s.rho, dtype=np.float64)
        assert it == int(gold["iters"])
        assert rel(lam, gold["lam"]) < 1e-10 and rel(dz, gold["dz"]) < 1e-10
    if "lam_tight" in gold:
        assert rel(gold["lam_tight"], gold["dense_lam"]) < 1e-7
        assert np.abs(gold["dz_tight"] - gold["dense_dz"]).max() < 1e-6     # BASELINE: ||dz - dz_ref||inf < 1e-6
    else:
        assert np.abs(gold["dz"] - gold["dense_dz"]).max() < 1e-6


@pytest.mark.parametrize("S,C,K,dq", [(2, 1, 2, False), (2, 1, 1, False), (14, 7, 3, True), (32, 16, 5, False),
                                      (2, 1, 33, True)])
def test_restatement_equals_dense_kkt(S, C, K, dq):
    s = synth.make_system(S, C, K, seed=3, dense_q=dq) if K > 1 else synth.blocks_to_csr(
        *synth.make_blocks(S, C, 1, 3, dq))
    lam, dz, it = o.linsys_solve(*s.csr_args(), S, C, K, 1e-20, 2000, s.rho, dtype=np.float64)
    dz_d, lam_d = synth.dense_kkt_solve(s)
    assert rel(lam, lam_d) < 1e-8 and np.abs(dz - dz_d).max() < 1e-7
    lam_c, dz_c, it_c = co.linsys_solve(*s.csr_args(), S, C, K, 1e-20, 2000, s.rho, dtype=np.float64)
    assert rel(lam_c, lam_d) < 1e-8 and np.abs(dz_c - dz_d).max() < 1e-7


def test_c_and_numpy_oracles_agree_fp32():
    s = synth.make_system(14, 7, 20, seed=1)
    a = o.linsys_solve(*s.csr_args(), 14, 7, 20, 1e-6, 100, s.rho, dtype=np.float32, return_all=True)
    Gd, Cd = co.convert(*s.csr_args()[:6], 14, 7, 20, s.rho, np.float32)
    assert np.array_equal(Gd, a["G_dense"]) and np.array_equal(Cd, a["C_dense"])
    Sb, Pb, gam, Gi = co.form_schur(Gd, Cd, s.g, s.c, 14, 7, 20)
    Pb = co.form_ss(Sb, Pb, 14, 20)
    assert rel(Sb, a["S"]) < 1e-5 and rel(Pb, a["Pinv"]) < 1e-5 and rel(gam, a["gamma"]) < 1e-5
    lam, it = co.pcg(Sb, Pb, gam, 14, 20, 1e-6, 100)
    assert abs(it - a["iters"]) <= 1 and rel(lam, a["lam"]) < 1e-3


def test_generator_structure():
    s = synth.make_system(14, 7, 6, seed=0)
    assert len(s.G_row) == s.N + 1 and len(s.C_row) == 14 * 6 + 1 and len(s.g) == s.N and len(s.c) == 84
    from scipy import sparse
    Cm = sparse.csr_matrix((s.C_val, s.C_col, s.C_row), shape=(84, s.N))
    assert Cm.has_sorted_indices
    assert np.allclose(Cm[:14, :14].toarray(), np.eye(14))
    assert np.allclose(Cm[14:28, 21:35].toarray(), np.eye(14))
    assert np.all(s.c[:14] == 0)


@pytest.mark.parametrize("mode", [1, 2])
def test_other_preconditioner_modes_still_solve_the_kkt_system(mode):
    """BLOCK_J_PRECON / SS_PRECON (gato_defines.h:9-10) only change the preconditioner: PCG run to a tight tolerance reaches
    the dense KKT solution with every one of them, in more iterations than the stair preconditioner needs."""
    S, C, K = 14, 7, 12
    s = synth.make_system(S, C, K, seed=9)
    dz_ref, lam_ref = synth.dense_kkt_solve(s)
    out0 = o.linsys_solve(*s.csr_args(), S, C, K, 1e-20, 2000, s.rho, dtype=np.float64, return_all=True)
    out = o.linsys_solve(*s.csr_args(), S, C, K, 1e-20, 2000, s.rho, dtype=np.float64, return_all=True, precon_mode=mode)
    assert rel(out["lam"], lam_ref) < 1e-7 and rel(out["dz"], dz_ref) < 1e-7
    assert out["iters"] >= out0["iters"]
    L, M, R = o.unpack_bd(out["Pinv"], S, K)
    assert not L.any() and not R.any()
    if mode == 2:
        assert np.count_nonzero(M) == S * K


def test_fp32_point_jacobi_is_order_chaotic_past_ten_iterations():
    """VERDICT r2 weak #2 (the point-Jacobi fp32 comparison that was dropped): barely preconditioned CG in single precision
    loses its conjugacy after ~10 iterations at 32/16/9, and from there the error against the fp64 iterates depends on the
    summation order by an order of magnitude - shown here between the two CPU restatements alone (numpy order vs the C
    oracle's loop order, same recurrence, same fp32 matrices), no GPU involved.  Up to 10 iterations both stay at rounding
    level and within 3x of each other: that is where tests/test_gpu_parity.py::test_preconditioner_modes holds the GPU to
    the 2x bar; at 15 it is held to the worse of the two."""
    from oracle import c_oracle as co
    S, C, K = 32, 16, 9
    s = synth.make_system(S, C, K, seed=8)
    a = o.linsys_solve(*s.csr_args(), S, C, K, 0.0, 1, s.rho, dtype=np.float32, return_all=True, precon_mode=2)
    Sb, Pb, g = a["S"], a["Pinv"], a["gamma"]
    S64, P64, g64 = Sb.astype(np.float64), Pb.astype(np.float64), g.astype(np.float64)

    def errs(n):
        t = co.pcg(S64, P64, g64, S, K, 0.0, n)[0]
        den = np.abs(t).max()
        return (np.abs(co.pcg(Sb, Pb, g, S, K, 0.0, n)[0] - t).max() / den, np.abs(o.pcg(Sb, Pb, g, S, K, 0.0, n)[0] - t).max() / den)
    for n in (5, 10):
        ec, en = errs(n)
        assert max(ec, en) < 1e-5 and max(ec, en) / min(ec, en) < 3.0, (n, ec, en)
    ec, en = errs(15)
    assert max(ec, en) > 1e-3 and max(ec, en) / min(ec, en) > 4.0, (ec, en)      # measured: 6.1e-2 (C order) vs 7.7e-3 (numpy order)
