"""The upstream KKT producer (gato_python_amd/kkt.py) against the reference-owned fixture and the oracle."""
import numpy as np
import pytest

from gato_python_amd import kkt, synth
from oracle import gato_oracle as o


def solve(p, tol, mi):
    lam, dz, iters = o.linsys_solve(*p.csr_args(), p.S, p.C, p.K, tol, mi, p.rho, np.float64)[:3]
    return lam, dz, iters


def test_pendulum_problem_reproduces_the_reference_literals():
    """K = 5 defaults = test_pendulum_5.py:9-24: index arrays entry for entry, values to the literals' print precision."""
    p, ref = kkt.pendulum_problem(), synth.PENDULUM
    assert (p.S, p.C, p.K) == (2, 1, 5)
    for name in ("G_row", "G_col", "C_row", "C_col"):
        assert np.array_equal(getattr(p, name), np.asarray(ref[name], np.int32)), name
    assert np.allclose(p.G_val, ref["G_val"], rtol=0, atol=0)
    assert np.allclose(p.C_val, ref["C_val"], rtol=1e-12)
    assert np.allclose(p.g, ref["g_val"], rtol=3e-6)          # -3.1416 / -314.159 are pi, 100 pi printed to 5-6 digits
    assert np.array_equal(p.c, np.zeros(10))
    lam, dz, iters = solve(p, 1e-12, 50)
    lam_ref, dz_ref, _ = solve(synth.pendulum_system(), 1e-12, 50)
    assert np.allclose(lam, lam_ref, rtol=1e-5) and np.allclose(dz, dz_ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("K", [2, 9, 40])
def test_get_kkt_matches_the_dense_definition(K):
    """Around a rolled-out, non-trivial trajectory: CSR == csr_matrix(dense KKT blocks), c == 0 (consistent rollout),
    and the oracle's solve equals the dense KKT solve."""
    from scipy import sparse
    rng = np.random.default_rng(K)
    plant = kkt.PendulumPlant()
    dt = 0.05
    u = 0.5 * rng.standard_normal((K - 1, 1))
    x = kkt.rollout(plant, (0.3, -0.2), u, dt)
    Q, R, QF = np.diag([2.0, 0.5]), np.array([[0.1]]), np.diag([50.0, 20.0])
    p = kkt.get_kkt(plant, x, u, (0.3, -0.2), (np.pi, 0.0), dt, Q, R, QF, rho=1e-3)
    n, N = 3, 3 * K - 1
    G = np.zeros((N, N)); Cm = np.zeros((2 * K, N)); g = np.zeros(N)
    for k in range(K):
        Qk = QF if k == K - 1 else Q
        G[k * n:k * n + 2, k * n:k * n + 2] = Qk
        g[k * n:k * n + 2] = Qk @ (x[k] - np.array([np.pi, 0.0]))
        if k < K - 1:
            G[k * n + 2, k * n + 2] = R[0, 0]
            g[k * n + 2] = R[0, 0] * u[k, 0]
    Cm[:2, :2] = np.eye(2)
    for k in range(1, K):
        A, B = plant.linearize(x[k - 1], u[k - 1], dt)
        Cm[2 * k:2 * k + 2, (k - 1) * n:(k - 1) * n + 2] = -A
        Cm[2 * k:2 * k + 2, (k - 1) * n + 2:(k - 1) * n + 3] = -B
        Cm[2 * k:2 * k + 2, k * n:k * n + 2] = np.eye(2)
    Gc, Cc = sparse.csr_matrix(G), sparse.csr_matrix(Cm)
    assert np.array_equal(p.G_row, Gc.indptr) and np.array_equal(p.G_col, Gc.indices) and np.allclose(p.G_val, Gc.data)
    assert np.array_equal(p.C_row, Cc.indptr) and np.array_equal(p.C_col, Cc.indices) and np.allclose(p.C_val, Cc.data)
    assert np.allclose(p.g, g) and np.abs(p.c).max() < 1e-14
    lam, dz, _ = solve(p, 1e-14, 400)
    dz_d, lam_d = synth.dense_kkt_solve(p)
    assert np.abs(dz - dz_d).max() < 1e-6 * max(1.0, np.abs(dz_d).max()) and np.allclose(lam, lam_d, rtol=1e-6, atol=1e-8)


def test_numeric_linearisation_default_and_linear_plant():
    class P(kkt.PendulumPlant):
        linearize = kkt.Plant.linearize                                   # force the finite-difference default
    x, u = np.array([0.7, -0.4]), np.array([0.3])
    A, B = kkt.PendulumPlant().linearize(x, u, 0.1)
    An, Bn = P().linearize(x, u, 0.1)
    assert np.allclose(A, An, atol=1e-8) and np.allclose(B, Bn, atol=1e-8)
    rng = np.random.default_rng(0)
    lp = kkt.LinearPlant(np.eye(4) + 0.01 * rng.standard_normal((4, 4)), 0.1 * rng.standard_normal((4, 2)))
    K = 6
    u = rng.standard_normal((K - 1, 2))
    xr = kkt.rollout(lp, np.ones(4), u, 0.0)
    p = kkt.get_kkt(lp, xr, u, np.ones(4), np.zeros(4), 0.0, np.eye(4), 0.1 * np.eye(2), 10 * np.eye(4))
    assert (p.S, p.C, p.K) == (4, 2, 6) and np.abs(p.c).max() < 1e-14
    lam, dz, _ = solve(p, 1e-14, 400)
    dz_d, lam_d = synth.dense_kkt_solve(p)
    assert np.abs(dz - dz_d).max() < 1e-6 and np.allclose(lam, lam_d, rtol=1e-6, atol=1e-8)
