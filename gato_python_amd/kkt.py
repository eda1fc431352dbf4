"""Upstream KKT producer.

The reference's IIWA script obtains `G, g, C, c` from code outside its repository
(`test_IIWA50.py:6-13`: `getIIWA(50)` then `getKKT(trajoptReference, x, u, xs, xg, dt)`), turns the dense matrices
into CSR with `scipy.sparse.csr_matrix` (`test_IIWA50.py:15-18`, `test_pendulum_5.py:28-29`) and hands the index
and value arrays to `gpu_library.linsys_solve`.  This module is that producer for any plant that can linearise itself:
one Gauss-Newton / SQP linearisation of the discrete-time optimal-control problem

    min  sum_k 1/2 (x_k - xg)' Q (x_k - xg) + 1/2 u_k' R u_k  +  1/2 (x_{K-1} - xg)' QF (x_{K-1} - xg)
    s.t. x_0 = xs,   x_{k+1} = f(x_k, u_k)

around a trajectory `(x, u)`, in exactly the block structure the solver's CSR scatter assumes
(`src/gato_schur.cuh:674-743`):

    G = blockdiag(Q, R, Q, R, ..., QF)             g = (Q (x_0 - xg), R u_0, ..., QF (x_{K-1} - xg))
    C = [ I                      ]                 c = (x_0 - xs, x_1 - f(x_0, u_0), ..., x_{K-1} - f(x_{K-2}, u_{K-2}))
        [ -A_0 -B_0  I           ]
        [          -A_1 -B_1  I  ]

`pendulum_problem()` with its defaults reproduces the reference's own literals (`test_pendulum_5.py:9-24`): the same
index arrays entry for entry (structural zeros dropped the way `csr_matrix(dense)` drops them) and the same values up
to the 5-6 digits the literals were printed with.  Input generation only: nothing here solves anything.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .synth import KKTSystem


class Plant:
    """A discrete-time plant: state size S, control size C, one-step map and its Jacobians."""
    S: int
    C: int

    def step(self, x: np.ndarray, u: np.ndarray, dt: float) -> np.ndarray:
        raise NotImplementedError

    def linearize(self, x: np.ndarray, u: np.ndarray, dt: float):
        """(A, B) = (d step / dx, d step / du) at (x, u); central differences unless a plant overrides it."""
        eps = 1e-6
        A = np.empty((self.S, self.S))
        B = np.empty((self.S, self.C))
        for j in range(self.S):
            e = np.zeros(self.S); e[j] = eps
            A[:, j] = (self.step(x + e, u, dt) - self.step(x - e, u, dt)) / (2 * eps)
        for j in range(self.C):
            e = np.zeros(self.C); e[j] = eps
            B[:, j] = (self.step(x, u + e, dt) - self.step(x, u - e, dt)) / (2 * eps)
        return A, B


@dataclass
class PendulumPlant(Plant):
    """Torque-driven pendulum, state (theta, omega), explicit Euler: the plant behind the reference's fixture
    (`-0.981 = -dt * gravity`, `0.1 = dt` in `test_pendulum_5.py:15`)."""
    gravity: float = 9.81
    S: int = 2
    C: int = 1

    def step(self, x, u, dt):
        th, om = x
        return np.array([th + dt * om, om + dt * (-self.gravity * np.sin(th) + u[0])])

    def linearize(self, x, u, dt):
        A = np.array([[1.0, dt], [-dt * self.gravity * np.cos(x[0]), 1.0]])
        B = np.array([[0.0], [dt]])
        return A, B


@dataclass
class LinearPlant(Plant):
    """x+ = Ad x + Bd u (already discrete; dt is ignored)."""
    Ad: np.ndarray
    Bd: np.ndarray

    def __post_init__(self):
        self.Ad = np.asarray(self.Ad, np.float64)
        self.Bd = np.asarray(self.Bd, np.float64)
        self.S, self.C = self.Bd.shape

    def step(self, x, u, dt):
        return self.Ad @ x + self.Bd @ u

    def linearize(self, x, u, dt):
        return self.Ad, self.Bd


def _dense_block_to_csr_rows(blocks):
    """blocks: per row a list of (first_col, row_values).  Returns (indptr, indices, data) with zeros dropped and
    columns ascending - what scipy.sparse.csr_matrix(dense) produces."""
    indptr, indices, data = [0], [], []
    for row in blocks:
        for col0, vals in sorted(row, key=lambda t: t[0]):
            nz = np.nonzero(vals)[0]
            indices.extend((col0 + nz).tolist())
            data.extend(np.asarray(vals)[nz].tolist())
        indptr.append(len(indices))
    return (np.asarray(indptr, np.int32), np.asarray(indices, np.int32), np.asarray(data, np.float64))


def get_kkt(plant: Plant, x, u, xs, xg, dt: float, Q, R, QF, rho: float = 1e-3) -> KKTSystem:
    """One linearisation of the OCP around (x [K,S], u [K-1,C]) as a KKTSystem in `linsys_solve` argument order."""
    x = np.atleast_2d(np.asarray(x, np.float64))
    K, S = x.shape
    C = plant.C
    u = np.asarray(u, np.float64).reshape(K - 1, C)
    xs, xg = np.asarray(xs, np.float64), np.asarray(xg, np.float64)
    Q, R, QF = (np.asarray(m, np.float64) for m in (Q, R, QF))
    assert S == plant.S and Q.shape == (S, S) and QF.shape == (S, S) and R.shape == (C, C)
    n = S + C
    G_rows, C_rows = [], []
    g = np.zeros(n * K - C)
    c = np.zeros(S * K)
    for k in range(K):
        Qk = QF if k == K - 1 else Q
        for i in range(S):
            G_rows.append([(k * n, Qk[i])])
        g[k * n:k * n + S] = Qk @ (x[k] - xg)
        if k < K - 1:
            for i in range(C):
                G_rows.append([(k * n + S, R[i])])
            g[k * n + S:(k + 1) * n] = R @ u[k]
    eye = np.eye(S)
    for i in range(S):
        C_rows.append([(0, eye[i])])
    c[:S] = x[0] - xs
    for k in range(1, K):
        A, B = plant.linearize(x[k - 1], u[k - 1], dt)
        for i in range(S):
            C_rows.append([((k - 1) * n, np.concatenate([-A[i], -B[i]])), (k * n, eye[i])])
        c[k * S:(k + 1) * S] = x[k] - plant.step(x[k - 1], u[k - 1], dt)
    G_row, G_col, G_val = _dense_block_to_csr_rows(G_rows)
    C_row, C_col, C_val = _dense_block_to_csr_rows(C_rows)
    return KKTSystem(S, C, K, G_row, G_col, G_val, C_row, C_col, C_val, g, c, rho)


def pendulum_problem(K: int = 5, dt: float = 0.1, x=None, u=None, xs=(0.0, 0.0), xg=(np.pi, 0.0),
                     q: float = 1.0, r: float = 0.1, qf: float = 100.0, rho: float = 1e-3) -> KKTSystem:
    """The reference fixture's problem for any horizon: swing-up target (pi, 0), Q = q I, R = r, QF = qf I, linearised
    around the rest trajectory unless (x, u) are given.  K = 5 gives `test_pendulum_5.py:9-24`."""
    plant = PendulumPlant()
    x = np.zeros((K, 2)) if x is None else np.asarray(x, np.float64)
    u = np.zeros((K - 1, 1)) if u is None else np.asarray(u, np.float64)
    return get_kkt(plant, x, u, xs, xg, dt, q * np.eye(2), r * np.eye(1), qf * np.eye(2), rho)


def rollout(plant: Plant, xs, u, dt: float) -> np.ndarray:
    """x [K,S] of the controls u [K-1,C] from xs: a dynamically consistent trajectory (c = 0 except numerically)."""
    u = np.asarray(u, np.float64)
    x = np.empty((u.shape[0] + 1, plant.S))
    x[0] = xs
    for k in range(u.shape[0]):
        x[k + 1] = plant.step(x[k], u[k], dt)
    return x
