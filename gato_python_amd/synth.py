"""Synthetic OCP-structured KKT inputs for the gato PCG/Schur hot path.

The reference ships exactly one set of inputs (the pendulum literals of
test_pendulum_5.py:9-24); its IIWA script imports data from outside the repository
(test_IIWA50.py:6-9).  Every other shape named in BASELINE.json therefore runs on
seeded synthetic systems with the same block structure the reference's CSR scatter
assumes (src/gato_schur.cuh:674-743):

  G = blockdiag(Q_0, R_0, Q_1, R_1, ..., Q_{K-1})                      N x N
  C = [ I                                   ]   row-block 0            SK x N
      [ A_0 B_0 I                           ]   row-block 1
      [         A_1 B_1 I                   ]   ...
  g = (q_0, r_0, q_1, ..., q_{K-1}),  c = (c_0, ..., c_{K-1})

with N = (S+C)K - C.  A_k, B_k are the raw stored values (= -A, -B of the dynamics,
as in the pendulum data).  This module is input generation only: it never solves
anything and is used by bench.py, the tests and the oracle alike.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

# Reference-owned fixture, verbatim values of test_pendulum_5.py:9-24 (== README.md:40-55).
PENDULUM = dict(
    S=2, C=1, K=5,
    G_row=[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14],
    G_col=[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13],
    G_val=[1., 1., 0.1, 1., 1., 0.1, 1., 1., 0.1, 1., 1., 0.1, 100., 100.],
    C_row=[0, 1, 2, 5, 9, 12, 16, 19, 23, 26, 30],
    C_col=[0, 1, 0, 1, 3, 0, 1, 2, 4, 3, 4, 6, 3, 4, 5, 7, 6, 7, 9, 6, 7, 8, 10,
           9, 10, 12, 9, 10, 11, 13],
    C_val=[1., 1., -1., -0.1, 1., 0.981, -1., -0.1, 1., -1., -0.1, 1., 0.981, -1.,
           -0.1, 1., -1., -0.1, 1., 0.981, -1., -0.1, 1., -1., -0.1, 1., 0.981,
           -1., -0.1, 1.],
    g_val=[-3.1416, 0., 0., -3.1416, 0., 0., -3.1416, 0., 0., -3.1416, 0., 0.,
           -314.159, 0.],
    c_val=[0.] * 10,
    input_lambda=[0.] * 10,
    testiters=10, exit_tol=1e-6, max_iters=10, warm_start=False, rho=.001,
)


@dataclass
class KKTSystem:
    """CSR inputs in the order gpu_library.linsys_solve takes them (gpu_library.cu:85-87)."""
    S: int
    C: int
    K: int
    G_row: np.ndarray
    G_col: np.ndarray
    G_val: np.ndarray
    C_row: np.ndarray
    C_col: np.ndarray
    C_val: np.ndarray
    g: np.ndarray
    c: np.ndarray
    rho: float = 1e-3

    @property
    def n(self) -> int:
        return self.S + self.C

    @property
    def N(self) -> int:
        return self.n * self.K - self.C

    def csr_args(self):
        return (self.G_row, self.G_col, self.G_val, self.C_row, self.C_col, self.C_val,
                self.g, self.c)

    def astype(self, dtype) -> "KKTSystem":
        return KKTSystem(self.S, self.C, self.K, self.G_row, self.G_col,
                         self.G_val.astype(dtype), self.C_row, self.C_col,
                         self.C_val.astype(dtype), self.g.astype(dtype),
                         self.c.astype(dtype), self.rho)


def pendulum_system() -> KKTSystem:
    p = PENDULUM
    return KKTSystem(p["S"], p["C"], p["K"],
                     np.asarray(p["G_row"], np.int32), np.asarray(p["G_col"], np.int32),
                     np.asarray(p["G_val"], np.float64),
                     np.asarray(p["C_row"], np.int32), np.asarray(p["C_col"], np.int32),
                     np.asarray(p["C_val"], np.float64),
                     np.asarray(p["g_val"], np.float64), np.asarray(p["c_val"], np.float64),
                     p["rho"])


def make_blocks(S: int, C: int, K: int, seed: int = 0, dense_q: bool = False):
    """Seeded per-knot blocks (SURVEY.md section 8d generator).  Returns fp64 arrays
    Q[K,S,S], R[K-1,C,C], A[K-1,S,S], B[K-1,S,C], q[K,S], r[K-1,C], c[K,S]."""
    rng = np.random.default_rng(seed)
    Q = np.zeros((K, S, S))
    idx = np.arange(S)
    if dense_q:
        M = 0.1 * rng.standard_normal((K, S, S))
        Q = M @ M.transpose(0, 2, 1)
        Q[:, idx, idx] += rng.uniform(0.5, 2.0, (K, S))
    else:
        Q[:, idx, idx] = rng.uniform(0.1, 10.0, (K, S))
    Q[K - 1] *= 100.0                                   # terminal cost, cf. test_pendulum_5.py:11
    R = np.zeros((K - 1, C, C))
    R[:, np.arange(C), np.arange(C)] = rng.uniform(0.01, 1.0, (K - 1, C))
    A = -(np.eye(S)[None] + 0.01 * rng.standard_normal((K - 1, S, S)))
    B = -0.1 * rng.standard_normal((K - 1, S, C))
    q = rng.standard_normal((K, S))
    r = rng.standard_normal((K - 1, C))
    c = 0.1 * rng.standard_normal((K, S))
    c[0] = 0.0
    return Q, R, A, B, q, r, c


def blocks_to_csr(Q, R, A, B, q, r, c, rho: float = 1e-3, dense_q: bool | None = None) -> KKTSystem:
    """Emit the CSR arrays exactly as scipy.sparse.csr_matrix(...).indptr/.indices/.data
    would (sorted columns, explicit structural entries only), cf. test_pendulum_5.py:28-29."""
    K, S, _ = Q.shape
    C = R.shape[1] if K > 1 else 0
    n = S + C
    N = n * K - C
    if dense_q is None:
        off = Q.copy()
        off[:, np.arange(S), np.arange(S)] = 0
        dense_q = bool(np.any(off != 0))

    # ---- G: block diagonal; Q rows hold S entries (dense) or 1 (diagonal); R is diagonal
    # unless it has off-diagonal entries.
    r_off = R.copy()
    if K > 1:
        r_off[:, np.arange(C), np.arange(C)] = 0
    dense_r = bool(np.any(r_off != 0))
    q_nnz = S if dense_q else 1
    r_nnz = C if dense_r else 1
    row_nnz = np.empty(N, np.int64)
    rows = np.arange(N)
    in_row = rows % n
    row_nnz[:] = np.where(in_row < S, q_nnz, r_nnz)
    G_row = np.zeros(N + 1, np.int64)
    np.cumsum(row_nnz, out=G_row[1:])
    G_col = np.empty(G_row[-1], np.int64)
    G_val = np.empty(G_row[-1], np.float64)
    knot = rows // n
    # state rows
    srows = rows[in_row < S]
    sk = knot[in_row < S]
    si = in_row[in_row < S]
    if dense_q:
        base = G_row[srows][:, None] + np.arange(S)[None]
        G_col[base] = (sk * n)[:, None] + np.arange(S)[None]
        G_val[base] = Q[sk, si, :]
    else:
        G_col[G_row[srows]] = srows
        G_val[G_row[srows]] = Q[sk, si, si]
    crows = rows[in_row >= S]
    if crows.size:
        ck = knot[in_row >= S]
        ci = in_row[in_row >= S] - S
        if dense_r:
            base = G_row[crows][:, None] + np.arange(C)[None]
            G_col[base] = (ck * n + S)[:, None] + np.arange(C)[None]
            G_val[base] = R[ck, ci, :]
        else:
            G_col[G_row[crows]] = crows
            G_val[G_row[crows]] = R[ck, ci, ci]

    # ---- C: row-block 0 identity; row-block k>=1: [A_{k-1} B_{k-1}] on (x_{k-1},u_{k-1}), I on x_k
    SK = S * K
    c_nnz = np.full(SK, n + 1, np.int64)
    c_nnz[:S] = 1
    C_row = np.zeros(SK + 1, np.int64)
    np.cumsum(c_nnz, out=C_row[1:])
    C_col = np.empty(C_row[-1], np.int64)
    C_val = np.empty(C_row[-1], np.float64)
    C_col[:S] = np.arange(S)
    C_val[:S] = 1.0
    if K > 1:
        rr = np.arange(S, SK)
        kb = rr // S - 1                      # block_row of csr_to_custom_C (gato_schur.cuh:730)
        ii = rr % S
        base = C_row[rr][:, None] + np.arange(n)[None]
        C_col[base] = (kb * n)[:, None] + np.arange(n)[None]
        C_val[base[:, :S]] = A[kb, ii, :]
        C_val[base[:, S:]] = B[kb, ii, :]
        last = C_row[rr] + n
        C_col[last] = (kb + 1) * n + ii
        C_val[last] = 1.0

    g = np.empty(N)
    gk = g[: (K - 1) * n].reshape(K - 1, n) if K > 1 else None
    if K > 1:
        gk[:, :S] = q[:-1]
        gk[:, S:] = r
    g[(K - 1) * n:] = q[-1]
    return KKTSystem(S, C, K, G_row.astype(np.int32), G_col.astype(np.int32), G_val,
                     C_row.astype(np.int32), C_col.astype(np.int32), C_val,
                     g, c.reshape(-1).copy(), rho)


def make_system(S: int, C: int, K: int, seed: int = 0, dense_q: bool = False,
                rho: float = 1e-3) -> KKTSystem:
    return blocks_to_csr(*make_blocks(S, C, K, seed, dense_q), rho=rho, dense_q=dense_q)


def dense_kkt(sys: KKTSystem, with_rho: bool = True):
    """Dense [[G(+rho I), C^T],[C, 0]] and rhs [g; c] in fp64 - the reference test's own
    oracle construction (test_pendulum_5.py:28-34); with_rho adds the rho the solver adds
    (gato_schur.cuh:697,700)."""
    from scipy import sparse
    N, SK = sys.N, sys.S * sys.K
    G = sparse.csr_matrix((sys.G_val.astype(np.float64), sys.G_col, sys.G_row), shape=(N, N)).toarray()
    Cm = sparse.csr_matrix((sys.C_val.astype(np.float64), sys.C_col, sys.C_row), shape=(SK, N)).toarray()
    if with_rho:
        G = G + sys.rho * np.eye(N)
    A = np.block([[G, Cm.T], [Cm, np.zeros((SK, SK))]])
    rhs = np.concatenate([sys.g.astype(np.float64), sys.c.astype(np.float64)])
    return A, rhs


def dense_kkt_solve(sys: KKTSystem, with_rho: bool = True):
    """(dz, lambda) of the dense fp64 KKT solve."""
    A, rhs = dense_kkt(sys, with_rho)
    x = np.linalg.solve(A, rhs)
    return x[: sys.N], x[sys.N:]
