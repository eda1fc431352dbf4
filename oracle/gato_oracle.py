"""ORACLE - test infrastructure, not product code.

CPU (numpy) restatement of the reference's hot path: CSR->dense scatter, Schur-complement
and block-Jacobi assembly, symmetric-stair preconditioner, PCG on the block-tridiagonal
Schur system, dz back-substitution.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; gato_python_amd never does.

Parity pin: the reference is CUDA-only (cooperative groups, cudaMallocAsync) and cannot be
built or run here, so there is no oracle/_ref.  The restatement is pinned by (1) the
reference-owned pendulum inputs (test_pendulum_5.py:9-24) checked with the reference test's
own oracle construction, the dense KKT solve (test_pendulum_5.py:28-37), with rho added as
the solver adds it, and (2) the same dense fp64 KKT solve on seeded synthetic systems for the
shapes the reference has no inputs for (IIWA 14/7, 32/16).  For those shapes parity is
pinned by the dense solve only ("unpinned by the reference").

All arrays are in the reference's memory order:
  G_dense : per knot [Q_k (S*S col-major) | R_k (C*C col-major)], last knot Q only   (gato_defines.h:36)
  C_dense : per knot k<K-1 [A_k (S*S col-major) | B_k (S*C col-major)]                (gato_defines.h:37)
  S, Pinv : per block-row [left | main | right], each S*S col-major                   (gato_utils.cuh:44-73)
Deviations from the reference: D1 (K4 boundary rule everywhere), D2 (last
state row of dz uses no A / lambda_K), D3 (inverses go to a separate buffer), D4 (gamma_0 gets +c_0).
"""
from __future__ import annotations

import numpy as np


# --------------------------------------------------------------------------------------------
# layout helpers
# --------------------------------------------------------------------------------------------
def g_dense_size(S, C, K):
    return (S * S + C * C) * K - C * C


def c_dense_size(S, C, K):
    return (S * S + S * C) * (K - 1)


def unpack_G(G_dense, S, C, K):
    """-> Q[K,S,S], R[K-1,C,C] as math matrices (row, col)."""
    st = S * S + C * C
    Q = np.empty((K, S, S), G_dense.dtype)
    R = np.empty((max(K - 1, 0), C, C), G_dense.dtype)
    for k in range(K):
        Q[k] = G_dense[k * st: k * st + S * S].reshape(S, S).T        # col-major -> math
        if k < K - 1:
            R[k] = G_dense[k * st + S * S: (k + 1) * st].reshape(C, C).T
    return Q, R


def pack_G(Q, R):
    K, S, _ = Q.shape
    C = R.shape[1] if K > 1 else 0
    st = S * S + C * C
    out = np.zeros(g_dense_size(S, C, K), Q.dtype)
    for k in range(K):
        out[k * st: k * st + S * S] = Q[k].T.reshape(-1)
        if k < K - 1:
            out[k * st + S * S: (k + 1) * st] = R[k].T.reshape(-1)
    return out


def unpack_C(C_dense, S, C, K):
    """-> A[K-1,S,S], B[K-1,S,C]."""
    st = S * S + S * C
    A = np.empty((K - 1, S, S), C_dense.dtype)
    B = np.empty((K - 1, S, C), C_dense.dtype)
    for k in range(K - 1):
        A[k] = C_dense[k * st: k * st + S * S].reshape(S, S).T
        B[k] = C_dense[k * st + S * S: (k + 1) * st].reshape(C, S).T
    return A, B


def unpack_bd(M_bd, S, K):
    """bd-format -> L[K,S,S], M[K,S,S], R[K,S,S] (math orientation)."""
    blk = M_bd.reshape(K, 3, S, S).transpose(0, 1, 3, 2)            # col-major -> math
    return blk[:, 0], blk[:, 1], blk[:, 2]


def pack_bd(L, M, R):
    K, S, _ = M.shape
    out = np.stack([L, M, R], axis=1).transpose(0, 1, 3, 2)
    return np.ascontiguousarray(out).reshape(-1)


# --------------------------------------------------------------------------------------------
# A1: CSR -> dense per-knot blocks            (src/gato_schur.cuh:674-743)
# --------------------------------------------------------------------------------------------
def convert(G_row, G_col, G_val, C_row, C_col, C_val, S, C, K, rho, dtype=np.float64):
    n = S + C
    N = n * K - C
    G_dense = np.zeros(g_dense_size(S, C, K), dtype)
    C_dense = np.zeros(c_dense_size(S, C, K), dtype)
    G_val = np.asarray(G_val, dtype)
    C_val = np.asarray(C_val, dtype)
    rho = dtype(rho)
    # csr_to_custom_G, gato_schur.cuh:674-704
    for row in range(N):
        in_set_row = row % n
        set_offset = (row // n) * (S * S + C * C)
        for it in range(G_row[row], G_row[row + 1]):
            col = int(G_col[it])
            in_set_col = col % n
            v = G_val[it] + (rho if col == row else dtype(0))
            if in_set_col < S:
                G_dense[set_offset + in_set_col * S + in_set_row] = v
            else:
                G_dense[set_offset + S * S + (in_set_col - S) * C + (in_set_row - S)] = v
    # csr_to_custom_C, gato_schur.cuh:707-743
    for row in range(S, S * K):
        block_row = row // S - 1
        for it in range(C_row[row], C_row[row + 1]):
            col = int(C_col[it])
            if col // n > block_row:
                continue
            C_dense[block_row * (S * S + S * C) + (col % n) * S + row % S] = C_val[it]
    return G_dense, C_dense


# --------------------------------------------------------------------------------------------
# A10: Gauss-Jordan inverse without pivoting   (src/gato_utils.cuh:468-586), batched over knots
# --------------------------------------------------------------------------------------------
def gauss_jordan_inverse(A):
    """A: [..., n, n].  Same elimination as invertMatrix: for each pivot, scale the pivot row
    by 1/pivot and eliminate the column from every other row, on the augmented pair (A | I)."""
    A = A.copy()
    n = A.shape[-1]
    I = np.zeros_like(A)
    I[..., np.arange(n), np.arange(n)] = 1
    for p in range(n):
        piv = A[..., p, p][..., None]
        rowA = A[..., p, :] / piv
        rowI = I[..., p, :] / piv
        f = A[..., :, p][..., None].copy()
        f[..., p, :] = 0
        A = A - f * rowA[..., None, :]
        I = I - f * rowI[..., None, :]
        A[..., p, :] = rowA
        I[..., p, :] = rowI
    return I


# --------------------------------------------------------------------------------------------
# A2: Schur complement + block-Jacobi main blocks + gamma   (src/gato_schur.cuh:13-460)
# --------------------------------------------------------------------------------------------
def form_schur(G_dense, C_dense, g, c, S, C, K):
    dtype = G_dense.dtype
    n = S + C
    Q, R = unpack_G(G_dense, S, C, K)
    A, B = unpack_C(C_dense, S, C, K)
    g = np.asarray(g, dtype)
    c = np.asarray(c, dtype).reshape(K, S)
    q = np.stack([g[k * n: k * n + S] for k in range(K)])
    r = np.stack([g[k * n + S: (k + 1) * n] for k in range(K - 1)]) if K > 1 else np.zeros((0, C), dtype)

    Qi = gauss_jordan_inverse(Q)                       # :86-88, :224-235
    Ri = gauss_jordan_inverse(R) if K > 1 else R

    Sl = np.zeros((K, S, S), dtype)
    Sm = np.zeros((K, S, S), dtype)
    Sr = np.zeros((K, S, S), dtype)
    Pm = np.zeros((K, S, S), dtype)
    gamma = np.zeros((K, S), dtype)

    # k = 0 (:26-147): S[0].main = -Q0^-1, Pinv[0].main = -Q0, gamma_0 = -Q0^-1 q0 (+ c_0: D4)
    Sm[0] = -Qi[0]
    Pm[0] = -Q[0]
    gamma[0] = c[0] - Qi[0] @ q[0]
    if K > 1:
        phi = A @ Qi[:-1]                              # :277-285  phi_k = A Q_{k-1}^-1
        BR = B @ Ri                                    # :293-301
        gt = (Qi[1:] @ q[1:, :, None])[..., 0] - c[1:]                       # :306-313
        gt = gt + (phi @ q[:-1, :, None])[..., 0] + (BR @ r[:, :, None])[..., 0]   # :316-338
        theta = phi @ A.transpose(0, 2, 1) + Qi[1:] + BR @ B.transpose(0, 2, 1)   # :342-384
        Sl[1:] = -phi                                  # :388-394
        Sm[1:] = -theta                                # :398-404
        Pm[1:] = -gauss_jordan_inverse(theta)          # :407-422
        gamma[1:] = -gt                                # :435-438
        Sr[:-1] = -phi.transpose(0, 2, 1)              # :443-455
    Z = np.zeros_like(Pm)
    return (pack_bd(Sl, Sm, Sr), pack_bd(Z, Pm, Z), gamma.reshape(-1),
            pack_G(Qi, Ri))


# --------------------------------------------------------------------------------------------
# A3: symmetric-stair off-diagonals of Pinv     (src/gato_schur.cuh:497-649)
# --------------------------------------------------------------------------------------------
def form_ss(S_bd, Pinv_bd, S, K):
    Sl, _, _ = unpack_bd(S_bd, S, K)
    _, Pm, _ = unpack_bd(Pinv_bd, S, K)
    Pl = np.zeros_like(Pm)
    Pr = np.zeros_like(Pm)
    if K > 1:
        Pl[1:] = -((Pm[1:] @ Sl[1:]) @ Pm[:-1])                                  # :578-611
        Pr[:-1] = -((Pm[:-1] @ Sl[1:].transpose(0, 2, 1)) @ Pm[1:])              # :614-648
    return pack_bd(Pl, Pm, Pr)


# --------------------------------------------------------------------------------------------
# A7: block-tridiagonal SpMV with the K4 boundary rule   (src/gato_utils.cuh:153-185)
# --------------------------------------------------------------------------------------------
def point_jacobi(S_bd, S, K):
    """The reference built with BLOCK_J_PRECON = SS_PRECON = 0 (include/gato_defines.h:9-10): Pinv[k].main =
    diag(1 / S[k].main_ii) (src/gato_schur.cuh:424-428), zeros elsewhere."""
    _, M, _ = unpack_bd(S_bd, S, K)
    Z = np.zeros_like(M)
    D = np.zeros_like(M)
    idx = np.arange(S)
    D[:, idx, idx] = 1.0 / M[:, idx, idx]
    return pack_bd(Z, D, Z)


def bt_matvec(L, M, R, x):
    """x: [K,S] -> y[K,S];  first row uses main,right; last row uses left,main."""
    y = (M @ x[:, :, None])[..., 0]
    if x.shape[0] > 1:
        y[1:] += (L[1:] @ x[:-1, :, None])[..., 0]
        y[:-1] += (R[:-1] @ x[1:, :, None])[..., 0]
    return y


# --------------------------------------------------------------------------------------------
# A5: PCG, K4 semantics                           (src/gato_pcg.cuh:270-439)
# --------------------------------------------------------------------------------------------
def pcg(S_bd, Pinv_bd, gamma, S, K, exit_tol, max_iters, return_history=False, lam0=None):
    """lam0: optional initial guess (true warm start, r0 = gamma - S lam0) - an extension; the reference resets
    lambda to zero whatever it is given (D5)."""
    dtype = S_bd.dtype
    Sl, Sm, Sr = unpack_bd(S_bd, S, K)
    Pl, Pm, Pr = unpack_bd(Pinv_bd, S, K)
    lam = np.zeros((K, S), dtype)                     # :303 (lambda reset to 0, D5)
    r = np.asarray(gamma, dtype).reshape(K, S).copy() # :301
    if lam0 is not None:
        lam = np.asarray(lam0, dtype).reshape(K, S).copy()
        r = r - bt_matvec(Sl, Sm, Sr, lam)
    rt = bt_matvec(Pl, Pm, Pr, r)                     # :316-318
    p = rt.copy()                                     # :322-325
    eta = dtype.type(np.sum(r * rt, dtype=dtype))     # :327-335
    iters = max_iters                                 # :311-313
    hist = [float(eta)]
    tol = dtype.type(exit_tol)
    for it in range(max_iters):                       # :348
        ups = bt_matvec(Sl, Sm, Sr, p)                # :349-351
        v = dtype.type(np.sum(p * ups, dtype=dtype))  # :353-357
        with np.errstate(all="ignore"):
            alpha = eta / v                           # :364
        lam = lam + alpha * p                         # :373-377
        r = r - alpha * ups
        rt = bt_matvec(Pl, Pm, Pr, r)                 # :380-381
        eta_new = dtype.type(np.sum(r * rt, dtype=dtype))   # :382-394
        hist.append(float(eta_new))
        if abs(eta_new) < tol:                        # :404-411
            iters = it
            break
        with np.errstate(all="ignore"):
            beta = eta_new / eta                      # :415
        p = rt + beta * p                             # :416-419
        eta = eta_new                                 # :420
    if return_history:
        return lam.reshape(-1), iters, hist
    return lam.reshape(-1), iters


def pcg_single_reduction(S_bd, Pinv_bd, gamma, S, K, exit_tol, max_iters):
    """Chronopoulos-Gear single-reduction PCG - NOT the reference's recurrence; restated here only to check the opt-in
    HIP variant (gato_pcg_cg1.hip).  Same exit test quantity (r . Pinv r) and iteration numbering as pcg()."""
    dtype = S_bd.dtype
    Sl, Sm, Sr = unpack_bd(S_bd, S, K)
    Pl, Pm, Pr = unpack_bd(Pinv_bd, S, K)
    lam = np.zeros((K, S), dtype)
    r = np.asarray(gamma, dtype).reshape(K, S).copy()
    u = bt_matvec(Pl, Pm, Pr, r)
    w = bt_matvec(Sl, Sm, Sr, u)
    gam = dtype.type(np.sum(r * u, dtype=dtype))
    delta = dtype.type(np.sum(w * u, dtype=dtype))
    p = np.zeros_like(r)
    s = np.zeros_like(r)
    beta = dtype.type(0)
    with np.errstate(all="ignore"):
        alpha = gam / delta
    iters = max_iters
    tol = dtype.type(exit_tol)
    for it in range(max_iters):
        p = u + beta * p
        s = w + beta * s
        lam = lam + alpha * p
        r = r - alpha * s
        u = bt_matvec(Pl, Pm, Pr, r)
        w = bt_matvec(Sl, Sm, Sr, u)
        gam_new = dtype.type(np.sum(r * u, dtype=dtype))
        delta = dtype.type(np.sum(w * u, dtype=dtype))
        if abs(gam_new) < tol:
            iters = it
            break
        with np.errstate(all="ignore"):
            beta = gam_new / gam
            alpha = gam_new / (delta - beta * gam_new / alpha)
        gam = gam_new
    return lam.reshape(-1), iters


def compute_dz(Ginv_dense, C_dense, g, lam, S, C, K):
    dtype = Ginv_dense.dtype
    n = S + C
    Qi, Ri = unpack_G(Ginv_dense, S, C, K)
    A, B = unpack_C(C_dense, S, C, K)
    g = np.asarray(g, dtype)
    lam = np.asarray(lam, dtype).reshape(K, S)
    dz = np.zeros(n * K - C, dtype)
    for k in range(K):
        qk = g[k * n: k * n + S]
        t = qk - lam[k]
        if k < K - 1:
            t = t - A[k].T @ lam[k + 1]                               # :833-846
            rk = g[k * n + S: (k + 1) * n]
            dz[k * n + S: (k + 1) * n] = Ri[k] @ (rk - B[k].T @ lam[k + 1])   # :763-810
        dz[k * n: k * n + S] = Qi[k] @ t                              # :849-866
    return dz


# --------------------------------------------------------------------------------------------
# L3/L4: whole solve                              (gpu_library.cu:25-83)
# --------------------------------------------------------------------------------------------
def linsys_solve(G_row, G_col, G_val, C_row, C_col, C_val, g, c, S, C, K,
                 exit_tol, max_iters, rho, dtype=np.float32, return_all=False, precon_mode=0):
    """precon_mode: 0 = symmetric stair (BLOCK_J_PRECON = SS_PRECON = 1, the reference's setting, gato_defines.h:9-10),
    1 = block-Jacobi (SS_PRECON = 0: gato_form_ss is not launched, gato_schur.cuh:965-970), 2 = point-Jacobi (both 0)."""
    dtype = np.dtype(dtype).type
    Gd, Cd = convert(G_row, G_col, G_val, C_row, C_col, C_val, S, C, K, rho, dtype)
    g = np.asarray(g, dtype)
    c = np.asarray(c, dtype)
    S_bd, P_bd, gamma, Ginv = form_schur(Gd, Cd, g, c, S, C, K)
    if precon_mode == 0:
        P_bd = form_ss(S_bd, P_bd, S, K)
    elif precon_mode == 2:
        P_bd = point_jacobi(S_bd, S, K)
    lam, iters, hist = pcg(S_bd, P_bd, gamma, S, K, exit_tol, max_iters, return_history=True)
    dz = compute_dz(Ginv, Cd, g, lam, S, C, K)
    if return_all:
        return dict(G_dense=Gd, C_dense=Cd, S=S_bd, Pinv=P_bd, gamma=gamma, Ginv=Ginv,
                    lam=lam, dz=dz, iters=iters, eta=hist)
    return lam, dz, iters
