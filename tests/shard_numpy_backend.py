"""Test-only backend for gato_python_amd.dist.ShardedPCG: the same shard protocol (records, ghost
advance, ring of gathered buffers) restated in numpy on top of the oracle's block-tridiagonal product,
so that the collective schedule can run under gloo on CPU ranks.  Not product code."""
import numpy as np
import torch

from gato_python_amd.dist import knot_ranges
from oracle import gato_oracle as o


class NumpyShardBackend:
    def __init__(self, S, K, rank, nranks, S_bd, P_bd, gamma, exit_tol, max_iters):
        self.S, self.K, self.rank, self.nranks = S, K, rank, nranks
        self.k0, self.k1 = knot_ranges(K, nranks)[rank]
        self.dtype = S_bd.dtype
        self.Sl, self.Sm, self.Sr = [x[self.k0:self.k1].copy() for x in o.unpack_bd(S_bd, S, K)]
        self.Pl, self.Pm, self.Pr = [x[self.k0:self.k1].copy() for x in o.unpack_bd(P_bd, S, K)]
        self.gamma = np.asarray(gamma, self.dtype).reshape(K, S)
        self.tol, self.max_iters = self.dtype.type(exit_tol), max_iters
        self.rec = 2 * S + 1
        self.first, self.last = self.k0 == 0, self.k1 == K
        self.is_done, self.iters = False, max_iters

    # ---- helpers ---------------------------------------------------------------------------------
    def new_record(self):
        return torch.zeros(self.rec, dtype=torch.from_numpy(np.zeros(1, self.dtype)).dtype)

    def new_gathered(self):
        return torch.zeros(self.rec * self.nranks, dtype=torch.from_numpy(np.zeros(1, self.dtype)).dtype)

    def _matvec(self, L, M, R, x, gl, gr):
        y = (M @ x[:, :, None])[..., 0]
        y[1:] += (L[1:] @ x[:-1, :, None])[..., 0]
        y[:-1] += (R[:-1] @ x[1:, :, None])[..., 0]
        if not self.first:
            y[0] += L[0] @ gl
        if not self.last:
            y[-1] += R[-1] @ gr
        return y

    def _pack(self, send, partial, y):
        rec = np.concatenate([[partial], y[0], y[-1]]).astype(self.dtype)
        send.copy_(torch.from_numpy(rec))

    def _sum(self, recv):      # rank-order sum of the partials
        v = recv.numpy().reshape(self.nranks, self.rec)[:, 0]
        acc = self.dtype.type(0)
        for x in v:
            acc = self.dtype.type(acc + x)
        return acc

    def _halo(self, recv):
        g = recv.numpy().reshape(self.nranks, self.rec)
        S = self.S
        gl = g[self.rank - 1, 1 + S:] if not self.first else np.zeros(S, self.dtype)
        gr = g[self.rank + 1, 1:1 + S] if not self.last else np.zeros(S, self.dtype)
        return gl.copy(), gr.copy()

    # ---- protocol --------------------------------------------------------------------------------
    def init(self, send):
        S = self.S
        self.lam = np.zeros((self.k1 - self.k0, S), self.dtype)
        self.r = self.gamma[self.k0:self.k1].copy()
        self.gr_l = self.gamma[self.k0 - 1].copy() if not self.first else np.zeros(S, self.dtype)
        self.gr_r = self.gamma[self.k1].copy() if not self.last else np.zeros(S, self.dtype)
        self.p = np.zeros_like(self.r)
        self.gp_l = np.zeros(S, self.dtype)
        self.gp_r = np.zeros(S, self.dtype)
        self.rt = self._matvec(self.Pl, self.Pm, self.Pr, self.r, self.gr_l, self.gr_r)
        self._pack(send, np.sum(self.r * self.rt, dtype=self.dtype), self.rt)

    def phase_a(self, it, recvB_cur, recvB_prev, send):
        if self.is_done:
            return
        beta = self.dtype.type(0)
        if it > 0:
            eta_new = self._sum(recvB_cur)
            if abs(eta_new) < self.tol:
                self.is_done, self.iters = True, it - 1
                return
            beta = eta_new / self._sum(recvB_prev)
        gl, gr = self._halo(recvB_cur)                      # neighbours' r~ blocks
        self.p = self.rt + beta * self.p
        self.gp_l = gl + beta * self.gp_l
        self.gp_r = gr + beta * self.gp_r
        self.ups = self._matvec(self.Sl, self.Sm, self.Sr, self.p, self.gp_l, self.gp_r)
        self._pack(send, np.sum(self.p * self.ups, dtype=self.dtype), self.ups)

    def phase_b(self, it, recvB_cur, recvA, send):
        if self.is_done:
            return
        alpha = self._sum(recvB_cur) / self._sum(recvA)
        gl, gr = self._halo(recvA)                          # neighbours' upsilon blocks
        self.lam += alpha * self.p
        self.r = self.r - alpha * self.ups
        self.gr_l = self.gr_l - alpha * gl
        self.gr_r = self.gr_r - alpha * gr
        self.rt = self._matvec(self.Pl, self.Pm, self.Pr, self.r, self.gr_l, self.gr_r)
        self._pack(send, np.sum(self.r * self.rt, dtype=self.dtype), self.rt)

    def finish(self, recvB_last):
        if not self.is_done and self.max_iters > 0 and abs(self._sum(recvB_last)) < self.tol:
            self.is_done, self.iters = True, self.max_iters - 1
        full = np.zeros((self.K, self.S), self.dtype)
        full[self.k0:self.k1] = self.lam
        return torch.from_numpy(full.reshape(-1)), torch.tensor([self.iters], dtype=torch.int32)

    def done(self):
        return self.is_done
