"""Knot-sharded PCG across the GPUs of one node (new work - the reference is
single-device, src/gato_utils.cuh:831, and has no communication layer).

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU
tests).  Rank r owns a contiguous range of block rows of S / Pinv.  Per PCG iteration there are exactly
two collectives, each ONE all-gather of a fixed-size record per rank

      record = [ partial dot | first S-block | last S-block ]   (2S+1 scalars)

of the vector the rank just produced (upsilon after S.p, r~ after Pinv.r).  The gathered array gives
every rank the global dot (summed in rank order on the device: bitwise identical everywhere, so every
rank takes the same exit decision) and its two neighbours' boundary blocks; ghost blocks of r and p are
advanced locally (ghost_r -= alpha ghost_upsilon ; ghost_p = ghost_r~ + beta ghost_p).  Nothing on the
host reads device data inside the loop: kernels and collectives are only enqueued.

ShardedPCG is the orchestration; the arithmetic lives behind a backend object (HipShardBackend: the HIP
kernels through the C ABI, gato_shard_pcg_* in include/gato_hip.h).

ClusterPCG is the xGMI-native transport (whole solve: linsys_solve_cluster; linsys_solve_auto takes it when it can be
connected and falls back to linsys_solve_sharded otherwise): ONE persistent launch per rank
per solve, in which the exchange above happens inside the kernel - every rank stores its {epoch, payload}
granules straight into the peers' IPC-mapped mirrors (system-scope stores over xGMI) and polls only its own
(gato_cluster_* in include/gato_hip.h, pcg_resident_kernel<..., MR>; option variant = 1: pcg_cg1_kernel<..., MR>, one
exchange per iteration).  torch.distributed carries the 64-byte
IPC handles once and the barrier after connecting; nothing of it runs inside the solve.  The all-gather
schedule above (two launches + two RCCL collectives per iteration, linsys_solve_sharded) stays as the portable
fallback.  connect_cluster() is the one place that decides: mirrors in each memory kind in turn, a probe solve, every
decision an AND over the ranks, ClusterUnavailable on all ranks together when nothing works.
"""
from __future__ import annotations

import ctypes as ct

import numpy as np


def knot_ranges(K: int, nranks: int):
    """Balanced contiguous ranges [k0,k1) in rank order; every rank owns at least one knot."""
    if nranks > K:
        raise ValueError(f"cannot shard {K} knots over {nranks} ranks")
    base, extra = divmod(K, nranks)
    out, k = [], 0
    for r in range(nranks):
        n = base + (1 if r < extra else 0)
        out.append((k, k + n))
        k += n
    return out


class ShardedPCG:
    """Collective schedule of the sharded PCG.  backend must provide
         new_record() / new_gathered()                      buffers for all_gather_into_tensor
         init(send) ; phase_a(it, recvB_cur, recvB_prev, send) ; phase_b(it, recvB_cur, recvA, send)
         finish(recvB_last) -> (lambda_full_masked, iters)  tensors
         done() -> bool                                     (only used when check_every > 0)
    """

    def __init__(self, backend, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.backend = backend
        self.group = group

    def _gather(self, out, send):
        if getattr(send, "is_cuda", False) and self.dist.get_backend(self.group) != "nccl":
            ho, hs = out.cpu(), send.cpu()                  # one-GPU rehearsals over gloo: through host memory
            self.dist.all_gather_into_tensor(ho, hs, group=self.group)
            out.copy_(ho)
            return
        self.dist.all_gather_into_tensor(out, send, group=self.group)

    def solve(self, max_iters: int, check_every: int = 0):
        b = self.backend
        send = b.new_record()
        recvB = [b.new_gathered() for _ in range(3)]
        recvA = b.new_gathered()
        b.init(send)
        self._gather(recvB[0], send)
        for it in range(max_iters):
            b.phase_a(it, recvB[it % 3], recvB[(it + 2) % 3], send)
            self._gather(recvA, send)
            b.phase_b(it, recvB[it % 3], recvA, send)
            self._gather(recvB[(it + 1) % 3], send)
            if check_every and (it + 1) % check_every == 0 and b.done():
                break          # every rank sees the same flag: the sums are identical on all ranks
        lam, iters = b.finish(recvB[max_iters % 3])
        if getattr(lam, "is_cuda", False):
            allreduce_sum_(lam, self.group)                                          # slices are disjoint, rest is 0
        else:
            self.dist.all_reduce(lam, op=self.dist.ReduceOp.SUM, group=self.group)
        return lam, iters


def run_lockstep(backends, max_iters: int):
    """The same schedule for a list of backends living in ONE process (rank i = backends[i]): the
    all-gather becomes a concatenation.  Used to exercise the shard kernels on a single GPU."""
    import torch
    n = len(backends)
    send = [b.new_record() for b in backends]
    cat = lambda: torch.cat(send)
    backends[0].torch = torch
    for b, s_ in zip(backends, send):
        b.init(s_)
    recvB = [None, None, None]
    recvB[0] = cat()
    recvA = None
    for it in range(max_iters):
        prev = recvB[(it + 2) % 3] if recvB[(it + 2) % 3] is not None else recvB[it % 3]
        for b, s_ in zip(backends, send):
            b.phase_a(it, recvB[it % 3], prev, s_)
        recvA = cat()
        for b, s_ in zip(backends, send):
            b.phase_b(it, recvB[it % 3], recvA, s_)
        recvB[(it + 1) % 3] = cat()
    outs = [b.finish(recvB[max_iters % 3]) for b in backends]
    lam = outs[0][0].clone()
    for o_ in outs[1:]:
        lam += o_[0]
    return lam, [o_[1] for o_ in outs]


class HipShardBackend:
    """The HIP kernels behind ShardedPCG: streaming PCG step on this rank's shard (C ABI)."""

    def __init__(self, solver, rank, nranks, d_S, d_Pinv, d_gamma, exit_tol, max_iters):
        import torch
        from . import _lib
        self.torch, self._lib, self.sol = torch, _lib, solver
        self.rank, self.nranks = rank, nranks
        self.k0, self.k1 = knot_ranges(solver.K, nranks)[rank]
        self.S_bd, self.P_bd, self.gamma = d_S, d_Pinv, d_gamma
        self.exit_tol, self.max_iters = float(exit_tol), int(max_iters)
        self.rec = 2 * solver.S + 1

    def _st(self):
        return self.sol._stream()

    def new_record(self):
        return self.sol.new(self.rec)

    def new_gathered(self):
        return self.sol.new(self.rec * self.nranks)

    @staticmethod
    def _p(t):
        return ct.c_void_p(t.data_ptr())

    def init(self, send):
        L = self._lib.lib()
        self._lib.check(L.gato_shard_pcg_init(self.sol._h, self.rank, self.nranks, self.k0, self.k1, self._p(self.S_bd),
                                              self._p(self.P_bd), self._p(self.gamma), self.exit_tol,
                                              self.max_iters, self._p(send), self._st()))

    def phase_a(self, it, recvB_cur, recvB_prev, send):
        self._lib.check(self._lib.lib().gato_shard_pcg_phase_a(self.sol._h, it, self._p(recvB_cur), self._p(recvB_prev),
                                                               self._p(send), self._st()))

    def phase_b(self, it, recvB_cur, recvA, send):
        self._lib.check(self._lib.lib().gato_shard_pcg_phase_b(self.sol._h, it, self._p(recvB_cur), self._p(recvA),
                                                               self._p(send), self._st()))

    def finish(self, recvB_last):
        lam = self.sol.new(self.sol.S * self.sol.K)
        iters = self.sol.new(1, self.torch.int32)
        self._lib.check(self._lib.lib().gato_shard_pcg_finish(self.sol._h, self._p(recvB_last), self._p(lam),
                                                              self._p(iters), self._st()))
        return lam, iters

    def done(self):
        v = ct.c_int(0)
        self._lib.check(self._lib.lib().gato_shard_pcg_done(self.sol._h, ct.byref(v), self._st()))
        return v.value != 0


def linsys_solve_sharded(sysm, exit_tol, max_iters, dtype=np.float32, device=None, group=None, sol=None):
    """Whole solve with the PCG sharded over the ranks of `group`.  Assembly (CSR scatter, Schur, stair)
    is replicated on every rank from the replicated CSR inputs - it is a one-off O(K) step and replication
    needs no exchange (SURVEY.md section 8e lists the halo terms it would otherwise need) - then the PCG
    runs on each rank's knot range and lambda is assembled by a sum-all-reduce of the disjoint slices; dz
    is computed redundantly on every rank.  Returns (lambda, dz, iters) as device tensors."""
    import torch
    import torch.distributed as dist
    from .solver import Solver
    rank, nranks = dist.get_rank(group), dist.get_world_size(group)
    dev = torch.cuda.current_device() if device is None else device
    if sol is None:                 # (pass the solver of an earlier call back in: creating one runs the XCD calibration trials)
        sol = Solver(sysm.S, sysm.C, sysm.K, dtype, dev)
    d = sol.upload_system(sysm)
    Gd, Cd = sol.convert(*d[:6], sysm.rho)
    Sb, Pb, gam, Gi = sol.form_schur(Gd, Cd, d[6], d[7])
    sol.form_ss(Sb, Pb)
    backend = HipShardBackend(sol, rank, nranks, Sb, Pb, gam, exit_tol, max_iters)
    lam, iters = ShardedPCG(backend, group).solve(max_iters)
    dz = sol.compute_dz(Gi, Cd, d[6], lam)
    return lam, dz, iters, sol


class ClusterUnavailable(RuntimeError):
    """Raised on EVERY rank alike when some rank could not allocate, export or map a mirror (callers fall back to the
    all-gather schedule over RCCL)."""


def _all_ranks_ok(ok: bool, group=None) -> bool:
    """Logical AND over the ranks (and a barrier), on whatever device the backend of `group` moves."""
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.cpu()[0]))


def _all_ranks_min(flags, group=None):
    """Element-wise minimum of a few 0/1 flags over the ranks (one collective; also a barrier)."""
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([1 if f else 0 for f in flags], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return [bool(int(v)) for v in t.cpu()]


def allreduce_sum_(t, group=None):
    """In-place sum of a device tensor over the ranks (through host memory when the backend is gloo)."""
    import torch.distributed as dist
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    else:
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    return t


class ClusterPCG:
    """One rank of the in-kernel cross-GPU PCG (gato_cluster_*).  Create once per (solver, group), then call
    pcg() for every solve: all ranks must issue the same sequence of calls (the epoch counters run in lock-step)."""

    def __init__(self, solver, rank, nranks, group=None, inprocess_peers=None):
        from . import _lib
        self._lib, self.sol = _lib, solver
        self.rank, self.nranks, self.group = rank, nranks, group
        L = _lib.lib()
        k0, k1 = ct.c_int(), ct.c_int()
        _lib.check(L.gato_cluster_knot_range(solver.K, rank, nranks, ct.byref(k0), ct.byref(k1)))
        self.k0, self.k1 = k0.value, k1.value
        self._handle = (ct.c_char * 64)()
        self._inprocess = inprocess_peers is not None
        self._peers, self.rewinds = None, 0          # ranks of one process (connect_inprocess); epoch-space renewals so far
        if self._inprocess:
            _lib.check(L.gato_cluster_create(solver._h, rank, nranks, None))
        else:
            self._connect_ipc()
        self.mirror = int(L.gato_cluster_local_mirror(solver._h))

    def _connect_ipc(self):
        """Collective over the group.  Every step that can fail on one rank only is followed by an AND over the ranks,
        so that all ranks raise ClusterUnavailable together instead of leaving the others in a collective."""
        import torch.distributed as dist
        L, err = self._lib.lib(), ""
        try:
            self._lib.check(L.gato_cluster_create(self.sol._h, self.rank, self.nranks, self._handle))
        except Exception as e:        # noqa: BLE001
            err = f"create: {e}"
        handles = [None] * self.nranks
        dist.all_gather_object(handles, bytes(self._handle.raw) if not err else b"", group=self.group)
        if not err and all(len(h) == 64 for h in handles):
            try:
                self._lib.check(L.gato_cluster_connect(self.sol._h, b"".join(handles), None))
            except Exception as e:    # noqa: BLE001
                err = f"connect: {e}"
        elif not err:
            err = "a peer could not export its mirror"
        fits = True
        if not err:                   # every rank's knots must fit a persistent launch on its GPU
            g = ct.c_int()
            L.gato_cluster_fits(self.sol._h, ct.byref(g), None)
            fits = g.value != 0
        # AND + barrier: every mirror is zeroed and mapped before anyone launches.  The size verdict travels with it: shards
        # differ by a knot and GPUs may differ in CUs, so "do not fit" may be true on ONE rank only - and it ends the search over
        # the memory kinds (first_working_kind), which every rank must leave together or the next collectives mismatch.
        ok_all, fits_all = _all_ranks_min([not err and fits, fits], self.group)
        if not ok_all:
            L.gato_cluster_destroy(self.sol._h)
            if not fits_all:
                raise ClusterUnavailable(f"{self.sol.K} knots over {self.nranks} ranks do not fit one persistent launch per GPU"
                                         + ("" if not fits else " (on another rank)"))
            raise ClusterUnavailable(err or "a peer could not map the mirrors")

    @staticmethod
    def connect_inprocess(clusters):
        """Ranks living in ONE process (tests on one GPU, or a single-process multi-GPU host): plain device pointers."""
        n = len(clusters)
        arr = (ct.c_void_p * n)(*[c.mirror for c in clusters])
        for c in clusters:
            c._lib.check(c._lib.lib().gato_cluster_connect(c.sol._h, None, arr))
            c._peers = list(clusters)

    def launches_left(self, max_iters):
        """Launches with this max_iters that still fit the cluster's 32-bit epoch space (the same number on every rank)."""
        left = ct.c_longlong()
        self._lib.check(self._lib.lib().gato_cluster_launches_left(self.sol._h, int(max_iters), ct.byref(left)))
        return left.value

    def _renew_epochs_if_used_up(self, max_iters):
        """The hand-off epochs only grow; after about ten million 200-iteration solves the space is used up - on every rank at the
        same solve, the counters run in lock-step.  Then: wait for the own launches, barrier (nobody stores into a mirror any
        more), gato_cluster_rewind (mirror and slots zeroed, counters back to 0), barrier, go on.  Ranks living in one process are
        rewound together by the first one that notices."""
        if self.launches_left(max_iters) > 0:
            return
        import torch
        L = self._lib.lib()
        if self._inprocess:
            torch.cuda.synchronize()
            for c in (self._peers or [self]):
                c._lib.check(L.gato_cluster_rewind(c.sol._h))
                c.rewinds += 1
            return
        import torch.distributed as dist
        torch.cuda.synchronize(self.sol.device)
        dist.barrier(group=self.group)
        self._lib.check(L.gato_cluster_rewind(self.sol._h))
        dist.barrier(group=self.group)
        self.rewinds += 1

    def pcg(self, Sb, Pb, gamma, exit_tol, max_iters, lam, iters, stream=None):
        """Enqueue this rank's launch.  lam: full-length S*K buffer, this rank's slice is written."""
        self._renew_epochs_if_used_up(max_iters)
        st = self.sol._stream() if stream is None else ct.c_void_p(stream)
        p = lambda t: ct.c_void_p(t.data_ptr())
        self._lib.check(self._lib.lib().gato_cluster_pcg(self.sol._h, p(Sb), p(Pb), p(gamma), p(lam), float(exit_tol),
                                                         int(max_iters), p(iters), st))

    def linsys(self, d, exit_tol, max_iters, rho, lam, dz, iters, stream=None):
        """Enqueue this rank's part of a WHOLE solve (gato_cluster_linsys): the stage kernels on the knots its shard reads, its
        persistent launch, dz on its range - one call, nothing on the host or in a collective in between.  d: the device inputs
        of Solver.upload_system (replicated on every rank); lam / dz: full-length buffers, this rank's rows are written."""
        self._renew_epochs_if_used_up(max_iters)
        st = self.sol._stream() if stream is None else ct.c_void_p(stream)
        p = lambda t: ct.c_void_p(t.data_ptr())
        self._lib.check(self._lib.lib().gato_cluster_linsys(self.sol._h, p(d[0]), p(d[1]), p(d[2]), p(d[3]), p(d[4]), p(d[5]), p(d[6]),
                                                            p(d[7]), float(exit_tol), int(max_iters), float(rho), p(lam), p(dz),
                                                            p(iters), st))

    def close(self):
        if self.sol is not None and self.sol._h:
            self._lib.lib().gato_cluster_destroy(self.sol._h)
        self.sol = None


_LOCKSTEP_STREAMS = []


def lockstep_streams(R):
    """R streams of this process, always the same ones.  Kernels of one process overlap only if their streams sit on
    different hardware queues; HIP has GPU_MAX_HW_QUEUES of them per process (default 4: more than 3 in-process ranks
    need it raised before HIP starts - tests/conftest.py does; one process per GPU, the product configuration, does
    not care)."""
    import torch
    while len(_LOCKSTEP_STREAMS) < R:
        _LOCKSTEP_STREAMS.append(torch.cuda.Stream())
    return _LOCKSTEP_STREAMS[:R]


def run_cluster_lockstep(solvers, Sb, Pb, gamma, exit_tol, max_iters):
    """R ranks of a cluster solve in ONE process on one GPU, each on its own stream (the launches must overlap: they
    wait for each other on the device).  Returns (lambda, [iters per rank])."""
    import torch
    R = len(solvers)
    cl = [ClusterPCG(s_, r, R, inprocess_peers=True) for r, s_ in enumerate(solvers)]
    ClusterPCG.connect_inprocess(cl)
    torch.cuda.synchronize()
    dev = Sb.device
    lam = torch.zeros(solvers[0].S * solvers[0].K, dtype=Sb.dtype, device=dev)
    its = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(R)]
    streams = lockstep_streams(R)
    torch.cuda.synchronize()
    for r in range(R):
        cl[r].pcg(Sb, Pb, gamma, exit_tol, max_iters, lam, its[r], stream=streams[r].cuda_stream)
    torch.cuda.synchronize()
    out = [int(i.cpu()[0]) for i in its]
    run_cluster_lockstep.last_flat = solvers[0].get_option("last_cluster_flat")     # which exchange ran (closing resets it)
    for c in cl:
        c.close()
    return lam, out


MIRROR_KINDS = ("uncached", "finegrained", "plain")


def first_working_kind(kinds, attempt):
    """The mirrors of the in-kernel transport in each memory kind the library knows, in turn (uncached device memory, then
    fine-grained, then plain hipMalloc; GATO_XMEM pins one): attempt(kind) -> (cluster, "") when every rank could export
    and map the mirrors AND the first solve came back complete on every rank, else (None, reason).  Returns the first
    cluster that works (or None) and what was tried before it.  A rejection for size ("do not fit") ends the search."""
    tried = []
    for kind in kinds:
        c, err = attempt(kind)
        if c is not None:
            return c, "; ".join(tried)[:300]
        tried.append(f"{kind}: {err}"[:120])
        if "do not fit" in err:
            break
    return None, "; ".join(tried)[:300]


def probe_cluster(cl, launch, expect_iters=None, group=None):
    """One probe solve of a freshly connected cluster: launch(cl) enqueues this rank's persistent launch (NO collective
    inside) and returns its `iters` tensor.  Whatever happens on this rank - an exception, a hand-off time-out (iters < 0,
    sticky status) - the device is synchronised before the ranks meet, and every rank reaches the same AND.  Returns
    (ok on ALL ranks, this rank's reason)."""
    ok, err = True, ""
    try:
        it = launch(cl)
        cl.sol.synchronize()
        got = int(it.cpu()[0])
        ok = got >= 0 if expect_iters is None else got == expect_iters
        try:
            cl.sol.check_status()
        except Exception:         # noqa: BLE001
            ok = False
        if not ok:
            err = "the first in-kernel exchange timed out"
    except Exception as e:        # noqa: BLE001
        ok, err = False, f"{type(e).__name__}: {e}"
        try:
            cl.sol.synchronize()          # the peers free / unmap their mirrors next: nothing of ours may still be storing into them
        except Exception:         # noqa: BLE001
            pass
    if _all_ranks_ok(ok, group):
        return True, ""
    return False, err or "the first in-kernel exchange timed out on another rank"


def connect_cluster(sol, rank, nranks, launch, group=None, expect_iters=None, kinds=None):
    """Connected, probed ClusterPCG for this solver or ClusterUnavailable on EVERY rank (callers then take the RCCL
    schedule).  Tries the mirror memory kinds in turn (env GATO_XMEM pins one).  Returns (cluster, what was rejected before)."""
    import os
    pinned = os.environ.get("GATO_XMEM")

    def attempt(kind):
        os.environ["GATO_XMEM"] = kind
        try:
            cl = ClusterPCG(sol, rank, nranks, group)
        except ClusterUnavailable as e:
            return None, f"mirrors unavailable: {e}"
        ok, err = probe_cluster(cl, launch, expect_iters, group)
        if ok:
            return cl, ""
        cl.close()
        return None, err

    try:
        cl, why = first_working_kind([pinned] if pinned else list(kinds or MIRROR_KINDS), attempt)
    finally:
        if not pinned:
            os.environ.pop("GATO_XMEM", None)
    if cl is None:
        raise ClusterUnavailable(why or "no mirror memory kind worked")
    return cl, why


class ClusterTimeout(ClusterUnavailable):
    """A hand-off of a cluster solve timed out on some rank (raised on every rank alike; the outputs are not valid)."""


class _GatherPlan:
    """lambda and dz of a sharded solve onto every rank with ONE all-gather of a fixed-size record per rank

          record = [ lambda rows of the rank's knots | dz rows of the rank's knots ]   (padded to the largest range)

    into buffers allocated once (no per-solve allocation, no zero-padded full-length sums): two slice copies into the send
    record, the collective, two index gathers out of the received records."""

    def __init__(self, sol, nranks, rank, device):
        import torch
        S, n, K = sol.S, sol.n, sol.K
        self.ranges = knot_ranges(K, nranks)
        mk = max(k1 - k0 for k0, k1 in self.ranges)
        self.nl, self.nd = mk * S, mk * n
        self.rec = self.nl + self.nd
        self.k0, self.k1 = self.ranges[rank]
        self.send = torch.zeros(self.rec, dtype=sol.dtype, device=device)
        self.recv = torch.zeros(self.rec * nranks, dtype=sol.dtype, device=device)
        il, iz = np.empty(S * K, np.int64), np.empty(sol.N, np.int64)
        for r, (k0, k1) in enumerate(self.ranges):
            il[k0 * S:k1 * S] = r * self.rec + np.arange((k1 - k0) * S)
            hi = min(k1 * n, sol.N)
            iz[k0 * n:hi] = r * self.rec + self.nl + np.arange(hi - k0 * n)
        self.il, self.iz = torch.from_numpy(il).to(device), torch.from_numpy(iz).to(device)
        self.lam_out = torch.empty(S * K, dtype=sol.dtype, device=device)
        self.dz_out = torch.empty(sol.N, dtype=sol.dtype, device=device)
        self.S, self.n, self.N = S, n, sol.N

    def gather(self, lam, dz, group=None):
        import torch
        import torch.distributed as dist
        S, n = self.S, self.n
        hi = min(self.k1 * n, self.N)
        self.send[:(self.k1 - self.k0) * S].copy_(lam[self.k0 * S:self.k1 * S])
        self.send[self.nl:self.nl + hi - self.k0 * n].copy_(dz[self.k0 * n:hi])
        if dist.get_backend(group) == "nccl":
            dist.all_gather_into_tensor(self.recv, self.send, group=group)
        else:                                             # one-GPU rehearsals over gloo: through host memory
            ho, hs = self.recv.cpu(), self.send.cpu()
            dist.all_gather_into_tensor(ho, hs, group=group)
            self.recv.copy_(ho)
        torch.index_select(self.recv, 0, self.il, out=self.lam_out)
        torch.index_select(self.recv, 0, self.iz, out=self.dz_out)
        return self.lam_out, self.dz_out


def close_state(state):
    """Frees what a state dict of linsys_solve_cluster / linsys_solve_auto owns (mirrors, solver arenas)."""
    if not state:
        return
    for key in ("cl", "sol", "rccl_sol"):
        obj = state.pop(key, None)
        try:
            if obj is not None:
                obj.close()
        except Exception:     # noqa: BLE001
            pass


def linsys_solve_cluster(sysm, exit_tol, max_iters, dtype=np.float32, device=None, group=None, state=None, check=True,
                         gather=True, variant=0, solver_options=None):
    """Whole solve with the PCG sharded over the ranks of `group` through the in-kernel xGMI hand-off.  Per solve every rank
    makes ONE library call (ClusterPCG.linsys -> gato_cluster_linsys: sharded assembly, its persistent launch, dz on its knot
    range - lambda_{k1}, the one block dz needs from the neighbouring rank, arrives inside the launch) and, with gather (default),
    one all-gather of [lambda rows | dz rows] records into buffers allocated at the first call (_GatherPlan); gather=False leaves
    lambda and dz sharded (full-length buffers of which this rank's rows are valid).  variant = 1: the single-reduction
    recurrence - ONE cross-GPU exchange per iteration instead of two (opt-in as on one GPU: rounding differs from the
    reference recurrence).  The first call connects the cluster (connect_cluster: memory kinds in turn, a probe solve,
    ClusterUnavailable on every rank when the transport cannot be used).  With check (default) the call synchronises at its end
    and raises ClusterTimeout on EVERY rank if any rank's launch reported a hand-off time-out (iters = -1 / sticky status): the
    outputs are garbage then and the caller takes linsys_solve_sharded.  `state` (returned as the last element) carries solver,
    device inputs, output buffers and the connected cluster across calls; a later call with ANOTHER system of the same shape
    copies its values into the same device buffers.  close_state(state) frees it.  solver_options: {option: value} set on
    the solver before the cluster is connected (the same on every rank)."""
    import torch
    import torch.distributed as dist
    from .solver import Solver
    rank, nranks = dist.get_rank(group), dist.get_world_size(group)
    created_here = state is None
    if state is None:
        dev = torch.cuda.current_device() if device is None else device
        sol = Solver(sysm.S, sysm.C, sysm.K, dtype, dev)
        sol.set_option("pcg_variant", int(variant))
        for name, value in (solver_options or {}).items():      # before connecting: every rank plans with the same options
            sol.set_option(name, value)
        tdev = f"cuda:{sol.device}"
        state = dict(sol=sol, d=sol.upload_system(sysm), sysm=sysm, variant=int(variant),
                     lam=torch.zeros(sol.S * sol.K, dtype=sol.dtype, device=tdev),
                     dz=torch.zeros(sol.N, dtype=sol.dtype, device=tdev),
                     iters=torch.zeros(1, dtype=torch.int32, device=tdev))

        def probe(cl):
            cl.linsys(state["d"], exit_tol, max_iters, sysm.rho, state["lam"], state["dz"], state["iters"])
            return state["iters"]
        try:
            state["cl"], state["rejected"] = connect_cluster(sol, rank, nranks, probe, group)
        except ClusterUnavailable:
            sol.close()
            raise
        state["plan"] = _GatherPlan(sol, nranks, rank, tdev)
    sol, d, cl = state["sol"], state["d"], state["cl"]
    if (sysm.S, sysm.C, sysm.K) != (sol.S, sol.C, sol.K) or np.dtype(dtype) != sol.np_dtype:
        raise ValueError("linsys_solve_cluster: the state belongs to another shape / dtype; close_state() it and start anew")
    if int(variant) != state["variant"]:
        sol.set_option("pcg_variant", int(variant))
        state["variant"] = int(variant)
    if sysm is not state["sysm"]:                      # new values, same shape: into the same device buffers
        for t, a in zip(d, (sysm.G_row, sysm.G_col, sysm.G_val, sysm.C_row, sysm.C_col, sysm.C_val, sysm.g, sysm.c)):
            if t.numel() != np.asarray(a).size:
                raise ValueError("linsys_solve_cluster: the sparsity pattern changed; close_state() the state and start anew")
            t.copy_(torch.from_numpy(np.ascontiguousarray(a, {torch.int32: np.int32}.get(t.dtype, sol.np_dtype))))
        state["sysm"] = sysm
    lam, dz, iters = state["lam"], state["dz"], state["iters"]
    cl.linsys(d, exit_tol, max_iters, sysm.rho, lam, dz, iters)
    if gather:
        lam, dz = state["plan"].gather(lam, dz, group)
    if check:
        sol.synchronize()
        ok = int(iters.cpu()[0]) >= 0
        try:
            sol.check_status()
        except Exception:         # noqa: BLE001
            ok = False
        if not _all_ranks_ok(ok, group):
            if created_here:        # the caller never saw this state: nobody else can free the mirrors and the arena
                close_state(state)
            raise ClusterTimeout("a hand-off of the cluster solve timed out" + ("" if ok else " on this rank"))
    return lam, dz, iters, state


MAX_CONSECUTIVE_TIMEOUTS = 3


def linsys_solve_auto(sysm, exit_tol, max_iters, dtype=np.float32, device=None, group=None, state=None, variant=0):
    """The product entry for a knot-sharded solve: the in-kernel transport when it can be connected and its solves come
    back complete, else (every rank together) the RCCL all-gather schedule.  Returns (lambda, dz, iters, state);
    state["transport"] says what the state runs on, state["last_transport"] what THIS solve ran on; close_state(state) frees what
    it holds.  A cluster that has worked and then reports a hand-off time-out (a rank's launch came later than timeout_ms: a
    stalled process, a busy GPU) is not given up at once: that solve is repeated over RCCL, the next one tries the cluster again -
    its epochs only grow, a late launch leaves nothing a later one could take for its own - and only MAX_CONSECUTIVE_TIMEOUTS
    time-outs in a row drop it for good."""
    if state is None or state.get("transport") == "xgmi":
        try:
            lam, dz, iters, st = linsys_solve_cluster(sysm, exit_tol, max_iters, dtype, device, group, state, variant=variant)
            st["transport"] = st["last_transport"] = "xgmi"
            st["timeouts"] = 0
            return lam, dz, iters, st
        except ClusterUnavailable as e:
            again = (isinstance(e, ClusterTimeout) and state is not None and state.get("cl") is not None and
                     state.get("timeouts", 0) + 1 < MAX_CONSECUTIVE_TIMEOUTS)
            if again:                                # every rank alike: ClusterTimeout is raised on all of them together
                state["timeouts"] = state.get("timeouts", 0) + 1
                state["why"] = str(e)[:300]
            else:
                close_state(state)                   # never connected, or timed out too often: keep going over RCCL
                state = dict(transport="rccl", why=str(e)[:300])
    # the RCCL schedule keeps its solver across calls (state["rccl_sol"]): a new one per call would run the calibration trials
    # again - but only for the shape, type and device it was made for
    sol = state.get("rccl_sol")
    if sol is not None and ((sol.S, sol.C, sol.K) != (sysm.S, sysm.C, sysm.K) or sol.np_dtype != np.dtype(dtype) or
                            (device is not None and sol.device != int(device))):
        sol.close()
        sol = state["rccl_sol"] = None
    lam, dz, iters, sol = linsys_solve_sharded(sysm, exit_tol, max_iters, dtype, device, group, sol=sol)
    state["rccl_sol"] = sol
    state["last_transport"] = "rccl"
    return lam, dz, iters, state
