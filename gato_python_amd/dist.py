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
(gato_cluster_* in include/gato_hip.h, pcg_resident_kernel<..., MR>).  torch.distributed carries the 64-byte
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

    def pcg(self, Sb, Pb, gamma, exit_tol, max_iters, lam, iters, stream=None):
        """Enqueue this rank's launch.  lam: full-length S*K buffer, this rank's slice is written."""
        st = self.sol._stream() if stream is None else ct.c_void_p(stream)
        p = lambda t: ct.c_void_p(t.data_ptr())
        self._lib.check(self._lib.lib().gato_cluster_pcg(self.sol._h, p(Sb), p(Pb), p(gamma), p(lam), float(exit_tol),
                                                         int(max_iters), p(iters), st))

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


def assemble_shard(sol, d, rho, k0, k1, out=None):
    """Stage kernels restricted to what the PCG shard [k0, k1) of this rank reads (full-size buffers, only these rows
    are written): S / Pinv rows k0..k1-1 complete (S[k].right comes from the Schur step of knot k+1, the stair blocks
    need theta^-1 of both neighbours) and gamma on k0-1..k1 (ghosts of the initial residual).  Hence
    gather + inversions on [k0-2, k1+1), Schur on [k0-1, k1+1), stair on [k0, k1).  Returns (Gd, Cd, Sb, Pb, gamma, Ginv)."""
    K = sol.K
    clip = lambda a: max(0, min(K, a))

    def rng(lo, hi):
        sol.set_option("knot_lo", clip(lo))
        sol.set_option("knot_hi", clip(hi))
    if out is None:
        import torch
        z = lambda n: torch.zeros(int(max(n, 1)), dtype=sol.dtype, device=f"cuda:{sol.device}")
        out = dict(Gd=z(sol.sizes["G_dense"]), Cd=z(sol.sizes["C_dense"]), Sb=z(sol.sizes["bd"]), Pb=z(sol.sizes["bd"]),
                   gam=z(sol.sizes["sk"]), Gi=z(sol.sizes["G_dense"]))
    from . import _lib
    from .solver import _ptr
    L = _lib.lib()
    rng(k0 - 2, k1 + 1)
    _lib.check(L.gato_convert(sol._h, _ptr(d[0]), _ptr(d[1]), _ptr(d[2]), _ptr(d[3]), _ptr(d[4]), _ptr(d[5]), float(rho),
                              _ptr(out["Gd"]), _ptr(out["Cd"]), sol._stream()))
    # gato_form_schur inverts the Q_k, R_k of its knot range first: the range must cover knot k-1 of every Schur step, so
    # the inversions run on [k0-2, k1+1) and the Schur steps on [k0-1, k1+1) (their extra first knot only rewrites the
    # rows k0-2 of S / Pinv, which nobody reads)
    _lib.check(L.gato_form_schur(sol._h, _ptr(out["Gd"]), _ptr(out["Cd"]), _ptr(d[6]), _ptr(d[7]), _ptr(out["Sb"]), _ptr(out["Pb"]),
                                 _ptr(out["gam"]), _ptr(out["Gi"]), sol._stream()))
    rng(k0, k1)
    _lib.check(L.gato_form_ss(sol._h, _ptr(out["Sb"]), _ptr(out["Pb"]), sol._stream()))
    rng(0, 0)
    return out


def dz_shard(sol, d, bufs, lam, dz, k0, k1):
    """dz rows of the knots [k0, k1) into the full-size buffer dz (lam: the assembled lambda - knot k needs lambda_{k+1})."""
    from . import _lib
    from .solver import _ptr
    sol.set_option("knot_lo", k0)
    sol.set_option("knot_hi", k1)
    _lib.check(_lib.lib().gato_compute_dz(sol._h, _ptr(bufs["Gi"]), _ptr(bufs["Cd"]), _ptr(d[6]), _ptr(lam), _ptr(dz), sol._stream()))
    sol.set_option("knot_lo", 0)
    sol.set_option("knot_hi", 0)
    return dz


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


def linsys_solve_cluster(sysm, exit_tol, max_iters, dtype=np.float32, device=None, group=None, state=None, check=True):
    """Whole solve with the PCG sharded over the ranks of `group` through the in-kernel xGMI hand-off.  Assembly is SHARDED
    too (assemble_shard: every rank forms the block rows its PCG shard reads plus the few halo knots they depend on), each
    rank's launch solves its knot range, lambda and dz are assembled by one sum-all-reduce of the disjoint slices each per
    solve (outside the iteration loop).  The first call connects the cluster (connect_cluster: memory kinds in turn, a
    probe solve, ClusterUnavailable on every rank when the transport cannot be used).  With check (default) the call
    synchronises at its end and raises ClusterTimeout on EVERY rank if any rank's launch reported a hand-off time-out
    (iters = -1 / sticky status): the outputs are garbage then and the caller takes linsys_solve_sharded.
    `state` (returned as the last element) carries solver, device inputs and the connected cluster across calls."""
    import torch
    import torch.distributed as dist
    from .solver import Solver
    rank, nranks = dist.get_rank(group), dist.get_world_size(group)
    if state is None:
        dev = torch.cuda.current_device() if device is None else device
        sol = Solver(sysm.S, sysm.C, sysm.K, dtype, dev)
        state = dict(sol=sol, d=sol.upload_system(sysm))

        def probe(cl):
            b = state["bufs"] = assemble_shard(sol, state["d"], sysm.rho, cl.k0, cl.k1, state.get("bufs"))
            lam = torch.zeros(sol.S * sol.K, dtype=sol.dtype, device=b["Sb"].device)
            it = torch.zeros(1, dtype=torch.int32, device=b["Sb"].device)
            cl.pcg(b["Sb"], b["Pb"], b["gam"], exit_tol, max_iters, lam, it)
            return it
        try:
            state["cl"], state["rejected"] = connect_cluster(sol, rank, nranks, probe, group)
        except ClusterUnavailable:
            sol.close()
            raise
    created_here = "transport" not in state and state.get("_fresh", True)
    state["_fresh"] = False
    sol, d, cl = state["sol"], state["d"], state["cl"]
    b = state["bufs"] = assemble_shard(sol, d, sysm.rho, cl.k0, cl.k1, state.get("bufs"))
    dev = b["Sb"].device
    lam = torch.zeros(sol.S * sol.K, dtype=sol.dtype, device=dev)
    dz = torch.zeros(sol.N, dtype=sol.dtype, device=dev)
    iters = torch.zeros(1, dtype=torch.int32, device=dev)
    cl.pcg(b["Sb"], b["Pb"], b["gam"], exit_tol, max_iters, lam, iters)
    allreduce_sum_(lam, group)                      # disjoint slices -> the whole lambda on every rank (once per solve)
    dz_shard(sol, d, b, lam, dz, cl.k0, cl.k1)
    allreduce_sum_(dz, group)
    if check:
        sol.synchronize()
        ok = int(iters.cpu()[0]) >= 0
        try:
            sol.check_status()
        except Exception:         # noqa: BLE001
            ok = False
        if not _all_ranks_ok(ok, group):
            if created_here:        # the caller never saw this state: nobody else can free the mirrors and the arena
                try:
                    cl.close()
                    sol.close()
                except Exception:     # noqa: BLE001
                    pass
            raise ClusterTimeout("a hand-off of the cluster solve timed out" + ("" if ok else " on this rank"))
    return lam, dz, iters, state


def linsys_solve_auto(sysm, exit_tol, max_iters, dtype=np.float32, device=None, group=None, state=None):
    """The product entry for a knot-sharded solve: the in-kernel transport when it can be connected and its solves come
    back complete, else (every rank together) the RCCL all-gather schedule.  Returns (lambda, dz, iters, state);
    state["transport"] says what ran."""
    if state is None or state.get("transport") == "xgmi":
        try:
            lam, dz, iters, st = linsys_solve_cluster(sysm, exit_tol, max_iters, dtype, device, group, state)
            st["transport"] = "xgmi"
            return lam, dz, iters, st
        except ClusterUnavailable as e:
            if state is not None:                   # a connected cluster timed out later: drop it, keep going over RCCL
                try:
                    state["cl"].close()
                    state["sol"].close()
                except Exception:     # noqa: BLE001
                    pass
            state = dict(transport="rccl", why=str(e)[:300])
    # the RCCL schedule keeps its solver across calls (state["rccl_sol"]): a new one per call would run the calibration trials again
    lam, dz, iters, sol = linsys_solve_sharded(sysm, exit_tol, max_iters, dtype, device, group, sol=state.get("rccl_sol"))
    state["rccl_sol"] = sol
    return lam, dz, iters, state
