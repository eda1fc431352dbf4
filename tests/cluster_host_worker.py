"""Worker of tests/test_dist_gloo.py::test_cluster_setup_fails_on_every_rank_together: the host-side protocol of
ClusterPCG (gato_python_amd/dist.py) on a gloo world, with the C library replaced by a stub whose mirror export fails on
ONE rank.  Every rank must raise ClusterUnavailable (and none may hang in a collective): that is what lets bench.py fall
back to the RCCL schedule on all ranks at once."""
import ctypes as ct
import os
import sys

import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gato_python_amd import _lib                                   # noqa: E402
from gato_python_amd import dist as gd                             # noqa: E402


class StubLib:
    def __init__(self, real, fail_create, fail_connect, fits):
        self.real, self.fail_create, self.fail_connect, self.fits = real, fail_create, fail_connect, fits
        self.destroyed = 0

    def gato_cluster_knot_range(self, *a):
        return self.real.gato_cluster_knot_range(*a)

    def gato_cluster_create(self, h, rank, nranks, handle):
        if self.fail_create:
            return -3
        if handle is not None:
            ct.memmove(handle, bytes([rank + 1]) * 64, 64)
        return 0

    def gato_cluster_connect(self, h, handles, ptrs):
        assert len(handles) == 64 * dist.get_world_size() and handles[64] == 2
        return -3 if self.fail_connect else 0

    def gato_cluster_fits(self, h, g, t):
        g._obj.value = 4 if self.fits else 0
        return 0

    def gato_cluster_destroy(self, h):
        self.destroyed += 1
        return 0

    def gato_cluster_local_mirror(self, h):
        return 0

    def gato_last_error(self):
        return b"stub failure"


class FakeSolver:
    K, _h = 64, None
    status_bad = False

    def synchronize(self):
        pass

    def check_status(self):
        if self.status_bad:
            raise RuntimeError("ETIMEOUT")

    def close(self):
        pass


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    real = _lib.lib()
    outcomes = []
    # (create fails on rank 1) / (connect fails on rank 0) / (rank 1's knots do not fit) / (all fine)
    for case, (fc, fn, fits) in enumerate([(rank == 1, False, True), (False, rank == 0, True), (False, False, rank != 1),
                                           (False, False, True)]):
        stub = StubLib(real, fc, fn, fits)
        _lib._LIB = stub
        try:
            cl = gd.ClusterPCG(FakeSolver(), rank, world)
            outcomes.append("ok")
            assert (cl.k0, cl.k1) == gd.knot_ranges(FakeSolver.K, world)[rank]
        except gd.ClusterUnavailable:
            outcomes.append("unavailable")
            assert stub.destroyed == 1
        finally:
            _lib._LIB = real
    assert outcomes == ["unavailable", "unavailable", "unavailable", "ok"], outcomes
    assert gd._all_ranks_ok(True) is True and gd._all_ranks_ok(rank != 0) is False

    # only rank 1's knots do not fit.  EVERY rank must see the size rejection ("do not fit" ends the search over
    # the mirror memory kinds): a rank that saw "a peer could not map the mirrors" instead would go on to the next kind and
    # sit in a collective the others never enter.
    stub = StubLib(real, False, False, rank != 1)
    _lib._LIB = stub
    created = []
    try:
        try:
            gd.ClusterPCG(FakeSolver(), rank, world)
            raise AssertionError("expected ClusterUnavailable")
        except gd.ClusterUnavailable as e:
            assert "do not fit" in str(e), str(e)
        orig = stub.gato_cluster_create

        def counting_create(*a):
            created.append(os.environ.get("GATO_XMEM"))
            return orig(*a)
        stub.gato_cluster_create = counting_create
        os.environ.pop("GATO_XMEM", None)
        try:
            gd.connect_cluster(FakeSolver(), rank, world, lambda cl: (_ for _ in ()).throw(AssertionError("no probe expected")))
            raise AssertionError("expected ClusterUnavailable")
        except gd.ClusterUnavailable as e:
            assert "do not fit" in str(e), str(e)
        assert created == ["uncached"], created          # one kind tried, on every rank alike
    finally:
        _lib._LIB = real

    # connect_cluster: mirrors in each memory kind in turn + a probe solve, every decision an AND over the ranks.
    #  kind "uncached": the probe launch raises on rank 1; "finegrained": rank 0's launch reports a time-out in band
    #  (iters = -1); "plain": every rank's launch completes -> taken, on every rank, with the two rejections recorded
    import torch
    stub = StubLib(real, False, False, True)
    _lib._LIB = stub
    seen = []

    def launch(cl):
        kind = os.environ["GATO_XMEM"]
        seen.append(kind)
        if kind == "uncached" and rank == 1:
            raise RuntimeError("launch failed here only")
        return torch.tensor([-1 if (kind == "finegrained" and rank == 0) else 7], dtype=torch.int32)
    try:
        os.environ.pop("GATO_XMEM", None)
        cl, why = gd.connect_cluster(FakeSolver(), rank, world, launch)
        assert seen == ["uncached", "finegrained", "plain"], seen
        assert why.count("timed out") + why.count("launch failed") == 2, why
        assert "GATO_XMEM" not in os.environ and cl is not None
        # a sticky status on ONE rank (no in-band mark) rejects the kind on every rank; nothing left -> ClusterUnavailable everywhere
        seen.clear()
        sol = FakeSolver()
        sol.status_bad = rank == 1
        try:
            gd.connect_cluster(sol, rank, world, launch, kinds=("plain",))
            raise AssertionError("expected ClusterUnavailable")
        except gd.ClusterUnavailable as e:
            assert "timed out" in str(e)
        # expect_iters: a launch that ends early is not a complete probe
        try:
            gd.connect_cluster(FakeSolver(), rank, world, launch, expect_iters=100, kinds=("plain",))
            raise AssertionError("expected ClusterUnavailable")
        except gd.ClusterUnavailable:
            pass
    finally:
        _lib._LIB = real
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{world} ok {outcomes}")


if __name__ == "__main__":
    main()
