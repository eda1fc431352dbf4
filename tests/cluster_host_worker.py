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
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{world} ok {outcomes}")


if __name__ == "__main__":
    main()
