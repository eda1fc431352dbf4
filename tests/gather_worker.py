"""Worker of tests/test_dist_gloo.py::test_gather_plan_gloo: one rank of a gloo world putting its knot range's lambda and dz rows
through dist._GatherPlan (ONE all-gather of fixed-size records into buffers allocated once) - CPU tensors, no GPU, no library."""
import os
import sys
import types

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gato_python_amd.dist import _GatherPlan, knot_ranges   # noqa: E402


def main():
    S, C, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n, N = S + C, (S + C) * K - C
    sol = types.SimpleNamespace(S=S, C=C, K=K, n=n, N=N, dtype=torch.float64)        # what the plan reads of a Solver
    plan = _GatherPlan(sol, world, rank, "cpu")
    k0, k1 = knot_ranges(K, world)[rank]
    want_l, want_z = np.arange(S * K, dtype=np.float64) + 0.5, -(np.arange(N, dtype=np.float64) + 0.25)
    for rep in range(3):                                   # the same buffers call after call
        lam = torch.full((S * K,), float("nan"), dtype=torch.float64)            # only the rank's own rows are valid ...
        dz = torch.full((N,), float("nan"), dtype=torch.float64)
        lam[k0 * S:k1 * S] = torch.from_numpy(want_l[k0 * S:k1 * S]) * (rep + 1)
        if rank < world - 1:
            lam[k1 * S:(k1 + 1) * S] = 7.0                 # ... and the neighbour's ghost block behind them, which must NOT be gathered
        hi = min(k1 * n, N)
        dz[k0 * n:hi] = torch.from_numpy(want_z[k0 * n:hi]) * (rep + 1)
        gl, gz = plan.gather(lam, dz)
        assert gl.data_ptr() == plan.lam_out.data_ptr() and gz.data_ptr() == plan.dz_out.data_ptr()
        assert np.array_equal(gl.numpy(), want_l * (rep + 1)) and np.array_equal(gz.numpy(), want_z * (rep + 1)), rep
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{world} ok gather [{k0},{k1})")


if __name__ == "__main__":
    main()
