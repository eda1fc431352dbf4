"""Worker of tests/test_dist_gloo.py: one rank of a gloo world running ShardedPCG with the numpy backend."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gato_python_amd import synth                      # noqa: E402
from gato_python_amd.dist import ShardedPCG, knot_ranges   # noqa: E402
from oracle import gato_oracle as o                    # noqa: E402
from shard_numpy_backend import NumpyShardBackend      # noqa: E402


def main():
    S, C, K, tol, max_iters, check_every = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]),
                                            int(sys.argv[5]), int(sys.argv[6]))
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    s = synth.make_system(S, C, K, seed=7)
    out = o.linsys_solve(*s.csr_args(), S, C, K, tol, max_iters, s.rho, dtype=np.float64, return_all=True)
    be = NumpyShardBackend(S, K, rank, world, out["S"], out["Pinv"], out["gamma"], tol, max_iters)
    lam, iters = ShardedPCG(be).solve(max_iters, check_every=check_every)
    lam = lam.numpy()
    err = np.abs(lam - out["lam"]).max() / np.abs(out["lam"]).max()
    assert int(iters[0]) == out["iters"], (int(iters[0]), out["iters"])
    assert err < 1e-10, err
    # every rank holds the same assembled lambda
    ref = torch.from_numpy(lam.copy())
    dist.broadcast(ref, 0)
    assert np.array_equal(ref.numpy(), lam)
    assert knot_ranges(K, world)[rank] == (be.k0, be.k1)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{world} ok iters={int(iters[0])} err={err:.2e}")


if __name__ == "__main__":
    main()
