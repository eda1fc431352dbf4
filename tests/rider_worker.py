"""Worker of tests/test_dist_gloo.py::test_rider_child_jobs_cannot_take_the_bench_line_down: two ranks on gloo call
dist_bench.rider_in_child.  Here (no GPU) the child job cannot even select a device, and with GATO_RIDER_TEST_HANG=1 it
never returns: either way every rank must come back within the deadline, rank 0 with an error object instead of numbers,
and nobody may be left waiting in a collective."""
import json
import os
import sys
import time

import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gato_python_amd import dist_bench                             # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    t0 = time.time()
    res = dist_bench.rider_in_child(dist, rank, world, "sharded_k4096_f32", 1, 0, deadline=float(os.environ["RIDER_DEADLINE"]))
    dist.barrier()
    if rank == 0:
        print("RIDER " + json.dumps({"res": res, "seconds": time.time() - t0}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
