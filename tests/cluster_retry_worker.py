"""Worker of tests/test_gpu_cluster.py::test_cluster_survives_a_late_rank: two processes on cuda:0, linsys_solve_auto.  A rank
that launches later than timeout_ms makes the hand-off time out on every rank; that solve must come back complete over the RCCL
schedule, the NEXT one must run on the cluster again, and only three time-outs in a row may drop the cluster for good."""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gato_python_amd import synth                                              # noqa: E402
from gato_python_amd.dist import MAX_CONSECUTIVE_TIMEOUTS, close_state, linsys_solve_auto   # noqa: E402
from oracle import c_oracle as co                                              # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    S, C, K, dt, tol, mi = 14, 7, 300, np.float64, 1e-9, 150
    systems = [synth.make_system(S, C, K, seed=40 + i) for i in range(10)]
    answers = [co.linsys_solve(*s.csr_args(), S, C, K, tol, mi, s.rho, dtype=dt) for s in systems]

    def solve(i, state, late):
        dist.barrier()
        if late and rank == world - 1:
            time.sleep(0.6)                                   # far beyond timeout_ms below: the other rank's launch gives up
        lam, dz, its, state = linsys_solve_auto(systems[i], tol, mi, dt, 0, None, state)
        torch.cuda.synchronize()
        lam_w, dz_w, it_w = answers[i]
        el = np.abs(lam.cpu().numpy() - lam_w).max() / np.abs(lam_w).max()
        ed = np.abs(dz.cpu().numpy() - dz_w).max() / np.abs(dz_w).max()
        assert el < 1e-8 and ed < 1e-8 and int(its.cpu().reshape(-1)[0]) == it_w, (i, el, ed)
        return state

    state = solve(0, None, False)
    assert state["transport"] == "xgmi" and state["last_transport"] == "xgmi"
    state["sol"].set_option("timeout_ms", 150)
    state = solve(1, state, True)                             # one late rank: this solve over RCCL, the cluster stays
    assert state["transport"] == "xgmi" and state["last_transport"] == "rccl" and state["timeouts"] == 1, state.get("why")
    state = solve(2, state, False)                            # ... and serves the next solve again
    assert state["last_transport"] == "xgmi" and state["timeouts"] == 0
    state = solve(3, state, True)
    state = solve(4, state, False)
    assert state["last_transport"] == "xgmi" and state["timeouts"] == 0
    for n in range(MAX_CONSECUTIVE_TIMEOUTS):                 # late again and again: dropped for good
        state = solve(5 + n, state, True)
        assert state["last_transport"] == "rccl"
    assert state["transport"] == "rccl" and "cl" not in state
    state = solve(9, state, False)
    assert state["last_transport"] == "rccl"
    close_state(state)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{world} ok")


if __name__ == "__main__":
    main()
