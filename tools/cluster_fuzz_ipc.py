"""Randomised whole solves over a cluster of PROCESSES (one rank per process, all on cuda:0 of a 1-GPU box; the code path - IPC-mapped
mirrors, one library call per rank and solve, one all-gather of the rows - is the one a multi-GPU node runs): shape, K, recurrence,
exchange form, workgroup size; every case connects a fresh cluster state (linsys_solve_cluster), solves twice through it (the second
time another system of the same shape, sometimes the other recurrence), closes it; now and then the automatic entry
(linsys_solve_auto).  At most 5 ranks on a 1-GPU box (the launcher counts towards its limit of 6 processes on the card).  Every rank checks the gathered lambda / dz against the oracle's whole solve.  All ranks draw the same cases.
      python -m torch.distributed.run --nnodes=1 --nproc-per-node R --master-addr 127.0.0.1 --master-port P tools/cluster_fuzz_ipc.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from gato_python_amd import synth
from gato_python_amd.dist import ClusterUnavailable, close_state, linsys_solve_auto, linsys_solve_cluster
from oracle import c_oracle as co


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dist.init_process_group("gloo")
    rank, R = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    rng = np.random.default_rng(seed)
    bad = 0
    for i in range(n):
        S, C = [(14, 7), (14, 7), (2, 1), (12, 6), (32, 16), (4, 2), (6, 3)][int(rng.integers(0, 7))]
        kmax = {2: 6000, 4: 3000, 6: 2500, 12: 1500, 14: 2500, 32: 600}[S]
        K = int(rng.integers(2 * R, kmax))
        variant, flat = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        threads = int(rng.choice([0, 0, 64, 128, 256, 512]))
        auto = int(rng.integers(0, 5) == 0)
        v2 = int(rng.integers(0, 2))
        tag = f"case {i}: {S}/{C}/{K} R={R} variant={variant}->{v2} flat={flat} threads={threads} auto={auto}"
        dt = np.float64
        tol, mi = 1e-9, 200
        sys_a = synth.make_system(S, C, K, seed=3000 + 2 * i)
        sys_b = synth.make_system(S, C, K, seed=3001 + 2 * i)
        ok, notes, state = True, [], None
        try:
            for rep, (sm, v) in enumerate(((sys_a, variant), (sys_b, v2))):
                if auto:
                    lam, dz, its, state = linsys_solve_auto(sm, tol, mi, dt, 0, None, state, variant=v)
                else:
                    opts = {"cluster_flat": flat, "pcg_threads": threads, "max_workgroups": max(1, 256 // R)}
                    lam, dz, its, state = linsys_solve_cluster(sm, tol, mi, dt, 0, None, state, variant=v, solver_options=opts)
                torch.cuda.synchronize()
                ran = state["sol"].get_option("last_variant") if "sol" in state else 0
                lam_w, dz_w, it_w = co.linsys_solve(*sm.csr_args(), S, C, K, tol, mi, sm.rho, dtype=dt)
                el, ed, it = rel(lam.cpu().numpy(), lam_w), rel(dz.cpu().numpy(), dz_w), int(its.cpu().reshape(-1)[0])
                bar = 1e-5 if ran else 1e-8
                good = el < bar and ed < bar and (ran or it == it_w) and abs(it - it_w) <= 2
                ok = ok and good
                notes.append(f"solve {rep}: transport {state.get('transport', 'xgmi')} ran_variant {ran} lam {el:.1e} dz {ed:.1e} iters {it} (oracle {it_w})")
        except ClusterUnavailable as e:              # every rank alike (a geometry the options do not allow)
            notes.append("unavailable: " + str(e)[:80])
        except Exception as e:                       # noqa: BLE001
            ok = False
            notes.append("RAISED " + repr(e)[:200])
        if state is not None:
            close_state(state)
        flags = [None] * R
        dist.all_gather_object(flags, ok)
        if rank == 0:
            print(("ok   " if all(flags) else f"FAIL (ranks {[r for r, f in enumerate(flags) if not f]}) ") + tag + " | " + "; ".join(notes), flush=True)
        bad += 0 if all(flags) else 1
    if rank == 0:
        print("FUZZ", "FAILED" if bad else "ok", bad, flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
