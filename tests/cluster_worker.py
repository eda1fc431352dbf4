"""Worker of tests/test_gpu_cluster.py: one rank of a cluster solve, one PROCESS per rank, all ranks on cuda:0 (a 1-GPU
box cannot give every rank its own GPU; the code path - IPC-mapped mirrors, system-scope peer stores, a launch per
process waiting for the other processes' launches on the device - is the one an 8-GPU node runs).  The gloo group
only carries the IPC handles and the barriers."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gato_python_amd import synth                      # noqa: E402
from gato_python_amd.dist import ClusterPCG            # noqa: E402
from gato_python_amd.solver import Solver              # noqa: E402
from oracle import c_oracle as co                      # noqa: E402


def main():
    S, C, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    dt = np.float64 if sys.argv[4] == "f64" else np.float32
    tol, mi, repeats = float(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    variant = int(sys.argv[8]) if len(sys.argv) > 8 else 0       # 1: the single-reduction recurrence (one exchange per iteration)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    s = synth.make_system(S, C, K, seed=13)
    Gd, Cd = co.convert(*s.csr_args()[:6], S, C, K, s.rho, dt)
    Sb, Pb, gam, _ = co.form_schur(Gd, Cd, s.g, s.c, S, C, K)
    Pb = co.form_ss(Sb, Pb, S, K)
    lam_o, it_o = co.pcg(Sb, Pb, gam, S, K, tol, mi)
    sol = Solver(S, C, K, dt)
    sol.set_option("pcg_variant", variant)
    cl = ClusterPCG(sol, rank, world)
    dS, dP, dg = sol.to_device(Sb), sol.to_device(Pb), sol.to_device(gam)
    f64 = dt == np.float64
    for rep in range(repeats):
        lam = torch.zeros(S * K, dtype=sol.dtype, device="cuda:0")
        iters = torch.zeros(1, dtype=torch.int32, device="cuda:0")
        dist.barrier()
        cl.pcg(dS, dP, dg, tol, mi, lam, iters)
        torch.cuda.synchronize()
        sol.check_status()
        it = int(iters.cpu()[0])
        assert sol.get_option("last_variant") == variant
        assert abs(it - it_o) <= ((0 if f64 else 2) + (1 if variant else 0)), (it, it_o)
        # the right neighbour's first lambda block arrived inside the launch (row k1: what dz of this rank's last knot needs)
        if rank < world - 1:
            ghost = lam.cpu().numpy()[cl.k1 * S:(cl.k1 + 1) * S]
            assert np.abs(ghost - lam_o[cl.k1 * S:(cl.k1 + 1) * S]).max() / np.abs(lam_o).max() < (1e-8 if f64 else 2e-3)
        mine = lam.cpu().numpy()[cl.k0 * S:cl.k1 * S]
        err = np.abs(mine - lam_o[cl.k0 * S:cl.k1 * S]).max() / np.abs(lam_o).max()
        assert err < ((1e-8 if variant else 1e-9) if f64 else 2e-3), err         # one rank's slice (the assembled solution is judged below)
        # the slices of all ranks assemble the full solution
        full = torch.zeros(S * K, dtype=lam.dtype)
        full[cl.k0 * S:cl.k1 * S] = lam.cpu()[cl.k0 * S:cl.k1 * S]
        dist.all_reduce(full)
        errf = np.abs(full.numpy() - lam_o).max() / np.abs(lam_o).max()
        if f64:
            assert errf < (1e-8 if variant else 1e-9), errf
        else:       # fp32: measured against the converged fp64 solution of the same matrices, beside the fp32 oracle's error
            truth = co.pcg(Sb.astype(np.float64), Pb.astype(np.float64), gam.astype(np.float64), S, K, 1e-14, 600)[0]
            den = np.abs(truth).max()
            eg, eo = np.abs(full.numpy() - truth).max() / den, np.abs(lam_o.astype(np.float64) - truth).max() / den
            assert eg <= 2.0 * eo + 5e-6, (eg, eo)
    # the cluster's 32-bit epoch space, two launches before its end on every rank (test hook): the third solve renews it - every
    # rank waits for its launches, barrier, mirrors and slots zeroed, barrier - and the solves go on
    need = 2 * mi + 8
    top = 0xFFFFFFFF - need - 8
    dist.barrier()
    sol.set_option("cluster_epoch", (top - need - 5) - (1 << 32))
    sol.set_option("pcg_epoch", (top - need - 5) - (1 << 32))
    assert cl.launches_left(mi) == 2
    for rep in range(4):
        lam = torch.zeros(S * K, dtype=sol.dtype, device="cuda:0")
        cl.pcg(dS, dP, dg, tol, mi, lam, iters)
        torch.cuda.synchronize()
        sol.check_status()
        assert abs(int(iters.cpu()[0]) - it_o) <= ((0 if f64 else 2) + (1 if variant else 0))
        mine = lam.cpu().numpy()[cl.k0 * S:cl.k1 * S]
        assert np.abs(mine - lam_o[cl.k0 * S:cl.k1 * S]).max() / np.abs(lam_o).max() < ((1e-8 if variant else 1e-9) if f64 else 2e-3)
        assert cl.rewinds == (0 if rep < 2 else 1), (rep, cl.rewinds)
    assert cl.launches_left(mi) > 1_000_000
    mem = sol.get_option("cluster_mem_kind")
    groups, threads = sol.get_option("last_groups"), sol.get_option("last_threads")
    dist.barrier()
    cl.close()
    sol.close()
    # the WHOLE solve through the product entry (sharded assembly + launch + dz in one library call per rank, one all-gather of
    # the lambda / dz rows), twice through one state, against the oracle's whole solve
    from gato_python_amd.dist import close_state, linsys_solve_cluster
    lam_w, dz_w, it_w = co.linsys_solve(*s.csr_args(), S, C, K, tol, mi, s.rho, dtype=dt)
    state = None
    for rep in range(2):
        lam_c, dz_c, its_c, state = linsys_solve_cluster(s, tol, mi, dt, 0, None, state, variant=variant)
        assert abs(int(its_c.cpu()[0]) - it_w) <= ((0 if f64 else 2) + (1 if variant else 0))
        el, ez = np.abs(lam_c.cpu().numpy() - lam_w).max() / np.abs(lam_w).max(), np.abs(dz_c.cpu().numpy() - dz_w).max() / np.abs(dz_w).max()
        if f64:
            assert el < 1e-8 and ez < 1e-8, (el, ez)
        else:
            s64 = s.astype(np.float32).astype(np.float64)
            lam_t, dz_t, _ = co.linsys_solve(*s64.csr_args(), S, C, K, 1e-14, 600, float(np.float32(s.rho)), dtype=np.float64)
            for got, orc, tr in ((lam_c, lam_w, lam_t), (dz_c, dz_w, dz_t)):
                den = np.abs(tr).max()
                eg, eo = np.abs(got.cpu().numpy() - tr).max() / den, np.abs(orc.astype(np.float64) - tr).max() / den
                assert eg <= 2.0 * eo + 5e-6, (eg, eo)
    assert state["sol"].get_option("last_variant") == variant
    close_state(state)
    dist.destroy_process_group()
    print(f"rank {rank}/{world} ok iters={it} err={errf:.2e} mem_kind={mem} geometry={groups}x{threads}")


if __name__ == "__main__":
    main()
