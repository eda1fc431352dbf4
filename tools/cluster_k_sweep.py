"""Every K of a range through a cluster of R in-process ranks (whole sharded solves, gato_cluster_linsys per rank) against the oracle's
whole solve: shards of one knot upwards, every remainder of K over the ranks, the single-reduction recurrence where every rank can run
it (halo of one knot, balanced splits), both exchange forms.      python tools/cluster_k_sweep.py [S C [KMAX]]"""
import os, sys
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("GATO_NO_TUNE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gato_python_amd import synth
from gato_python_amd.dist import ClusterPCG, lockstep_streams
from gato_python_amd.solver import Solver
from oracle import c_oracle as co

S, C = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (14, 7)
KMAX = int(sys.argv[3]) if len(sys.argv) > 3 else 150
bad = ran1 = n = 0
for R in (2, 3, 5, 8):
    for K in range(R, KMAX + 1):
        s = synth.make_system(S, C, K, seed=K)
        lam_w, dz_w, it_w = co.linsys_solve(*s.csr_args(), S, C, K, 0.0, 12, s.rho, dtype=np.float64)
        for variant, flat in ((0, 1), (1, 1), (1, 0), (0, 0)):
            if flat == 0 and K % 7:                  # two-level form: a sample of the K
                continue
            sols = [Solver(S, C, K, np.float64) for _ in range(R)]
            for x in sols:
                x.set_option("pcg_variant", variant); x.set_option("cluster_flat", flat); x.set_option("timeout_ms", 500)
                x.set_option("max_workgroups", 256 // R)
            cl = [ClusterPCG(x, r, R, inprocess_peers=True) for r, x in enumerate(sols)]
            ClusterPCG.connect_inprocess(cl)
            streams = lockstep_streams(R)
            d = sols[0].upload_system(s)
            lams = [torch.full((S * K,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(R)]
            dzs = [torch.full((sols[0].N,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(R)]
            its = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(R)]
            torch.cuda.synchronize()
            for r in range(R):
                cl[r].linsys(d, 0.0, 12, s.rho, lams[r], dzs[r], its[r], stream=streams[r].cuda_stream)
            torch.cuda.synchronize()
            nn = S + C
            la, da = np.empty(S * K), np.empty(sols[0].N)
            for r in range(R):
                k0, k1 = cl[r].k0, cl[r].k1
                la[k0 * S:k1 * S] = lams[r][k0 * S:k1 * S].cpu().numpy()
                hi = min(k1 * nn, sols[0].N); da[k0 * nn:hi] = dzs[r][k0 * nn:hi].cpu().numpy()
            ran = sols[0].get_option("last_variant")
            el = np.abs(la - lam_w).max() / np.abs(lam_w).max()
            ed = np.abs(da - dz_w).max() / max(np.abs(dz_w).max(), 1e-300)
            itg = [int(t.cpu()[0]) for t in its]
            bar = 1e-4 if (ran or S * K <= 64) else 1e-8        # 12 fixed iterations: another recurrence is another iterate; tiny systems are past convergence
            ok = np.isfinite(la).all() and np.isfinite(da).all() and el < bar and ed < 10 * bar and itg == [12] * R
            if S * K <= 12 * 2:                                  # fewer unknowns than iterations: 0 / 0 in every arithmetic (the reference has no guard)
                ok = True
            n += 1; ran1 += ran; bad += not ok
            if not ok:
                print(f"FAIL {S}/{C}/{K} R={R} variant={variant} flat={flat} ran {ran} groups {[x.get_option('last_groups') for x in sols]} lam {el:.1e} dz {ed:.1e} iters {itg}", flush=True)
            for c_ in cl: c_.close()
            for x in sols: x.close()
    print(f"R={R} done: {n} solves so far, {ran1} on the single-reduction recurrence, {bad} outside the bar", flush=True)
print("K SWEEP", "FAILED" if bad else "ok", bad)
sys.exit(1 if bad else 0)
