"""Soak of the cluster path on one GPU: R ranks as streams of one process, thousands of back-to-back WHOLE sharded solves
(gato_cluster_linsys: sharded assembly, the rank's persistent launch with its in-kernel exchanges and the lambda ghost block, dz),
every rank's lambda rows, ghost block and dz rows compared bit for bit with the first solve, status words checked: an exchange
must never lose or reorder a granule, the ghost block must never come from another launch.  Both recurrences, both exchange forms.
      python tools/cluster_soak.py [solves per case]"""
import os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gato_python_amd import synth
from gato_python_amd.dist import ClusterPCG, lockstep_streams
from gato_python_amd.solver import Solver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
bad = 0
for (S, C, K, R, dt) in [(14, 7, 600, 3, np.float32), (14, 7, 4096, 2, np.float32), (14, 7, 300, 4, np.float64), (32, 16, 256, 2, np.float32)]:
    s = synth.make_system(S, C, K, seed=5)
    for variant in (0, 1):
        for flat in (1, 0):
            sols = [Solver(S, C, K, dt) for _ in range(R)]
            for x in sols:
                x.set_option("pcg_variant", variant); x.set_option("cluster_flat", flat)
            cl = [ClusterPCG(x, r, R, inprocess_peers=True) for r, x in enumerate(sols)]
            ClusterPCG.connect_inprocess(cl)
            d = sols[0].upload_system(s)
            streams = lockstep_streams(R)
            lams = [torch.zeros(S * K, dtype=sols[0].dtype, device="cuda") for _ in range(R)]
            dzs = [torch.zeros(sols[0].N, dtype=sols[0].dtype, device="cuda") for _ in range(R)]
            its = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(R)]
            torch.cuda.synchronize()

            def solve():
                for r in range(R):
                    cl[r].linsys(d, 0.0, 30, s.rho, lams[r], dzs[r], its[r], stream=streams[r].cuda_stream)
            solve(); torch.cuda.synchronize()
            nn = S + C
            def snap():
                out = []
                for r in range(R):
                    k0, k1 = cl[r].k0, cl[r].k1
                    hi = min(k1 + 1, K)
                    out.append((lams[r][k0 * S:hi * S].clone(), dzs[r][k0 * nn:min(k1 * nn, sols[0].N)].clone(), int(its[r].cpu()[0])))
                return out
            ref = snap()
            t0 = time.time()
            mism = 0
            for i in range(n):
                solve()
                if i % 100 == 99:
                    torch.cuda.synchronize()
                    cur = snap()
                    for a, b in zip(cur, ref):
                        if not (torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2]):
                            mism += 1
                    for x in sols:
                        x.check_status()
            torch.cuda.synchronize()
            bad += mism
            print(f"{S}/{C}/{K} {np.dtype(dt).name} R={R} variant {sols[0].get_option('last_variant')} flat {sols[0].get_option('last_cluster_flat')}: "
                  f"{n} solves in {time.time() - t0:.1f} s, iters {[x[2] for x in ref]}, mismatching checks {mism}", flush=True)
            for c in cl: c.close()
            for x in sols: x.close()
print("SOAK", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
