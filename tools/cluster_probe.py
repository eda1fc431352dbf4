#!/usr/bin/env python3
"""Diagnostic (GPU box): R ranks of a cluster solve as R solvers on R streams of one process; prints what every rank saw."""
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")     # one hardware queue per in-process rank
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                         # noqa: E402
from gato_python_amd import synth                                    # noqa: E402
from gato_python_amd.dist import ClusterPCG                          # noqa: E402
from gato_python_amd.solver import Solver                            # noqa: E402


def main():
    S, C, K, R = (int(x) for x in sys.argv[1:5])
    dt = np.float64 if (len(sys.argv) < 6 or sys.argv[5] == "f64") else np.float32
    mi = int(sys.argv[6]) if len(sys.argv) > 6 else 100
    sysm = synth.make_system(S, C, K, seed=13)
    one = Solver(S, C, K, dt)
    d = one.upload_system(sysm)
    Gd, Cd = one.convert(*d[:6], sysm.rho)
    Sb, Pb, gam, _ = one.form_schur(Gd, Cd, d[6], d[7])
    one.form_ss(Sb, Pb)
    lam1, it1 = one.pcg(Sb, Pb, gam, 0.0, mi)
    sols = [Solver(S, C, K, dt) for _ in range(R)]
    for s_ in sols:
        s_.set_option("timeout_ms", 300)
        s_.set_option("cluster_flat", int(os.environ.get("PROBE_FLAT", "1")))
        s_.set_option("pcg_variant", int(os.environ.get("PROBE_VARIANT", "0")))     # 1: single reduction, one exchange per iteration
        if R > 4:
            s_.set_option("max_workgroups", 256 // R)
    cl = [ClusterPCG(s_, r, R, inprocess_peers=True) for r, s_ in enumerate(sols)]
    ClusterPCG.connect_inprocess(cl)
    skip = int(os.environ.get("PROBE_SKIP_STREAMS", "0"))
    _unused = [torch.cuda.Stream() for _ in range(skip)]
    tol = float(os.environ.get("PROBE_TOL", "0"))
    streams = [torch.cuda.Stream() for _ in range(R)]
    print("streams", [hex(s_.cuda_stream) for s_ in streams], "mem_kind", sols[0].get_option("cluster_mem_kind"))
    for rep in range(3):
        lam = torch.zeros(S * K, dtype=one.dtype, device="cuda")
        its = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(R)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(R):
            cl[r].pcg(Sb, Pb, gam, tol, mi, lam, its[r], stream=streams[r].cuda_stream)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        err = float((lam - lam1).abs().max() / lam1.abs().max())
        print(f"rep {rep}: {el * 1e3:.2f} ms ({el / mi * 1e6:.2f} us/iter) iters {[int(i.cpu()[0]) for i in its]} err vs one GPU {err:.2e} "
              f"geometry {[(s_.get_option('last_groups'), s_.get_option('last_threads')) for s_ in sols]} flat {sols[0].get_option('last_cluster_flat')} "
              f"variant {sols[0].get_option('last_variant')}")
    # device time of one launch per rank
    for s_ in sols:
        s_.set_option("time_pcg", 1)
    lam = torch.zeros(S * K, dtype=one.dtype, device="cuda")
    torch.cuda.synchronize()
    for r in range(R):
        cl[r].pcg(Sb, Pb, gam, tol, mi, lam, its[r], stream=streams[r].cuda_stream)
    torch.cuda.synchronize()
    print("device ms per rank", [round(s_.pcg_last_ms(), 4) for s_ in sols])
    one.set_option("time_pcg", 1)
    one.pcg(Sb, Pb, gam, 0.0, mi)
    print("one launch on the whole GPU: ms", round(one.pcg_last_ms(), 4), "geometry", one.get_option("last_groups"), one.get_option("last_threads"))


if __name__ == "__main__":
    main()
