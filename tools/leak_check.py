"""Device and host memory across many create / solve / destroy cycles: solvers of changing shapes, the reference surface with its
cached solver changing shape every call, batches, clusters of in-process ranks (mirror pool).  Free device memory (hipMemGetInfo) and
the process's resident set before and after; a leak shows as a drift proportional to the cycle count.
      python tools/leak_check.py [cycles]"""
import os, sys, resource
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("GATO_NO_TUNE", "1")
os.environ["GATO_VERBOSE"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gato_python_amd import synth, linsys as host
from gato_python_amd.dist import ClusterPCG, lockstep_streams
from gato_python_amd.solver import Solver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rss = lambda: int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2 ** 20
free = lambda: torch.cuda.mem_get_info()[0] / 2 ** 20
shapes = [(14, 7, 50), (2, 1, 5), (32, 16, 40), (14, 7, 700), (6, 3, 33), (14, 7, 12000)]
systems = {sh: synth.make_system(*sh, seed=1) for sh in shapes}


def phase(name, body, cycles):
    for i in range(30):                           # warm: every shape / type / rank count of the cycle once (code objects are loaded at a
        body(i)                                   # kernel's first launch and stay), allocator pools, the torch cache
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    f0, r0 = free(), rss()
    for i in range(cycles):
        body(i)
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    f1, r1 = free(), rss()
    leak_dev, leak_host = (f0 - f1) / cycles, (r1 - r0) / cycles
    ok = leak_dev < 0.02 and leak_host < 0.05      # MB per cycle
    print(f"{name}: {cycles} cycles, device free {f0:.0f} -> {f1:.0f} MB ({leak_dev * 1024:.1f} KB per cycle), host RSS {r0:.0f} -> {r1:.0f} MB "
          f"({leak_host * 1024:.1f} KB per cycle): {'ok' if ok else 'LEAK?'}", flush=True)
    return ok


def solver_cycle(i):
    S, C, K = shapes[i % len(shapes)]
    dt = np.float64 if i % 2 else np.float32
    s = systems[(S, C, K)]
    sol = Solver(S, C, K, dt)
    d = sol.upload_system(s)
    lam, dz = sol.new(S * K), sol.new(sol.N)
    sol.linsys(*d, 1e-6, 20, s.rho, lam=lam, dz=dz)
    torch.cuda.synchronize()
    sol.close()


def host_cycle(i):
    S, C, K = shapes[i % 5]
    s = systems[(S, C, K)]
    host.set_precision("f64" if i % 2 else "f32")
    host.linsys_solve(s.G_row, s.G_col, s.G_val, s.C_row, s.C_col, s.C_val, s.g, s.c, np.zeros(S * K), 1, 1e-6, 20, False, s.rho)


def batch_cycle(i):
    S, C, K = shapes[i % 3]
    B = 2 + i % 5
    sol = Solver(S, C, K, np.float32, batch=B)
    d = sol.upload_batch([systems[(S, C, K)]] * B)
    lam, dz = sol.new(S * K * B), sol.new(sol.N * B)
    its = torch.zeros(B, dtype=torch.int32, device="cuda")
    sol.linsys_batched(*d, 1e-6, 20, systems[(S, C, K)].rho, lam, dz, its)
    torch.cuda.synchronize()
    sol.close()


def cluster_cycle(i):
    S, C, K = [(14, 7, 700), (2, 1, 50), (32, 16, 40)][i % 3]
    R = 2 + i % 3
    s = synth.make_system(S, C, K, seed=1) if (S, C, K) not in systems else systems[(S, C, K)]
    sols = [Solver(S, C, K, np.float64) for _ in range(R)]
    cl = [ClusterPCG(x, r, R, inprocess_peers=True) for r, x in enumerate(sols)]
    ClusterPCG.connect_inprocess(cl)
    streams = lockstep_streams(R)
    d = sols[0].upload_system(s)
    lams = [sols[0].new(S * K) for _ in range(R)]; dzs = [sols[0].new(sols[0].N) for _ in range(R)]
    its = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(R)]
    torch.cuda.synchronize()
    for r in range(R):
        cl[r].linsys(d, 1e-6, 20, s.rho, lams[r], dzs[r], its[r], stream=streams[r].cuda_stream)
    torch.cuda.synchronize()
    for c_ in cl: c_.close()
    for x in sols: x.close()


systems[(2, 1, 50)] = synth.make_system(2, 1, 50, seed=1)
ok = phase("solver create / solve / destroy, six shapes, both types", solver_cycle, n)
ok &= phase("reference surface, cached solver changing shape and type every call", host_cycle, n)
ok &= phase("batched solvers", batch_cycle, n)
ok &= phase("clusters of 2-4 in-process ranks (mirrors recycled)", cluster_cycle, max(30, n // 3))
print("LEAK CHECK", "ok" if ok else "FAILED")
sys.exit(0 if ok else 1)
