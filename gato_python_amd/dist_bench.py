"""bench.py leg for the knot-sharded solve (BASELINE.json configs[3]: IIWA 14/7, K = 4096, knot points
sharded over the ranks, RCCL all-gathers for the CG dots and halos).  One process per GPU, launched by
torch.distributed.run; strong scaling: the K = 4096 system is fixed, each rank owns K/N block rows.

A step = one whole solve: replicated assembly + sharded PCG (exactly max_iters = 100 iterations,
exit_tol = 0) + lambda all-reduce + dz.  value = PCG iterations / s, max over ranks.
"""
from __future__ import annotations

import json
import os
import time

import numpy as np

MAX_ITERS = 100
WORKLOADS = {"sharded_k4096_f32": (14, 7, 4096, np.float32), "sharded_k4096_f64": (14, 7, 4096, np.float64),
             "sharded_s32_k1024_f32": (32, 16, 1024, np.float32),
             "sharded_k262144_f32": (14, 7, 262144, np.float32)}


def main_batched(args, torch, dist, rank, local, world):
    """--workload batched_*: every rank solves its own batch of independent 14/7/50 systems, no data-path
    collective - the weak-scaling mode in which more GPUs do pay for this shape."""
    from . import synth
    from .solver import Solver
    S, C, K, B = 14, 7, 50, 512
    dt = np.float32 if args.workload.endswith("f32") else np.float64
    base = synth.make_system(S, C, K, seed=rank)
    sol = Solver(S, C, K, dt, local, batch=B)
    dev = sol.upload_batch([base] * B)
    lam, dz = sol.new(B * S * K), sol.new(B * sol.N)
    iters = sol.new(B, torch.int32)
    step = lambda: sol.linsys_batched(*dev, 0.0, MAX_ITERS, base.rho, lam, dz, iters)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    el = float(el.item())
    if rank == 0:
        val = MAX_ITERS * B * world * args.steps / el
        print(json.dumps({"metric": "PCG iterations/s", "value": val, "unit": "iterations/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f64" if dt == np.float64 else "f32", "data": "synthetic",
                          "config": {"workload": args.workload, "STATE_SIZE": S, "CONTROL_SIZE": C, "KNOT_POINTS": K,
                                     "systems_per_gpu": B, "max_iters": MAX_ITERS, "exit_tol": 0.0,
                                     "parallelism": f"independent batches x{world}, no collective"}}))
    dist.destroy_process_group()


def main(args):
    import torch
    import torch.distributed as dist
    from . import synth
    from .dist import HipShardBackend, ShardedPCG
    from .solver import Solver

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    if (args.workload or "").startswith("batched"):
        return main_batched(args, torch, dist, rank, local, world)
    name = args.workload if args.workload in WORKLOADS else "sharded_k4096_f32"
    S, C, K, dt = WORKLOADS[name]
    sysm = synth.make_system(S, C, K, seed=0)
    sol = Solver(S, C, K, dt, local)
    d = sol.upload_system(sysm)

    def step(tol=0.0):
        Gd, Cd = sol.convert(*d[:6], sysm.rho)
        Sb, Pb, gam, Gi = sol.form_schur(Gd, Cd, d[6], d[7])
        sol.form_ss(Sb, Pb)
        be = HipShardBackend(sol, rank, world, Sb, Pb, gam, tol, MAX_ITERS)
        lam, iters = ShardedPCG(be).solve(MAX_ITERS)
        dz = sol.compute_dz(Gi, Cd, d[6], lam)
        return lam, dz, iters

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lam, dz, iters = step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    el = float(el.item())

    # parity of the sharded result against the single-GPU resident kernel on the same system (rank 0)
    parity = None
    if rank == 0:
        lam1, dz1 = sol.new(S * K), sol.new(sol.N)
        sol.linsys(*d, 0.0, MAX_ITERS, sysm.rho, lam1, dz1)
        torch.cuda.synchronize()
        den = float(lam1.abs().max())
        # the same system on ONE GPU with the register-resident kernel, timed here so that the strong-scaling ratio of
        # this line can be read off without comparing against bench.py's N=1 workload (a different shape)
        for _ in range(3):
            sol.linsys(*d, 0.0, MAX_ITERS, sysm.rho, lam1, dz1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n1 = 20
        for _ in range(n1):
            sol.linsys(*d, 0.0, MAX_ITERS, sysm.rho, lam1, dz1)
        torch.cuda.synchronize()
        single = MAX_ITERS * n1 / (time.perf_counter() - t1)
        parity = {"lam_rel_err_vs_single_gpu": float((lam - lam1).abs().max()) / den,
                  "dz_abs_err_vs_single_gpu": float((dz - dz1).abs().max()), "iters": int(iters.cpu()[0]),
                  "same_system_on_one_gpu_resident_iters_per_s": single}
    if rank == 0:
        w = np.dtype(dt).itemsize
        b_iter = ((6 * K - 4) * S * S + 13 * S * K) * w
        val = MAX_ITERS * args.steps / el
        out = {"metric": "PCG iterations/s", "value": val, "unit": "iterations/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f64" if w == 8 else "f32", "data": "synthetic",
               "config": {"workload": name, "baseline_config": "configs[3]", "STATE_SIZE": S, "CONTROL_SIZE": C,
                          "KNOT_POINTS": K, "knots_per_gpu": K // world, "max_iters": MAX_ITERS, "exit_tol": 0.0,
                          "parallelism": f"knot-sharded x{world}, 2 RCCL all-gathers of (2S+1) scalars per iteration",
                          "note": "N=1 (bench.py default) runs configs[1] on the register-resident kernel; this "
                                  "line is the sharded streaming solver, not comparable with it"},
               "roofline": {"bound": "hbm", "achieved": b_iter * val / 1e9, "peak": 8000.0 * world, "unit": "GB/s",
                            "frac": b_iter * val / 1e9 / (8000.0 * world), "traffic": None,
                            "kernel": "stream_step (whole sharded iteration incl. collectives)"},
               "parity": parity}
        print(json.dumps(out))
    dist.destroy_process_group()
