"""bench.py legs for more than one GPU.  One process per GPU, launched by torch.distributed.run.

Default (--gpus N, no --workload): the N = 1 workload (BASELINE.json configs[1], IIWA 14/7/50 fp64 whole step) on
every rank, each rank its own system, no data-path collective: weak scaling, value = all ranks' PCG iterations / s,
directly comparable with the N = 1 line.  The same JSON line carries, as "sharded", the knot-sharded solve of
configs[3] (IIWA 14/7, K = 4096 split over the ranks, RCCL all-gathers for the CG dots and halos; strong scaling: a
step = replicated assembly + sharded PCG of exactly 100 iterations + dz) with its parity against the one-GPU result.
--workload sharded_* makes that solve the line itself; --workload batched_* runs 512 systems per rank per call.
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


def replicas_leg(args, torch, dist, rank, local, world):
    """Default for --gpus N > 1: bench.py's N = 1 workload (BASELINE configs[1], IIWA 14/7/50 fp64, whole step) on every
    rank, each rank its own system - independent solves, no data-path collective, weak scaling.  The line is directly
    comparable with the N = 1 line; the knot-sharded solve of configs[3] rides along as the "sharded" object."""
    from . import synth
    from .solver import Solver
    name = "iiwa_14_7_k50_f64"
    S, C, K, dt, cfg = 14, 7, 50, np.float64, "configs[1]"          # = bench.py WORKLOADS[name]
    sysm = synth.make_system(S, C, K, seed=rank)
    sol = Solver(S, C, K, dt, local)
    dev = sol.upload_system(sysm)
    lam, dz = sol.new(S * K), sol.new(sol.N)
    step = lambda: sol.linsys(*dev, 0.0, MAX_ITERS, sysm.rho, lam, dz)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    sol.check_status()
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
    sol.check_status()
    # dominant kernel on rank 0: the PCG launch between HIP events on its own stream
    sol.set_option("time_pcg", 1)
    bufs = [sol.buffer_ptr(i) for i in (3, 4, 5)]
    ms = []
    for i in range(13):
        sol.pcg(bufs[0], bufs[1], bufs[2], 0.0, MAX_ITERS, lam=lam, check=False)
        if i >= 3:
            ms.append(sol.pcg_last_ms())
    pcg_ms = float(np.mean(ms))
    groups, threads = sol.get_option("last_groups"), sol.get_option("last_threads")
    sol.close()
    w = np.dtype(dt).itemsize
    bytes_launch = ((6 * K - 4) * S * S + 13 * S * K) * w * MAX_ITERS
    gbs = bytes_launch / (pcg_ms * 1e-3) / 1e9
    return {"metric": "PCG iterations/s", "value": MAX_ITERS * args.steps * world / el, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": name, "baseline_config": cfg, "STATE_SIZE": S, "CONTROL_SIZE": C, "KNOT_POINTS": K,
                       "max_iters": MAX_ITERS, "exit_tol": 0.0,
                       "step": "convert + Schur/stair assembly + PCG(100 iterations) + dz, inputs resident in HBM",
                       "parallelism": f"{world} independent systems, one per GPU, no data-path collective",
                       "pcg_kernel": "resident", "pcg_workgroups": groups, "pcg_threads": threads},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
                         "traffic": None, "kernel": "pcg_resident (rank 0, per GPU)", "launch_ms": pcg_ms,
                         "algorithmic_bytes_per_launch": bytes_launch}}


def sharded_leg(args, torch, dist, rank, local, world, name, steps, warmup):
    from . import synth
    from .dist import HipShardBackend, ShardedPCG
    from .solver import Solver
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

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
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
    out = None
    if rank == 0:
        w = np.dtype(dt).itemsize
        b_iter = ((6 * K - 4) * S * S + 13 * S * K) * w
        val = MAX_ITERS * steps / el
        out = {"metric": "PCG iterations/s", "value": val, "unit": "iterations/s", "n_gpus": world,
               "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * el / steps,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f64" if w == 8 else "f32", "data": "synthetic",
               "config": {"workload": name, "baseline_config": "configs[3]", "STATE_SIZE": S, "CONTROL_SIZE": C,
                          "KNOT_POINTS": K, "knots_per_gpu": K // world, "max_iters": MAX_ITERS, "exit_tol": 0.0,
                          "parallelism": f"knot-sharded x{world}, 2 RCCL all-gathers of (2S+1) scalars per iteration",
                          "note": "one K = 4096 system split over the ranks (strong scaling); an iteration is bound by "
                                  "the two collectives' latency, see same_system_on_one_gpu_resident_iters_per_s"},
               "roofline": {"bound": "hbm", "achieved": b_iter * val / 1e9, "peak": 8000.0 * world, "unit": "GB/s",
                            "frac": b_iter * val / 1e9 / (8000.0 * world), "traffic": None,
                            "kernel": "stream_step (whole sharded iteration incl. collectives)"},
               "parity": parity}
    sol.close()
    return out


def main(args):
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    wl = args.workload or ""
    if wl.startswith("batched"):
        return main_batched(args, torch, dist, rank, local, world)
    if wl in WORKLOADS:                                           # the knot-sharded solve as the line itself
        out = sharded_leg(args, torch, dist, rank, local, world, wl, args.steps, args.warmup)
    else:
        out = replicas_leg(args, torch, dist, rank, local, world)
        try:            # the line above must survive whatever happens in the rider (an error raised on every rank alike)
            sh = sharded_leg(args, torch, dist, rank, local, world, "sharded_k4096_f32", min(args.steps, 20), min(args.warmup, 3))
            if rank == 0:
                out["sharded"] = {k: sh[k] for k in ("value", "unit", "ms_per_step", "scaling", "dtype", "config", "roofline", "parity")}
        except Exception as e:   # noqa: BLE001
            if rank == 0:
                out["sharded"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0:
        print(json.dumps(out))
    dist.destroy_process_group()
