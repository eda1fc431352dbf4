"""bench.py legs for more than one GPU.  One process per GPU, launched by torch.distributed.run.

Default (--gpus N, no --workload): the last line is LED BY THE SHARDED SYSTEM - `value` = PCG iterations/s of ONE IIWA
14/7/4096 system (BASELINE configs[3]) knot-sharded over the N GPUs, whole steps (sharded assembly + sharded PCG of exactly
100 iterations + sharded dz + the gather of lambda / dz), "scaling": "strong", with the same system on one GPU beside it
(`one_gpu_value`, `sharded_speedup`).  The N = 1 workload (BASELINE configs[1], IIWA 14/7/50 fp64 whole step) runs on every
rank first, each rank its own system, no data-path collective (weak scaling: that shape is one workgroup on one CU, it does
not shard): it is printed as an EARLIER line and rides in the last one under `replicas`.  configs[4] (32/16/1024) and
K = 262144 (the size where splitting can pay) ride under `config.sharded`, each with its us per iteration next to the same
system on ONE GPU.
Transport of the sharded PCG: "xgmi" = ONE persistent launch per rank with the dot + halo exchange inside the kernel
(peer stores into IPC-mapped mirrors, gato_cluster_*); if the mirrors cannot be mapped or the first solve times out,
"rccl" = two launches + two RCCL all-gathers per iteration (gato_shard_pcg_*).  The line says which one ran.
The riders run as CHILD jobs of the ranks (rider_in_child: own rendezvous, a deadline, results through a file), so that a
fault or a hang in the cross-GPU exchange cannot take the line with the replicas value down with it.
--workload sharded_* makes that solve the line itself; --workload batched_* runs 512 systems per rank per call.

GATO_BENCH_ONE_GPU=1 (rehearsal on a 1-GPU box): every rank uses cuda:0, host collectives run over gloo, each rank
counts on its share of the CUs only.
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
ONE_GPU = os.environ.get("GATO_BENCH_ONE_GPU") == "1"


def _max_over_ranks(x, torch, dist):
    t = torch.tensor([x], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _timed(step, steps, warmup, torch, dist):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over the ranks."""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    return _max_over_ranks(time.perf_counter() - t0, torch, dist)


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
    el = _timed(lambda: sol.linsys_batched(*dev, 0.0, MAX_ITERS, base.rho, lam, dz, iters), args.steps, args.warmup, torch, dist)
    if rank == 0:
        val = MAX_ITERS * B * world * args.steps / el
        print(dumps_strict({"metric": "PCG iterations/s", "value": val, "unit": "iterations/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f64" if dt == np.float64 else "f32", "data": "synthetic",
                          "config": {"workload": args.workload, "STATE_SIZE": S, "CONTROL_SIZE": C, "KNOT_POINTS": K,
                                     "systems_per_gpu": B, "max_iters": MAX_ITERS, "exit_tol": 0.0,
                                     "parallelism": f"independent batches x{world}, no collective"}}))
    dist.destroy_process_group()


def replicas_leg(args, torch, dist, rank, local, world):
    """Default for --gpus N > 1: bench.py's N = 1 workload (BASELINE configs[1], IIWA 14/7/50 fp64, whole step) on every
    rank, each rank its own system - independent solves, no data-path collective, weak scaling ("replicas only": one
    K = 50 system is one workgroup on one CU and cannot be split).  Directly comparable with the N = 1 line."""
    from . import synth
    from .solver import Solver
    name = "iiwa_14_7_k50_f64"
    S, C, K, dt, cfg = 14, 7, 50, np.float64, "configs[1]"          # = bench.py WORKLOADS[name]
    sysm = synth.make_system(S, C, K, seed=rank)
    sol = Solver(S, C, K, dt, local)
    dev = sol.upload_system(sysm)
    lam, dz = sol.new(S * K), sol.new(sol.N)
    el = _timed(lambda: sol.linsys(*dev, 0.0, MAX_ITERS, sysm.rho, lam, dz), args.steps, args.warmup, torch, dist)
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
                       "parallelism": f"replicas only: {world} independent systems, one per GPU, no data-path collective; "
                                      "the knot-sharded solves (real exchange) are in config.sharded",
                       "pcg_kernel": "resident", "pcg_workgroups": groups, "pcg_threads": threads},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
                         "traffic": None, "limiter": "latency (matrices register/LDS-resident: HBM read once per launch)",
                         "kernel": "pcg_resident (rank 0, per GPU)", "launch_ms": pcg_ms,
                         "algorithmic_bytes_per_launch": bytes_launch}}


from .dist import first_working_kind          # noqa: E402,F401  (tests import it from here too)


def sharded_leg(args, torch, dist, rank, local, world, name, steps, warmup):
    """One knot-sharded system over the ranks, strong scaling.  A step = this rank's sharded assembly + its persistent launch
    (exactly MAX_ITERS iterations) + dz on its knots (ONE library call, gato_cluster_linsys) + one all-gather of the lambda / dz
    rows (dist._GatherPlan).  Timed for the default recurrence (two in-kernel exchanges per iteration: `value`) and for the
    single-reduction recurrence (one exchange: "single_reduction"), with the outputs gathered and left sharded."""
    from . import synth
    from .dist import ClusterUnavailable, HipShardBackend, ShardedPCG, close_state, linsys_solve_cluster
    from .solver import Solver
    S, C, K, dt = WORKLOADS[name]
    sysm = synth.make_system(S, C, K, seed=0)
    want = os.environ.get("GATO_SHARD_TRANSPORT", "xgmi")
    transport, why, state, sol = "rccl", "", None, None
    opts = {"max_workgroups": max(1, 240 // world)} if ONE_GPU else {}

    def step_xgmi(variant=0, gather=True):
        return linsys_solve_cluster(sysm, 0.0, MAX_ITERS, dt, local, None, state, check=False, gather=gather, variant=variant)[:3]

    if want == "xgmi":
        try:
            state = linsys_solve_cluster(sysm, 0.0, MAX_ITERS, dt, local, None, None, check=True, solver_options=opts)[3]
            transport, why, sol = "xgmi", state.get("rejected", ""), state["sol"]
        except ClusterUnavailable as e:
            state, why = None, str(e)[:300]
    if transport != "xgmi":
        sol = Solver(S, C, K, dt, local)
        for k_, v_ in opts.items():
            sol.set_option(k_, v_)
        d = sol.upload_system(sysm)

        def step_rccl():
            Gd, Cd = sol.convert(*d[:6], sysm.rho)
            Sb, Pb, gam, Gi = sol.form_schur(Gd, Cd, d[6], d[7])
            sol.form_ss(Sb, Pb)
            be = HipShardBackend(sol, rank, world, Sb, Pb, gam, 0.0, MAX_ITERS)
            lam, iters = ShardedPCG(be).solve(MAX_ITERS)
            return lam, sol.compute_dz(Gi, Cd, d[6], lam), iters
    step = step_xgmi if transport == "xgmi" else step_rccl
    el = _timed(lambda: step(), steps, warmup, torch, dist)
    lam, dz, iters = step()
    torch.cuda.synchronize()
    lam, dz = lam.clone(), dz.clone()

    def launch_us(fn):                                    # device time of this rank's launch (it waits for its peers inside)
        sol.set_option("time_pcg", 1)
        ms = []
        for _ in range(5):
            dist.barrier()
            fn()
            ms.append(sol.pcg_last_ms())
        sol.set_option("time_pcg", 0)
        return 1e3 * float(np.mean(ms[1:])) / MAX_ITERS
    pcg_us, extra = None, {}
    if transport == "xgmi":
        pcg_us = launch_us(step)
        groups, threads, semi = sol.get_option("last_groups"), sol.get_option("last_threads"), sol.get_option("last_semi")
        # the same step with lambda / dz left sharded (no collective at all), and the single-reduction recurrence both ways
        el_ng = _timed(lambda: step_xgmi(0, False), steps, warmup, torch, dist)
        extra = {"ms_per_step_outputs_sharded": 1e3 * el_ng / steps, "flat_exchange": bool(sol.get_option("last_cluster_flat"))}
        try:
            el1 = _timed(lambda: step_xgmi(1, True), steps, warmup, torch, dist)
            lam_v1 = step_xgmi(1, True)[0]
            torch.cuda.synchronize()
            ran = sol.get_option("last_variant")
            us1 = launch_us(lambda: step_xgmi(1, True))
            el1_ng = _timed(lambda: step_xgmi(1, False), steps, warmup, torch, dist)
            extra["single_reduction"] = {"ran_variant": ran, "iters_per_s": MAX_ITERS * steps / el1, "ms_per_step": 1e3 * el1 / steps,
                                         "ms_per_step_outputs_sharded": 1e3 * el1_ng / steps, "pcg_us_per_iter": us1,
                                         "workgroups_per_gpu": sol.get_option("last_groups"), "threads": sol.get_option("last_threads"),
                                         "flat_exchange": bool(sol.get_option("last_cluster_flat")),
                                         "lam_rel_diff_vs_default_recurrence": float((lam_v1 - lam).abs().max()) / float(lam.abs().max())}
        except Exception as e:    # noqa: BLE001  (raised on every rank alike: option and geometry are the same everywhere)
            extra["single_reduction"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        step_xgmi(0, True)
        torch.cuda.synchronize()
    else:
        groups, threads, semi = sol.get_option("last_groups"), sol.get_option("last_threads"), sol.get_option("last_semi")
    mem_kind = sol.get_option("cluster_mem_kind")

    # the same system on ONE GPU (rank 0), for parity and for the strong-scaling ratio of this very shape
    parity = None
    if rank == 0:
        one = Solver(S, C, K, dt, local)
        d1 = one.upload_system(sysm)
        lam1, dz1 = one.new(S * K), one.new(one.N)
        single = single_us = None
        try:
            one.linsys(*d1, 0.0, MAX_ITERS, sysm.rho, lam1, dz1)
            torch.cuda.synchronize()
            one.check_status()
            for _ in range(2):
                one.linsys(*d1, 0.0, MAX_ITERS, sysm.rho, lam1, dz1)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n1 = 10
            for _ in range(n1):
                one.linsys(*d1, 0.0, MAX_ITERS, sysm.rho, lam1, dz1)
            torch.cuda.synchronize()
            single = MAX_ITERS * n1 / (time.perf_counter() - t1)
            one.set_option("time_pcg", 1)
            bufs = [one.buffer_ptr(i) for i in (3, 4, 5)]
            mm = []
            for _ in range(4):
                one.pcg(bufs[0], bufs[1], bufs[2], 0.0, MAX_ITERS, lam=lam1, check=False)
                mm.append(one.pcg_last_ms())
            single_us = 1e3 * float(np.mean(mm[1:])) / MAX_ITERS
            one.linsys(*d1, 0.0, MAX_ITERS, sysm.rho, lam1, dz1)
            torch.cuda.synchronize()
            den = float(lam1.abs().max())
            if isinstance(extra.get("single_reduction"), dict) and "error" not in extra["single_reduction"]:
                # the same system on one GPU with the single-reduction recurrence: what the sharded single-reduction numbers stand beside
                v1 = Solver(S, C, K, dt, local)
                v1.set_option("pcg_variant", 1)
                la, da = v1.new(S * K), v1.new(v1.N)
                for _ in range(3):
                    v1.linsys(*d1, 0.0, MAX_ITERS, sysm.rho, la, da)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(n1):
                    v1.linsys(*d1, 0.0, MAX_ITERS, sysm.rho, la, da)
                torch.cuda.synchronize()
                extra["single_reduction"]["one_gpu_iters_per_s"] = MAX_ITERS * n1 / (time.perf_counter() - t1)
                v1.set_option("time_pcg", 1)
                b1 = [v1.buffer_ptr(i) for i in (3, 4, 5)]
                mm1 = []
                for _ in range(4):
                    v1.pcg(b1[0], b1[1], b1[2], 0.0, MAX_ITERS, lam=la, check=False)
                    mm1.append(v1.pcg_last_ms())
                extra["single_reduction"]["one_gpu_us_per_iter"] = 1e3 * float(np.mean(mm1[1:])) / MAX_ITERS
                extra["single_reduction"]["one_gpu_ran_variant"] = v1.get_option("last_variant")
                v1.close()
            parity = {"lam_rel_err_vs_single_gpu": float((lam - lam1).abs().max()) / den,
                      "dz_abs_err_vs_single_gpu": float((dz - dz1).abs().max()), "iters": int(iters.cpu()[0]),
                      "same_system_on_one_gpu_iters_per_s": single, "same_system_on_one_gpu_pcg_us_per_iter": single_us,
                      "same_system_on_one_gpu_kernel": {1: "resident", 2: "streaming"}.get(one.get_option("last_mode")) +
                                                       (" (semi)" if one.get_option("last_semi") else "")}
        except Exception as e:    # noqa: BLE001
            parity = {"error": f"{type(e).__name__}: {e}"[:200], "iters": int(iters.cpu()[0])}
        one.close()
    if ONE_GPU:
        dist.barrier()
    out = None
    if rank == 0:
        w = np.dtype(dt).itemsize
        b_iter = ((6 * K - 4) * S * S + 13 * S * K) * w
        val = MAX_ITERS * steps / el
        ms_step = 1e3 * el / steps
        out = {"metric": "PCG iterations/s", "value": val, "unit": "iterations/s", "n_gpus": world,
               "steps": steps, "warmup": warmup, "ms_per_step": ms_step,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f64" if w == 8 else "f32", "data": "synthetic",
               "config": {"workload": name, "baseline_config": "configs[3]" if K == 4096 else ("configs[4]" if S == 32 else "beyond BASELINE: the size where sharding pays"),
                          "STATE_SIZE": S, "CONTROL_SIZE": C, "KNOT_POINTS": K, "knots_per_gpu": K // world,
                          "max_iters": MAX_ITERS, "exit_tol": 0.0, "transport": transport, "transport_fallback_reason": why,
                          "mirror_memory": {0: "uncached", 1: "fine-grained", 2: "hipMalloc"}.get(mem_kind),
                          "pcg_workgroups_per_gpu": groups, "pcg_threads": threads, "semi_resident": bool(semi),
                          "step": "sharded assembly + persistent launch (100 iterations) + dz on the rank's knots: one library call per rank; "
                                  "then one all-gather of the lambda / dz rows" if transport == "xgmi" else "replicated assembly + sharded PCG + dz",
                          "parallelism": (f"knot-sharded x{world}; xgmi: one persistent launch per GPU, 2 in-kernel exchanges per iteration "
                                          "(rank totals to every peer, edge blocks to the neighbours, peer stores into IPC-mapped mirrors); "
                                          "single_reduction: 1 exchange per iteration"
                                          if transport == "xgmi" else
                                          f"knot-sharded x{world}; rccl: 2 launches + 2 RCCL all-gathers of (2S+1) scalars per iteration"),
                          "one_gpu_rehearsal": ONE_GPU},
               "pcg_us_per_iter": pcg_us,
               "out_of_loop_ms": (ms_step - pcg_us * MAX_ITERS * 1e-3) if pcg_us else None,
               "roofline": {"bound": "hbm", "achieved": b_iter * val / 1e9, "peak": 8000.0 * world, "unit": "GB/s",
                            "frac": b_iter * val / 1e9 / (8000.0 * world), "traffic": None,
                            "kernel": "pcg_resident (cluster launch)" if transport == "xgmi" else "stream_step + all-gathers",
                            "note": "whole sharded step incl. assembly, dz and the gather of the outputs"},
               "parity": parity}
        out.update(extra)
    if state is not None:
        close_state(state)
    elif sol is not None:
        sol.close()
    return out


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def rider_in_child(dist, rank, world, rider, steps, warmup, deadline):
    """One knot-sharded rider as a CHILD job: every rank starts `python -m gato_python_amd.dist_bench --rider ...` with its
    own RANK / LOCAL_RANK / WORLD_SIZE and a fresh rendezvous port, waits for it with a deadline and kills exactly the
    process group it started when the deadline passes.  The in-kernel cross-GPU exchange is the one part of this bench
    that can fault or hang for reasons outside the process (peer mappings, fabric): the line with the replicas value must
    be printed whatever happens there, and a Python try/except does not survive a GPU fault or a rank stuck in a
    collective.  Rank 0's child leaves its JSON in a temporary file."""
    import signal
    import subprocess
    import sys
    import tempfile
    port = [_free_port() if rank == 0 else 0]
    dist.broadcast_object_list(port, src=0)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tmp = tempfile.gettempdir()
    out_path = os.path.join(tmp, f"gato_rider_{port[0]}_{rider}.json")
    log_path = os.path.join(tmp, f"gato_rider_{port[0]}_{rider}_rank{rank}.log")
    # a job of its own: under torch.distributed.run the environment tells init_process_group to join the AGENT's store
    # (TORCHELASTIC_USE_AGENT_STORE) - on the new port nobody would be serving it; without those variables rank 0 of the
    # child job hosts its store itself
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env["MASTER_PORT"] = str(port[0])
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "gato_python_amd.dist_bench", "--rider", rider, "--steps", str(steps), "--warmup", str(warmup),
           "--out", out_path]
    rc = None
    with open(log_path, "w") as log:
        p = subprocess.Popen(cmd, env=env, cwd=root, stdout=log, stderr=subprocess.STDOUT, start_new_session=True)
        try:
            rc = p.wait(timeout=deadline)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)          # the session this rank started, nothing else
            except ProcessLookupError:
                pass
            p.wait()
            rc = "deadline"
    dist.barrier()
    res = None
    if rank == 0:
        try:
            res = json.loads(open(out_path).read())
        except Exception:     # noqa: BLE001
            tail = ""
            try:
                tail = open(log_path).read()[-400:]
            except OSError:
                pass
            res = {"error": f"rider child job did not deliver (exit {rc}, deadline {deadline} s)", "log_tail": tail}
        for f in (out_path,):
            try:
                os.remove(f)
            except OSError:
                pass
    try:
        os.remove(log_path)
    except OSError:
        pass
    return res


RIDER_KEYS = ("value", "unit", "ms_per_step", "scaling", "dtype", "config", "pcg_us_per_iter", "out_of_loop_ms", "roofline", "parity",
              "ms_per_step_outputs_sharded", "flat_exchange", "single_reduction")
# the knot-sharded solves that ride along with the replicas line: configs[3] (K = 4096 over the ranks), configs[4]
# (32/16/1024, "1 vs 8 GPUs") and the size where sharding can pay
DEFAULT_RIDERS = ("sharded_k4096_f32", "sharded_s32_k1024_f32", "sharded_k262144_f32")
# deadline of one rider's child job (a rider takes 5-15 s; three riders at their deadlines must stay well inside the few
# minutes a caller gives the whole bench): 80 s, 120 s for the K = 262 144 one
RIDER_DEADLINE_S = float(os.environ.get("GATO_RIDER_DEADLINE", "80"))
LINE_LIMIT = 4096


def dumps_strict(obj, limit=None):
    import importlib
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in _sys.path:
        _sys.path.insert(0, root)
    return importlib.import_module("bench").dumps_strict(obj, limit)


def rider_summary(r):
    """What the last line keeps of one sharded rider: the strong-scaling numbers where a fixed-key parser
    sees them - us per iteration of the sharded launch, the same system on ONE GPU, the transport that ran."""
    if not isinstance(r, dict) or "error" in r or "value" not in r:
        return {"error": str((r or {}).get("error", "no result"))[:120]}
    par, cfg = r.get("parity") or {}, r.get("config") or {}
    one_ips = par.get("same_system_on_one_gpu_iters_per_s")
    return {"transport": cfg.get("transport"), "mirror_memory": cfg.get("mirror_memory"),
            "fallback_reason": (cfg.get("transport_fallback_reason") or "")[:100] or None,
            "knots_per_gpu": cfg.get("knots_per_gpu"), "workgroups_per_gpu": cfg.get("pcg_workgroups_per_gpu"),
            "iters_per_s": r["value"], "us_per_iter": r.get("pcg_us_per_iter"), "ms_per_step": r.get("ms_per_step"),
            "one_gpu_iters_per_s": one_ips, "one_gpu_us_per_iter": par.get("same_system_on_one_gpu_pcg_us_per_iter"),
            "one_gpu_kernel": par.get("same_system_on_one_gpu_kernel"),
            "speedup_vs_one_gpu": (r["value"] / one_ips) if one_ips else None,
            "lam_rel_err_vs_one_gpu": par.get("lam_rel_err_vs_single_gpu"), "iters": par.get("iters")}


LEAD = "sharded_k4096_f32"           # BASELINE configs[3]: the largest BASELINE system that shards - what --gpus N leads with


def attach_riders(rep, riders):
    """The last line of `bench.py --gpus N`: LED BY THE SHARDED SYSTEM - `value` = PCG iterations/s of ONE
    14/7/4096 system (BASELINE configs[3]) knot-sharded over the N GPUs, whole steps, "scaling": "strong" - with the same
    system on ONE GPU beside it (`one_gpu_value`, `sharded_speedup`), so that a SCALE record answers north_star's question
    directly.  The N replicas of the N = 1 workload (configs[1], K = 50: does not shard, reads ~N x on any node) ride along
    under `replicas`, the other sharded shapes under `config.sharded`.  If the lead rider did not deliver (its child job
    faulted, hung or was refused), the replicas line is the headline, unchanged, with the reason."""
    lead = riders.get(LEAD)
    summaries = {name: rider_summary(r) for name, r in riders.items()}
    k4 = summaries.get(LEAD, {})
    if not isinstance(lead, dict) or "error" in lead or "value" not in lead:
        out = rep
        out["config"]["sharded"] = summaries
        out["sharded_speedup"] = None
        out["lead_error"] = f"{LEAD} did not deliver: the replicas value leads instead"
        return out
    out = {"metric": "PCG iterations/s", "value": lead["value"], "unit": "iterations/s", "n_gpus": rep["n_gpus"],
           "steps": rep["steps"], "warmup": rep["warmup"], "ms_per_step": lead["ms_per_step"], "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": lead["dtype"], "data": "synthetic", "config": dict(lead["config"]),
           "roofline": lead.get("roofline"), "pcg_us_per_iter": lead.get("pcg_us_per_iter"), "out_of_loop_ms": lead.get("out_of_loop_ms"),
           "one_gpu_value": k4.get("one_gpu_iters_per_s"), "one_gpu_us_per_iter": k4.get("one_gpu_us_per_iter"),
           "sharded_speedup": k4.get("speedup_vs_one_gpu"),
           "sharded_speedup_workload": f"{LEAD} (strong scaling: one 14/7/4096 system over the ranks vs the same system on one GPU)",
           "single_reduction": lead.get("single_reduction"),
           "replicas": {"value": rep["value"], "unit": rep["unit"], "scaling": "weak", "ms_per_step": rep["ms_per_step"],
                        "workload": rep["config"]["workload"], "roofline_frac": (rep.get("roofline") or {}).get("frac"),
                        "note": "N independent configs[1] systems, one per GPU, no data-path collective (that shape does not shard)"}}
    out["config"]["sharded"] = {n_: v for n_, v in summaries.items() if n_ != LEAD}
    out["config"]["lam_rel_err_vs_one_gpu"] = k4.get("lam_rel_err_vs_one_gpu")
    return out


def main(args):
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if ONE_GPU else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    torch.cuda.set_device(local)
    if ONE_GPU:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    wl = args.workload or ""
    rider_out = getattr(args, "out", None)
    if wl.startswith("batched"):
        return main_batched(args, torch, dist, rank, local, world)
    if wl in WORKLOADS:                                           # the knot-sharded solve as the line itself (or a rider's child job)
        out = sharded_leg(args, torch, dist, rank, local, world, wl, args.steps, args.warmup)
        if rider_out and rank == 0:
            with open(rider_out + ".tmp", "w") as f:
                f.write(dumps_strict({k: out.get(k) for k in RIDER_KEYS}))
            os.replace(rider_out + ".tmp", rider_out)
            out = None
    else:
        out = replicas_leg(args, torch, dist, rank, local, world)
        # riders: child jobs (see rider_in_child) unless asked otherwise; a rehearsal on one GPU with more than 3 ranks
        # would put more processes on the card than a box allows, so it keeps them in this process
        inproc = os.environ.get("GATO_BENCH_RIDERS_INPROC") == "1" or (ONE_GPU and world > 3)
        if rank == 0:
            # the replicas line NOW, before any rider starts: if the riders together outlast the caller's own limit, the last
            # complete JSON line on stdout is still a valid line (the final line below is led by the sharded system)
            print(dumps_strict(out, LINE_LIMIT), flush=True)
        riders = {}
        for rider in DEFAULT_RIDERS:
            if rider == "sharded_k262144_f32" and world < 2:
                continue                                          # one GPU runs it through the streaming kernels: not a sharding number
            big = rider == "sharded_k262144_f32"
            # the LEAD rider is the line's `value`: exactly the K steps / W warm-up steps the caller asked for
            st, wu = (3, 1) if big else ((args.steps, args.warmup) if rider == LEAD else (min(args.steps, 20), min(args.warmup, 3)))
            if not inproc:
                res = rider_in_child(dist, rank, world, rider, st, wu, RIDER_DEADLINE_S * (1.5 if big or rider == LEAD else 1.0))
                if rank == 0:
                    riders[rider] = res
                continue
            try:        # the line above must survive whatever happens in a rider (an error raised on every rank alike)
                sh = sharded_leg(args, torch, dist, rank, local, world, rider, st, wu)
                if rank == 0:
                    riders[rider] = {k: sh.get(k) for k in RIDER_KEYS}
            except Exception as e:   # noqa: BLE001
                if rank == 0:
                    riders[rider] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if rank == 0:
            for name, r in riders.items():                        # the full rider objects: EARLIER stdout lines, never the last
                print(dumps_strict({"rider": name, "result": r}), flush=True)
            out = attach_riders(out, riders)
    if rank == 0 and out is not None:
        try:
            line = dumps_strict(out, LINE_LIMIT)
        except ValueError:              # never lose the headline to its riders: drop them, largest first
            for key in ("sharded", "parallelism", "step"):
                if isinstance(out.get("config"), dict):
                    out["config"].pop(key, None)
            out.pop("single_reduction", None)
            line = dumps_strict(out, LINE_LIMIT)
        print(line, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="child job of bench.py --gpus N: one knot-sharded rider")
    ap.add_argument("--rider", required=True, choices=sorted(WORKLOADS))
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    a.workload = a.rider
    if os.environ.get("GATO_RIDER_TEST_HANG") == "1":             # tests/test_dist_gloo.py: the parent's deadline must end this
        time.sleep(3600)
    main(a)
