#!/usr/bin/env python3
"""bench.py - PCG iterations/s of the gato hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--no-sweep] [--no-cpu] [--sweep-out FILE]

A step = one pass of the hot path over one synthetic KKT system already resident in HBM:
CSR->dense scatter, Schur/preconditioner assembly, PCG with exit_tol = 0 (exactly max_iters = 100
iterations, BASELINE.json configs[1]) and the dz back-substitution - gato_linsys of the reference
(gpu_library.cu:25-83) through the C ABI (gato_linsys_device).  value = PCG iterations / second
over the whole step.  `roofline` is for the dominant kernel (the PCG launch), timed with HIP events
recorded on the launch stream right around it: ALGORITHMIC bytes per iteration
B_iter = [(6K-4) S^2 + 13 S K] w x iterations / launch time, against 8 TB/s.
`cpu_baseline` = the C restatement of the same step (oracle/, "port") on the host cores.

OUTPUT: the LAST stdout line is ONE compact strict-JSON object (< 4 KB: metric, value, config, roofline incl. traffic /
limiter / latency_floor numbers, cpu_baseline, a one-record-per-entry sweep summary) - the line the driver parses.  Every
sweep entry is printed in full on an EARLIER line ({"sweep_entry": ...}) as it finishes, and the whole sweep plus the method
notes go to --sweep-out (default gpurun_out/bench_sweep.json).  A sweep entry that raises cannot take the headline down.

N = 1 runs BASELINE.json configs[1] (IIWA 14/7, K = 50, fp64).  N > 1 (gato_python_amd/dist_bench.py): `value` is
the same workload on every rank, each rank its own system - "replicas only", no data-path collective (one K = 50 system
is a single workgroup and cannot shard) - and the knot-sharded solves that do exchange data ride along in `config.sharded`
of the same line (configs[3]: K = 4096 split over the ranks; configs[4]: 32/16/1024; K = 262144), each with its us/iteration
beside the same system on one GPU and the transport it ran on (in-kernel xGMI peer stores, or the RCCL fallback).
`--workload sharded_*` makes a sharded solve the line itself; `--workload batched_512x_f64` (any N) runs independent
batches per rank (weak scaling, no collective).  `--gpus N` without a torch.distributed.run environment starts the N
ranks itself.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.3 measured copy)

WORKLOADS = {
    # name: (S, C, K, dtype, BASELINE.json config)
    "iiwa_14_7_k50_f64": (14, 7, 50, np.float64, "configs[1]"),
    "iiwa_14_7_k50_f32": (14, 7, 50, np.float32, "configs[1] in the reference's fp32"),
    "iiwa_14_7_k512_f32": (14, 7, 512, np.float32, "configs[2]"),
    "iiwa_14_7_k4096_f32": (14, 7, 4096, np.float32, "configs[3] on one GPU"),
    "iiwa_14_7_k4096_f64": (14, 7, 4096, np.float64, "configs[3] on one GPU, fp64"),
    "s32_c16_k1024_f32": (32, 16, 1024, np.float32, "configs[4]"),
    # beyond register residency: semi-resident persistent launch / streaming kernels
    "iiwa_14_7_k16384_f32": (14, 7, 16384, np.float32, "K just beyond the register file (matrices 77 MB: L2 / Infinity Cache)"),
    # genuinely HBM-bound: matrices 1.2 GB > 256 MB Infinity Cache
    "iiwa_14_7_k131072_f32": (14, 7, 131072, np.float32, "K far beyond residency (HBM-roofline run)"),
    "s32_c16_k32768_f32": (32, 16, 32768, np.float32, "configs[4]'s shape far beyond residency (S + Pinv 805 MB: HBM-bound)"),
    # fp64 beyond residency: S + Pinv 617 MB; LDS-DMA ring (pcg_semi = 3), semi-resident (auto), streaming kernels
    "iiwa_14_7_k65536_f64": (14, 7, 65536, np.float64, "fp64 far beyond residency (S + Pinv 617 MB: HBM-bound)"),
}
MAX_ITERS = 100
PCG_VARIANT = 0     # 1 = opt-in single-reduction (Chronopoulos-Gear) resident kernel, sweep entries only
PCG_SEMI = None     # force a variant of the persistent launch for K beyond the register file (option pcg_semi), sweep entries only


def b_iter(S, K, w):
    return ((6 * K - 4) * S * S + 13 * S * K) * w


def dtype_name(dt):
    return "f64" if np.dtype(dt) == np.float64 else "f32"


def run_single(name, steps, warmup, torch, pcg_mode=None, pcg_reps=20, max_iters=None, variant=0, min_seconds=0.0, pcg_semi=None):
    global MAX_ITERS, PCG_VARIANT, PCG_SEMI
    PCG_VARIANT = variant
    PCG_SEMI = pcg_semi
    saved = MAX_ITERS
    if max_iters is not None:
        MAX_ITERS = max_iters
    try:
        return _run_single(name, steps, warmup, torch, pcg_mode, pcg_reps, min_seconds)
    finally:
        MAX_ITERS = saved


def latency_floor(sol, bufs, lam, production_us):
    """What bounds a register-resident launch is not HBM but the dependent chain of an iteration.  Measured live with
    the timing-only switches of the diagnostic kernel build (option `ablate`; the results of these launches are
    garbage and are not used): the time of the two block-tridiagonal products alone and of the two reductions /
    hand-offs alone.  Their sum is the floor an iteration could reach if everything else (vector exchange barriers,
    the scalar tail, the skew between waves) cost nothing."""
    def us(abl):
        sol.set_option("ablate", abl)
        ms = []
        for i in range(10):
            sol.pcg(bufs[0], bufs[1], bufs[2], 0.0, MAX_ITERS, lam=lam, check=False)
            if i >= 2:
                ms.append(sol.pcg_last_ms())
        return 1e3 * float(np.median(ms)) / MAX_ITERS
    try:
        sol.set_option("stamp_pcg", 2)        # the build with the timing-only switches and no cycle stamps
        full, no_spmv, no_red, nothing = us(0), us(3), us(4), us(15)
    finally:
        sol.set_option("ablate", 0)
        sol.set_option("stamp_pcg", 0)
    spmv, red = max(full - no_spmv, 0.0), max(full - no_red, 0.0)
    floor = spmv + red
    out = {"us_per_iteration": floor, "products_us": spmv, "reductions_and_handoffs_us": red,
           "diagnostic_build_full_us": full, "loop_skeleton_us": nothing, "production_us_per_iteration": production_us,
           "frac_of_floor": floor / full if full > 0 else None}
    W = sol.get_option("last_groups")
    if W > 1:
        # what one all-to-all round of W workgroups costs with no arithmetic at all (tools/micro/pingpong.hip, MI355X)
        pp = 0.42 if W <= 2 else 0.59 if W <= 15 else 0.84 if W <= 32 else 1.07
        out["handoff_floor_us"] = pp
        out["us_per_iteration_with_handoff_floor"] = spmv + 2 * pp
    return out


def _run_single(name, steps, warmup, torch, pcg_mode=None, pcg_reps=20, min_seconds=0.0):
    from gato_python_amd import synth
    from gato_python_amd.solver import Solver
    S, C, K, dt, _ = WORKLOADS[name]
    sysm = synth.make_system(S, C, K, seed=0)
    sol = Solver(S, C, K, dt)
    if pcg_mode is not None:
        sol.set_option("pcg_mode", pcg_mode)
    if PCG_VARIANT:
        sol.set_option("pcg_variant", PCG_VARIANT)
    if PCG_SEMI is not None:
        sol.set_option("pcg_semi", PCG_SEMI)
    dev = sol.upload_system(sysm)
    lam, dz = sol.new(S * K), sol.new(sol.N)

    def step():
        sol.linsys(*dev, 0.0, MAX_ITERS, sysm.rho, lam, dz)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    sol.check_status()
    # exactly `steps` steps between synchronisations = one block; blocks are repeated until min_seconds of timed work
    # have accumulated (the driver's --steps 20 is 5 ms of this workload: too thin a sample for one number)
    dt_s, blocks = 0.0, 0
    while True:
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt_s += time.perf_counter() - t0
        blocks += 1
        if dt_s >= min_seconds or blocks >= 1000:
            break
    steps_timed = steps * blocks
    sol.check_status()

    # dominant kernel: the PCG launch AS IT RUNS IN THE STEP (one-workgroup launches also carry the dz back-substitution
    # in their epilogue), HIP events recorded on its stream right around the launch (gato_pcg_last_ms)
    sol.set_option("time_pcg", 1)
    bufs = [sol.buffer_ptr(i) for i in (3, 4, 5)]
    ms = []
    for i in range(pcg_reps + 3):
        step()
        v = sol.pcg_last_ms()
        if i >= 3:
            ms.append(v)
    pcg_ms = float(np.mean(ms))
    bytes_launch = b_iter(S, K, np.dtype(dt).itemsize) * MAX_ITERS
    mode = sol.get_option("last_mode")
    mode_name = {1: "resident", 2: "streaming"}.get(mode) + {0: "", 1: " (semi)", 2: " (semi, no resident rows)",
                                                             3: " (LDS-DMA ring)"}[sol.get_option("last_semi")] + \
        {0: "", 1: " (one workgroup, 2 rows/lane)", 2: " (one workgroup, mixed 2/1 rows per lane)"}[sol.get_option("last_pair")]
    groups, threads = sol.get_option("last_groups"), sol.get_option("last_threads")
    floor = None
    # the diagnostic builds (timing-only switches) exist for pcg_resident_kernel and the fp64 mixed-rows kernel (last_pair = 2)
    if mode == 1 and not PCG_VARIANT and sol.get_option("last_pair") in (0, 2) and not sol.get_option("last_semi"):
        floor = latency_floor(sol, bufs, lam, 1e3 * pcg_ms / MAX_ITERS)
    sol.set_option("time_pcg", 0)
    res = dict(
        workload=name, S=S, C=C, K=K, dtype=dtype_name(dt),
        iters_per_s=MAX_ITERS * steps_timed / dt_s, ms_per_step=1e3 * dt_s / steps_timed, timed_steps=steps_timed,
        latency_floor=floor,
        pcg_launch_ms=pcg_ms, pcg_launch_ms_min=float(np.min(ms)),
        pcg_iters_per_s=MAX_ITERS / (pcg_ms * 1e-3), pcg_us_per_iter=1e3 * pcg_ms / MAX_ITERS,
        pcg_mode=mode_name, pcg_groups=groups, pcg_threads=threads,
        algorithmic_bytes_per_launch=bytes_launch,
        achieved_gbs=bytes_launch / (pcg_ms * 1e-3) / 1e9,
    )
    sol.close()
    return res, sysm


def run_batched(S, C, K, dt, B, steps, warmup, torch):
    """B independent systems per call (one workgroup per system in the PCG launch)."""
    from gato_python_amd import synth
    from gato_python_amd.solver import Solver
    base = synth.make_system(S, C, K, seed=0)
    systems = [base] * B                       # same values in every system: timing only
    sol = Solver(S, C, K, dt, batch=B)
    dev = sol.upload_batch(systems)
    lam, dz = sol.new(B * S * K), sol.new(B * sol.N)
    iters = sol.new(B, torch.int32)
    step = lambda: sol.linsys_batched(*dev, 0.0, MAX_ITERS, base.rho, lam, dz, iters)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    sol.set_option("time_pcg", 1)
    bufs = [sol.buffer_ptr(i) for i in (3, 4, 5)]
    ms = []
    for i in range(8):
        sol.pcg(bufs[0], bufs[1], bufs[2], 0.0, MAX_ITERS, lam=lam, iters=iters, check=False)
        ms.append(sol.pcg_last_ms())
    pcg_ms = float(np.mean(ms[2:]))
    w = np.dtype(dt).itemsize
    bytes_launch = b_iter(S, K, w) * MAX_ITERS * B
    r = dict(workload=f"batched_{B}x_{S}_{C}_k{K}_{dtype_name(dt)}", S=S, C=C, K=K, dtype=dtype_name(dt), batch=B,
             iters_per_s=MAX_ITERS * B * steps / el, ms_per_step=1e3 * el / steps, pcg_launch_ms=pcg_ms,
             pcg_iters_per_s=MAX_ITERS * B / (pcg_ms * 1e-3), pcg_mode="resident, one workgroup per system",
             algorithmic_bytes_per_launch=bytes_launch, achieved_gbs=bytes_launch / (pcg_ms * 1e-3) / 1e9)
    r["bound"] = "latency / issue inside one CU per system (matrices register-resident: HBM bytes per launch = the matrices once)"
    r["algorithmic_over_hbm_peak"] = r["achieved_gbs"] / HBM_PEAK_GBS     # > 1 = finishes sooner than HBM could stream the matrices per iteration
    sol.close()
    return r


def cpu_baseline(sysm, dt, budget_s=10.0):
    """The C restatement (oracle/, test infrastructure) timed on the host as the reported CPU baseline."""
    from oracle import c_oracle as co
    S, C, K = sysm.S, sysm.C, sysm.K
    best, tried = None, []
    # thread counts tried: 1 and the box's CPU share (the GPU boxes expose all host cores but grant ~16 per GPU;
    # oversubscribing the OpenMP loops beyond that only slows them down)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    for threads in sorted({1, min(co.max_threads(), avail, 16)}):
        co.set_threads(threads)
        co.linsys_solve(*sysm.csr_args(), S, C, K, 0.0, MAX_ITERS, sysm.rho, dtype=dt)   # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            co.linsys_solve(*sysm.csr_args(), S, C, K, 0.0, MAX_ITERS, sysm.rho, dtype=dt)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s / 2 or n >= 20000:
                break
        v = MAX_ITERS * n / el
        tried.append({"cores": threads, "value": v})
        if best is None or v > best["value"]:
            best = dict(value=v, unit="PCG iterations/s", cores=threads, kind="port",
                        sample=f"{n} whole solves (assembly + {MAX_ITERS} PCG iterations + dz) of the same "
                               f"{S}/{C}/{K} {dtype_name(dt)} system in {el:.1f} s, oracle/gato_oracle_impl.h")
    best["tried"] = tried                 # every thread count timed (1 and the box's CPU share); value = the best of them
    return best


def scipy_cg_baseline(sysm, dt):
    """Secondary CPU sanity baseline: scipy.sparse.linalg.cg on the assembled -S with -Pinv as the
    preconditioner (both positive definite), same right-hand side, MAX_ITERS iterations."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    from oracle import gato_oracle as o
    S, C, K = sysm.S, sysm.C, sysm.K
    out = o.linsys_solve(*sysm.csr_args(), S, C, K, 0.0, 1, sysm.rho, dtype=dt, return_all=True)

    def bd_to_csr(bd):
        L, M, R = o.unpack_bd(bd, S, K)
        blocks = [[None] * K for _ in range(K)]
        for k in range(K):
            blocks[k][k] = -M[k]
            if k > 0:
                blocks[k][k - 1] = -L[k]
            if k < K - 1:
                blocks[k][k + 1] = -R[k]
        return sp.bmat(blocks, format="csr")
    A, Mi, b = bd_to_csr(out["S"]), bd_to_csr(out["Pinv"]), -out["gamma"]
    its = [0]

    def cb(_):
        its[0] += 1
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 2.0:
        its[0] = 0
        spl.cg(A, b, rtol=0.0, atol=0.0, maxiter=MAX_ITERS, M=Mi, callback=cb)
        n += 1
    el = time.perf_counter() - t0
    return {"value": its[0] * n / el, "unit": "PCG iterations/s", "cores": 1, "kind": "scipy.sparse.linalg.cg on -S, M = -Pinv (CSR)",
            "sample": f"{n} runs of {its[0]} iterations in {el:.1f} s"}


def committed_traffic(name, res=None):
    """HBM bytes per PCG launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json) with their provenance, or
    (None, reason): the entry is refused when the kernel or launch geometry it was collected on is not what ran now."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(p):
        return None, "profiles/pmc_traffic.json missing"
    try:
        import hashlib
        raw = open(p, "rb").read()
        blob = hashlib.sha1(b"blob %d\0" % len(raw) + raw).hexdigest()        # = git hash-object
        e = json.loads(raw).get(name)
    except Exception as ex:       # noqa: BLE001
        return None, f"unreadable: {ex}"
    if not e:
        return None, "no entry for this workload"
    src = {"file": "profiles/pmc_traffic.json", "git_blob": blob, "kernel": e.get("kernel"), "grid_threads": e.get("grid_threads"),
           "collected_by": "tools/profile.sh + tools/summarize_profile.py (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"}
    if res is not None and res.get("pcg_groups"):
        want = res["pcg_groups"] * res["pcg_threads"]
        fam = "double" if res["dtype"] == "f64" else "float"
        k = e.get("kernel", "")
        g_ = e.get("grid_threads") or 0
        helpers = want > 0 and g_ % want == 0 and (g_ // want - 1) % 8 == 0    # one-workgroup launches of ONE system bring 8 x h helper blocks
        if g_ not in (want, 8 * want) and not helpers:                      # one-XCD launches use an 8x oversubscribed grid
            return None, f"stale: collected on a grid of {e.get('grid_threads')} threads, this run launched {want}"
        single = ("pcg_single_f32x2" in k and fam == "float") or ("pcg_single_f64m" in k and fam == "double")
        if not single and (fam + ", " + str(res["S"])) not in k:
            return None, f"stale: collected on {k}"
    return e.get("hbm_bytes_per_launch"), src


def annotate(r, name=None):
    """Roofline fields of a sweep entry.  An HBM fraction is printed only where the launch really streams its matrices
    every iteration (measured HBM bytes >= 70 % of the algorithmic bytes); register-resident launches read them once
    per LAUNCH, so for them the ratio of measured to algorithmic bytes and the latency floor are what is reported."""
    t, src = committed_traffic(name or r["workload"], r)
    r["hbm_bytes_per_launch_pmc"] = t
    r["traffic_source"] = src
    ratio = (t / r["algorithmic_bytes_per_launch"]) if t else None
    r["hbm_traffic_over_algorithmic"] = ratio
    streams = (ratio is not None and ratio >= 0.7) or "streaming" in str(r.get("pcg_mode", ""))
    if streams:
        r["roofline_frac"] = r["achieved_gbs"] / HBM_PEAK_GBS
        r["bound"] = "hbm"
    else:
        r["bound"] = "latency (matrices register-resident: see latency_floor and hbm_traffic_over_algorithmic)"
    return r


NOTES = {
    "latency_floor": "live: diagnostic builds of the same kernel with timing-only switches - ablate = 3 (no products) / 4 (no "
                     "reductions, no hand-offs) / 15 (loop skeleton); floor = products + reductions; frac_of_floor = floor / "
                     "full.  The one-workgroup fp64 kernel has the switches as compile-time variants (its 'full' IS the "
                     "production kernel); pcg_resident_kernel has them as uniform run-time branches in a build of its own",
    "handoff_floor": "tools/micro/pingpong.hip all-to-all round, one polling wave per workgroup: 0.42 (W=2) / 0.59 (W=15) / "
                     "0.84 (W=29) us on one XCD, 1.03-1.10 us chip-wide",
    "traffic": "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile.sh + "
               "tools/summarize_profile.py), 2*FETCH_SIZE + WRITE_SIZE per launch; refused when kernel family or grid differ from this run",
    "regime": "register/LDS-resident launches read the matrices from HBM once per LAUNCH; an iteration is bound by its two "
              "dependent block reductions (hand-offs for W > 1) and LDS operand reads (DESIGN.md 3.1); the HBM-bound run of "
              "this path is sweep entry iiwa_14_7_k131072_f32",
}
LINE_LIMIT = 4096          # the driver keeps a tail of stdout: the LAST line must be the whole headline object


def _clean(x, digits=6):
    """Strict-JSON values: NaN / inf -> null, floats rounded to `digits` significant digits, numpy scalars -> Python."""
    import math
    if isinstance(x, dict):
        return {str(k): _clean(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_clean(v, digits) for v in x]
    if isinstance(x, (bool, type(None), str, int)):
        return x
    if isinstance(x, np.integer):
        return int(x)
    if isinstance(x, (float, np.floating)):
        x = float(x)
        if not math.isfinite(x):
            return None
        return float(f"{x:.{digits}g}")
    return str(x)


def dumps_strict(obj, limit=None):
    line = json.dumps(_clean(obj), allow_nan=False, separators=(",", ":"))
    if limit is not None and len(line) >= limit:
        raise ValueError(f"bench line is {len(line)} bytes, limit {limit}")
    return line


def roofline_object(res, traffic, traffic_src):
    """The `roofline` object of a line: bound = the roofline that bounds the PATH (HBM: GEMV-shaped iteration); `limiter` =
    what the measured launch actually sits on, from the ratio of measured HBM traffic to algorithmic bytes."""
    alg = res["algorithmic_bytes_per_launch"]
    ratio = (traffic / alg) if traffic else None
    streams = (ratio is not None and ratio >= 0.7) or "streaming" in str(res.get("pcg_mode", ""))
    lf = res.get("latency_floor")
    ro = {"bound": "hbm", "achieved": res["achieved_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
          "frac": res["achieved_gbs"] / HBM_PEAK_GBS, "traffic": traffic,
          "traffic_over_algorithmic": ratio,
          "limiter": "hbm" if streams else "latency (matrices register/LDS-resident: HBM read once per launch)",
          "kernel": "pcg_" + str(res["pcg_mode"]), "launch_ms": res["pcg_launch_ms"],
          "algorithmic_bytes_per_launch": alg, "pcg_only_iterations_per_s": res["pcg_iters_per_s"]}
    if isinstance(traffic_src, dict):
        ro["traffic_source"] = {"file": traffic_src.get("file"), "git_blob": (traffic_src.get("git_blob") or "")[:12],
                                "kernel": str(traffic_src.get("kernel"))[:80], "grid_threads": traffic_src.get("grid_threads")}
    else:
        ro["traffic_source"] = str(traffic_src)[:120]
    if lf:
        ro["latency_floor"] = {k: lf[k] for k in ("us_per_iteration", "products_us", "reductions_and_handoffs_us",
                                                   "production_us_per_iteration", "frac_of_floor", "handoff_floor_us",
                                                   "us_per_iteration_with_handoff_floor") if k in lf}
    return ro


def headline(res, name, steps, warmup, traffic, traffic_src, cpu=None, sweep_file=None):
    """The ONE line the driver parses (last line of stdout, < LINE_LIMIT bytes, strict JSON).  Everything verbose - the
    sweep, the method notes - is printed on earlier lines and written to `sweep_file`."""
    S, C, K, dt, cfg = WORKLOADS[name]
    out = {
        "metric": "PCG iterations/s", "value": res["iters_per_s"], "unit": "iterations/s",
        "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": res["ms_per_step"],
        "timed_steps": res["timed_steps"],      # blocks of exactly `steps` steps, repeated until 1 s of timed work
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": res["dtype"], "data": "synthetic",
        "config": {"workload": name, "baseline_config": cfg, "STATE_SIZE": S, "CONTROL_SIZE": C, "KNOT_POINTS": K,
                   "max_iters": MAX_ITERS, "exit_tol": 0.0,
                   "step": "convert + Schur/stair assembly + PCG(100 iterations) + dz, inputs resident in HBM",
                   "pcg_kernel": res["pcg_mode"], "pcg_workgroups": res["pcg_groups"],
                   "pcg_threads": res["pcg_threads"]},
        "roofline": roofline_object(res, traffic, traffic_src),
    }
    if cpu is not None:
        out["cpu_baseline"] = {k: cpu[k] for k in ("value", "unit", "cores", "kind", "sample") if k in cpu}
        if "tried" in cpu:
            out["cpu_baseline"]["tried"] = cpu["tried"]
        sec = cpu.get("secondary")
        if isinstance(sec, dict) and "value" in sec:
            out["cpu_baseline"]["scipy_cg_value"] = sec["value"]
    if sweep_file:
        out["sweep_file"] = sweep_file
    return out


def sweep_summary(entry):
    """One short record per sweep entry for the headline line (the full entries are earlier stdout lines + the sweep file).
    Short keys, four significant digits - two dozen entries must fit the 4 KB line: us = us per PCG iteration, ips = iterations/s
    over whole steps, and the entry's ROOF: hbm = fraction of the 8 TB/s HBM roofline where the launch really streams its
    matrices every iteration, else (register-resident launches, whose algorithmic GB/s may exceed the HBM peak) gbs = algorithmic
    GB/s with floor = measured latency floor / production time (products + reductions / hand-offs of the same kernel, live
    ablation) where the kernel has the timing-only switches."""
    def sig(x):
        return float(f"{x:.4g}") if isinstance(x, (int, float)) and x == x else x
    r = {"us": sig(entry.get("pcg_us_per_iter")), "ips": sig(entry.get("iters_per_s"))}
    if entry.get("roofline_frac") is not None:
        r["hbm"] = sig(entry["roofline_frac"])
    elif entry.get("achieved_gbs") is not None:
        r["gbs"] = sig(entry["achieved_gbs"])
        lf = entry.get("latency_floor")
        if isinstance(lf, dict) and lf.get("frac_of_floor") is not None:
            r["floor"] = sig(lf["frac_of_floor"])
    return r


def run_sweep(args, torch, emit):
    """The other single-GPU shapes; every finished entry goes to emit() at once (its own stdout line + the sweep file)."""
    def aux(other, **kw):
        r, _ = run_single(other, max(10, args.steps // 10), 3, torch, **kw)
        return r
    for other in ("iiwa_14_7_k50_f32", "iiwa_14_7_k512_f32", "iiwa_14_7_k4096_f32", "iiwa_14_7_k4096_f64",
                  "s32_c16_k1024_f32"):
        r, r2 = aux(other), aux(other)                 # auxiliary entries: best of two short runs
        if r2["iters_per_s"] > r["iters_per_s"]:
            r = r2
        annotate(r)
        if not args.no_cpu:
            So, Co, Ko, dto, _ = WORKLOADS[other]
            from gato_python_amd import synth as _synth
            r["cpu_baseline"] = cpu_baseline(_synth.make_system(So, Co, Ko, seed=0), dto, budget_s=3.0)
        emit(r)
    # opt-in single-reduction variant (one hand-off per iteration; rounding differs from the reference recurrence)
    for other in ("iiwa_14_7_k512_f32", "iiwa_14_7_k4096_f32", "s32_c16_k1024_f32"):
        r = aux(other, variant=1)
        annotate(r, other + "_single_reduction_variant")
        r["workload"] += "_single_reduction_variant"
        emit(r)
    # SURVEY.md 8d run 3: the K=512 system through the STREAMING kernels (matrices re-read every iteration; they fit L2,
    # so the PMC passes show how little of that reaches HBM) beside the register-resident entry above
    r = aux("iiwa_14_7_k512_f32", pcg_mode=2)
    annotate(r, "iiwa_14_7_k512_f32_streaming")
    r["workload"] += "_streaming"
    emit(r)
    # batches of independent systems: throughput mode of the K=50 shape
    emit(run_batched(14, 7, 50, np.float64, 512, 10, 2, torch))
    emit(run_batched(14, 7, 50, np.float32, 512, 10, 2, torch))
    r, _ = run_single("iiwa_14_7_k16384_f32", 5, 2, torch, pcg_reps=5)
    emit(annotate(r))
    # HBM-bound regime (matrices 1.2 GB): 20 iterations per solve keep the run short.  Auto = the persistent launch
    # whose block rows stream through an LDS-DMA ring (gato_pcg_dma.hip: one workgroup per CU, all vectors in
    # registers); beside it the semi-resident persistent launch (7 % of the block rows in registers, the rest re-read
    # by plain loads every product) and the streaming kernels (two launches per iteration, LDS-DMA tiles).
    for mode, semi, tag in ((None, None, ""), (None, 1, "_semi"), (2, None, "_streaming")):
        r, _ = run_single("iiwa_14_7_k131072_f32", 3, 1, torch, pcg_mode=mode, pcg_reps=5, max_iters=20, pcg_semi=semi)
        annotate(r, "iiwa_14_7_k131072_f32" + tag)
        r["workload"] += tag
        r["max_iters"] = 20
        emit(r)
    # fp64 in the same regime: auto = the ring with 8-byte slots (from 550 MB of S + Pinv: measured cross-over against the
    # semi-resident launch, profiles/r05_ring_crossover.log), beside it the semi-resident launch and the streaming kernels
    for mode, semi, tag in ((None, None, ""), (None, 1, "_semi"), (2, None, "_streaming")):
        r, _ = run_single("iiwa_14_7_k65536_f64", 3, 1, torch, pcg_mode=mode, pcg_reps=5, max_iters=20, pcg_semi=semi)
        annotate(r, "iiwa_14_7_k65536_f64" + tag)
        r["workload"] += tag
        r["max_iters"] = 20
        emit(r)
    # the same regime at configs[4]'s shape: LDS-DMA ring (auto) beside the semi-resident launch
    for semi, tag in ((None, ""), (1, "_semi")):
        r, _ = run_single("s32_c16_k32768_f32", 3, 1, torch, pcg_reps=5, max_iters=20, pcg_semi=semi)
        annotate(r, "s32_c16_k32768_f32" + tag)
        r["workload"] += tag
        r["max_iters"] = 20
        emit(r)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None)
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--pcg-mode", type=int, default=None)
    ap.add_argument("--sweep-out", default=os.path.join("gpurun_out", "bench_sweep.json"),
                    help="file that receives the sweep entries and the method notes (the last stdout line stays compact)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # started by hand without the launcher: start the N ranks as a child job (nothing has touched the GPU yet) and
        # pass its exit code on - never a silent one-rank run
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if world > 1 and args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    import torch
    if args.gpus > 1 or world > 1 or (args.workload or "").startswith(("sharded", "batched", "replicas")):
        from gato_python_amd import dist_bench
        return dist_bench.main(args)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(0)
    name = args.workload or "iiwa_14_7_k50_f64"
    S, C, K, dt, cfg = WORKLOADS[name]
    res, sysm = run_single(name, args.steps, args.warmup, torch, args.pcg_mode, min_seconds=1.0)
    traffic, traffic_src = committed_traffic(name, res)
    cpu = None
    if not args.no_cpu:
        cpu = cpu_baseline(sysm, dt)
        try:
            cpu["secondary"] = scipy_cg_baseline(sysm, dt)
        except Exception as ex:       # noqa: BLE001
            cpu["secondary"] = {"error": f"{type(ex).__name__}: {ex}"[:200]}
    sweep, sweep_err = [], None
    do_sweep = not args.no_sweep and args.workload is None

    def emit(entry):
        sweep.append(entry)
        print(dumps_strict({"sweep_entry": entry}), flush=True)          # an EARLIER stdout line, never the last one

    if do_sweep:
        try:
            run_sweep(args, torch, emit)
        except Exception as ex:       # noqa: BLE001  - the headline below must be printed whatever a sweep entry does
            sweep_err = f"{type(ex).__name__}: {ex}"[:300]
    sweep_file = None
    if do_sweep:
        try:
            os.makedirs(os.path.dirname(args.sweep_out) or ".", exist_ok=True)
            with open(args.sweep_out, "w") as f:
                f.write(dumps_strict({"headline_workload": name, "headline": res, "cpu_baseline": cpu, "sweep": sweep,
                                      "sweep_error": sweep_err, "notes": NOTES}))
            sweep_file = args.sweep_out
        except OSError:
            pass
    out = headline(res, name, args.steps, args.warmup, traffic, traffic_src, cpu, sweep_file)
    if do_sweep:
        out["sweep_keys"] = "us=us/PCG iteration, ips=iterations/s over whole steps, hbm=fraction of 8 TB/s (launch streams its matrices), gbs=algorithmic GB/s (register-resident), floor=latency floor/production"
        out["sweep"] = {e["workload"]: sweep_summary(e) for e in sweep}
        if sweep_err:
            out["sweep_error"] = sweep_err[:160]
    try:
        line = dumps_strict(out, LINE_LIMIT)
    except ValueError:                 # never lose the headline to its riders
        out.pop("sweep", None)
        out.pop("sweep_keys", None)
        line = dumps_strict(out, LINE_LIMIT)
    print(line, flush=True)


if __name__ == "__main__":
    main()
