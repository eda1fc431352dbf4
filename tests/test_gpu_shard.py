"""GPU suite: the HIP shard kernels (gato_shard_pcg_* in include/gato_hip.h) against the oracle.  A
1-GPU box cannot host several RCCL ranks, so R shards live in one process and the all-gather is a
concatenation (gato_python_amd.dist.run_lockstep); the collective schedule itself is covered by the gloo
tests, and a world-size-1 RCCL run of bench.py's sharded leg checks the torch.distributed plumbing."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
from gato_python_amd import synth                                   # noqa: E402
from gato_python_amd.dist import HipShardBackend, run_lockstep      # noqa: E402
from oracle import c_oracle as co                                   # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("S,C,K,R,dt", [(14, 7, 50, 1, np.float64), (14, 7, 50, 2, np.float64), (14, 7, 50, 3, np.float32),
                                        (14, 7, 64, 8, np.float64), (2, 1, 9, 5, np.float64), (32, 16, 24, 4, np.float64),
                                        (14, 7, 4096, 8, np.float32), (14, 7, 4096, 8, np.float64), (32, 16, 1024, 8, np.float64)])
def test_shard_kernels_lockstep(S, C, K, R, dt):
    from gato_python_amd.solver import Solver
    s = synth.make_system(S, C, K, seed=13)
    Gd, Cd = co.convert(*s.csr_args()[:6], S, C, K, s.rho, dt)
    Sb, Pb, gam, _ = co.form_schur(Gd, Cd, s.g, s.c, S, C, K)
    Pb = co.form_ss(Sb, Pb, S, K)
    f64 = dt == np.float64
    tol, mi = (1e-9, 150) if f64 else (1e-4, 60)
    lam_o, it_o = co.pcg(Sb, Pb, gam, S, K, tol, mi)
    sols = [Solver(S, C, K, dt) for _ in range(R)]
    dS, dP, dg = sols[0].to_device(Sb), sols[0].to_device(Pb), sols[0].to_device(gam)
    bes = [HipShardBackend(sols[r], r, R, dS, dP, dg, tol, mi) for r in range(R)]
    lam, iters = run_lockstep(bes, mi)
    torch.cuda.synchronize()
    its = [int(i.cpu()[0]) for i in iters]
    assert len(set(its)) == 1 and abs(its[0] - it_o) <= (0 if f64 else 2), (its, it_o)
    if f64:
        err = np.abs(lam.cpu().numpy() - lam_o).max() / np.abs(lam_o).max()
        assert err < 1e-9, err
    else:           # fp32: measured against the converged fp64 solution of the same matrices (tests/f32_parity.py)
        from f32_parity import check_f32
        truth = co.pcg(Sb.astype(np.float64), Pb.astype(np.float64), gam.astype(np.float64), S, K, 1e-14, 600)[0]
        check_f32(f"{R} lock-step shards {S}/{C}/{K}", lam.cpu().numpy(), lam_o, truth)
    assert all(b.done() for b in bes) == (its[0] < mi)              # the host-side convergence poll agrees
    for x in sols:
        x.close()


def test_bench_sharded_leg_world1():
    """bench.py's multi-GPU leg with one RCCL rank: torch.distributed + HIP shard kernels end to end."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "sharded_k4096_f32",
                        "--steps", "2", "--warmup", "1", "--no-cpu"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["workload"].startswith("sharded")
    assert d["parity"]["lam_rel_err_vs_single_gpu"] < 1e-3      # two fp32 HIP runs of 100 fixed iterations (a plumbing check)


def test_bench_batched_leg_world1():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "batched_512x_f64",
                        "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["scaling"] == "weak" and d["value"] > 1e6 and d["config"]["systems_per_gpu"] == 512


def test_bench_default_multi_gpu_line_world1():
    """What `bench.py --gpus N` prints for N > 1, with one rank: the last line is led by the knot-sharded configs[3] system
    (strong scaling), the replicas of the N = 1 workload and the other sharded shapes ride along."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29535", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "replicas",
                        "--steps", "5", "--warmup", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = r.stdout.strip().splitlines()
    last = lines[-1]
    d = json.loads(last)
    assert len(last) < 4096
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["value"] > 1e4 and d["dtype"] == "f32" and d["steps"] == 5 and d["warmup"] == 2
    cfg = d["config"]
    assert cfg["workload"] == "sharded_k4096_f32" and cfg["transport"] in ("xgmi", "rccl") and cfg["lam_rel_err_vs_one_gpu"] < 1e-3
    assert d["one_gpu_value"] > 0 and d["sharded_speedup"] > 0 and d["pcg_us_per_iter"] > 0
    rep = d["replicas"]
    assert rep["scaling"] == "weak" and rep["value"] > 1e5 and rep["workload"] == "iiwa_14_7_k50_f64" and 0 < rep["roofline_frac"] < 1
    assert cfg["sharded"]["sharded_s32_k1024_f32"]["iters_per_s"] > 0                     # configs[4] rides along too
    # earlier lines: the replicas line on its own (a valid headline should the riders outlast the caller) and the full riders
    first = json.loads([x for x in lines if x.startswith('{"metric"')][0])
    assert first["scaling"] == "weak" and first["config"]["workload"] == "iiwa_14_7_k50_f64"
    full = [json.loads(x) for x in lines[:-1] if x.startswith('{"rider"')]
    assert {x["rider"] for x in full} == set(cfg["sharded"]) | {"sharded_k4096_f32"} and all(x["result"]["scaling"] == "strong" for x in full)
