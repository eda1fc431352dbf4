"""CPU suite: the knot-sharded PCG schedule (gato_python_amd/dist.py) on gloo worlds of 2 and 3 ranks.
The arithmetic behind the schedule is the test-only numpy backend (tests/shard_numpy_backend.py); the
result must equal the single-process oracle PCG: same iteration count, lambda to 1e-10."""
import os
import socket
import subprocess
import sys

import pytest

from gato_python_amd.dist import knot_ranges

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,S,C,K,tol,mi,chk", [(2, 14, 7, 11, 1e-10, 200, 0), (3, 2, 1, 8, 1e-12, 100, 0),
                                                    (2, 2, 1, 2, 1e-12, 50, 0), (2, 14, 7, 16, 0.0, 7, 0),
                                                    (2, 14, 7, 11, 1e-10, 200, 4)])
def test_sharded_pcg_gloo(world, S, C, K, tol, mi, chk):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"],
           os.path.join(ROOT, "tests", "dist_worker.py"), str(S), str(C), str(K), str(tol), str(mi), str(chk)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count(" ok iters=") == world


@pytest.mark.parametrize("world,S,C,K", [(2, 14, 7, 16), (3, 14, 7, 11), (2, 2, 1, 2), (3, 2, 1, 3)])
def test_gather_plan_gloo(world, S, C, K):
    """What follows a sharded solve on N > 1 ranks (linsys_solve_cluster): every rank's lambda / dz rows onto all ranks by ONE
    all-gather of fixed-size records into preallocated buffers - equal and ragged knot ranges, the last rank's shorter dz
    slice, the ghost lambda block behind a rank's slice left out."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"],
           os.path.join(ROOT, "tests", "gather_worker.py"), str(S), str(C), str(K)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count(" ok gather ") == world


def test_knot_ranges():
    assert knot_ranges(4096, 8) == [(i * 512, (i + 1) * 512) for i in range(8)]
    assert knot_ranges(10, 3) == [(0, 4), (4, 7), (7, 10)]
    with pytest.raises(ValueError):
        knot_ranges(2, 3)


def test_cluster_setup_fails_on_every_rank_together():
    """ClusterPCG's collective set-up (handle exchange, mapping, fit check): a failure on ONE rank must surface as
    ClusterUnavailable on EVERY rank - nobody left waiting in a collective - so that all ranks take the RCCL fallback."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"], os.path.join(ROOT, "tests", "cluster_host_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count(" ok [") == 2


@pytest.mark.parametrize("hang", [0, 1])
def test_rider_child_jobs_cannot_take_the_bench_line_down(hang):
    """bench.py --gpus N runs the knot-sharded riders as child jobs of the ranks (dist_bench.rider_in_child).  A child that
    dies at once (here: no GPU) or never returns (GATO_RIDER_TEST_HANG) must cost at most the deadline, leave an error
    object on rank 0 and no rank waiting in a collective."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), OMP_NUM_THREADS="1",
               RIDER_DEADLINE="8" if hang else "120", GATO_RIDER_TEST_HANG=str(hang))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"], os.path.join(ROOT, "tests", "rider_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("RIDER ")]
    assert len(line) == 1
    out = json.loads(line[0][6:])
    assert "error" in out["res"] and ("deadline" in out["res"]["error"])
    if hang:
        assert "exit deadline" in out["res"]["error"] and out["seconds"] < 60
