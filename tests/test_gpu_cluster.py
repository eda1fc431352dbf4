"""GPU suite: the in-kernel cross-GPU hand-off of the persistent PCG (gato_cluster_*, pcg_resident_kernel<..., MR>)
against the oracle.  A 1-GPU box gives every rank the same GPU: ranks are either solvers on separate streams of ONE
process (mirrors shared as plain pointers) or separate PROCESSES (mirrors shared through hipIpc handles, exactly as
on a multi-GPU node); in both the launches of all ranks run concurrently and wait for each other on the device."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
from gato_python_amd import _lib, synth                             # noqa: E402
from gato_python_amd.dist import knot_ranges, lockstep_streams, run_cluster_lockstep   # noqa: E402
from oracle import c_oracle as co                                   # noqa: E402
from oracle import gato_oracle as o                                 # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def f32_judged(what, gpu, oracle32, truth64):
    """fp32 results are measured, not given a flat tolerance: the GPU's error against the fp64 iteration on the same fp32
    matrices beside the fp32 oracle's own error (tests/f32_parity.py: same bar and the same log as the parity suite)."""
    from f32_parity import check_f32
    check_f32(what, gpu, oracle32, truth64)


def converged_f64(Sb, Pb, gam, S, K):
    return co.pcg(Sb.astype(np.float64), Pb.astype(np.float64), gam.astype(np.float64), S, K, 1e-14, 600)[0]


def oracle_blocks(S, C, K, dt, seed=13):
    s = synth.make_system(S, C, K, seed=seed)
    Gd, Cd = co.convert(*s.csr_args()[:6], S, C, K, s.rho, dt)
    Sb, Pb, gam, _ = co.form_schur(Gd, Cd, s.g, s.c, S, C, K)
    return Sb, co.form_ss(Sb, Pb, S, K), gam


# (S, C, K, ranks, dtype): one workgroup per rank (level 2 only), many per rank, ragged ranges, 8 ranks, f32,
# BASELINE configs[3] and configs[4] at full size in f64 (equal iteration count with the oracle)
@pytest.mark.parametrize("S,C,K,R,dt", [(14, 7, 50, 2, np.float64), (14, 7, 53, 3, np.float64), (2, 1, 9, 4, np.float64),
                                        (14, 7, 300, 2, np.float32), (32, 16, 100, 3, np.float64),
                                        (14, 7, 4096, 2, np.float64), (32, 16, 1024, 2, np.float64),
                                        (14, 7, 4096, 8, np.float32), (14, 7, 4096, 8, np.float64),
                                        (32, 16, 1024, 8, np.float64), (32, 16, 1024, 8, np.float32)])   # configs[4] names fp32
@pytest.mark.parametrize("flat", [1, 0])
def test_cluster_ranks_in_one_process(S, C, K, R, dt, flat):
    """flat = 1: the one-level exchange (every workgroup's partial straight into every mirror; taken whenever ranks x
    workgroups <= 256); flat = 0: the two-level exchange (what shards beyond 256 workgroups use), forced."""
    _cluster_case(S, C, K, R, dt, flat)


@pytest.mark.parametrize("S,C,K,R,dt,dpp", [(14, 7, 300, 2, np.float32, 1), (14, 7, 4096, 4, np.float32, 1), (14, 7, 4096, 2, np.float64, 0),
                                            (32, 16, 100, 3, np.float32, 0), (12, 6, 200, 3, np.float64, 0), (12, 6, 200, 3, np.float32, 1)])
@pytest.mark.parametrize("flat", [1, 0])
def test_cluster_ranks_in_either_row_layout(S, C, K, R, dt, dpp, flat):
    """The cluster launches exist in both row layouts (LDS operand windows / DPP rows, solver option dpp_rows): here the one the
    default does NOT take for the shape, forced."""
    _cluster_case(S, C, K, R, dt, flat, dpp)


# the single-reduction recurrence in a cluster - ONE cross-GPU exchange per iteration (pcg_cg1_kernel<..., MR>), both
# exchange forms, 2 / 4 / 8 ranks, configs[3] and configs[4] at full size; against its own restatement (oracle.pcg_single_reduction)
# and the reference recurrence's solution
@pytest.mark.parametrize("S,C,K,R,dt", [(14, 7, 50, 2, np.float64), (14, 7, 53, 3, np.float64), (2, 1, 9, 4, np.float64),
                                        (14, 7, 300, 2, np.float32), (32, 16, 100, 3, np.float64),
                                        (14, 7, 4096, 2, np.float64), (14, 7, 4096, 4, np.float32), (32, 16, 1024, 2, np.float32),
                                        (14, 7, 4096, 8, np.float32), (14, 7, 4096, 8, np.float64), (32, 16, 1024, 8, np.float32)])
@pytest.mark.parametrize("flat", [1, 0])
def test_cluster_single_reduction_variant(S, C, K, R, dt, flat):
    _cluster_case(S, C, K, R, dt, flat, variant=1)


def _cluster_case(S, C, K, R, dt, flat, dpp=None, variant=0):
    from gato_python_amd.solver import Solver
    Sb, Pb, gam = oracle_blocks(S, C, K, dt)
    f64 = dt == np.float64
    tol, mi = (1e-9, 150) if f64 else (1e-4, 60)
    lam_o, it_o = co.pcg(Sb, Pb, gam, S, K, tol, mi)
    sols = [Solver(S, C, K, dt) for _ in range(R)]
    for x in sols:
        x.set_option("cluster_flat", flat)
        x.set_option("pcg_variant", variant)
        if R > 4:
            x.set_option("max_workgroups", 256 // R)      # the ranks share ONE GPU here
        if dpp is not None:
            x.set_option("dpp_rows", dpp)
    dS, dP, dg = sols[0].to_device(Sb), sols[0].to_device(Pb), sols[0].to_device(gam)
    lam, its = run_cluster_lockstep(sols, dS, dP, dg, tol, mi)
    for x in sols:
        x.check_status()
        assert x.get_option("last_variant") == variant
        if not variant:
            want = dpp if dpp is not None else (1 if (f64 or S > 16) and S in (12, 14, 16, 32) else 0)      # cluster launches: fp64, and S = 32
            assert x.get_option("last_dpp") == want, (x.get_option("last_dpp"), want)
    fits_flat = sum(x.get_option("last_groups") for x in sols) <= 256
    assert run_cluster_lockstep.last_flat == (1 if flat and fits_flat else 0)
    got = lam.cpu().numpy()
    if variant:
        lam_cg, it_cg = o.pcg_single_reduction(Sb, Pb, gam, S, K, tol, mi)
        assert len(set(its)) == 1 and abs(its[0] - it_cg) <= (0 if f64 else 2) and abs(its[0] - it_o) <= 2, (its, it_cg, it_o)
        if f64:
            assert np.abs(got - lam_cg).max() / np.abs(lam_cg).max() < 1e-8 and np.abs(got - lam_o).max() / np.abs(lam_o).max() < 1e-6
        else:
            conv = converged_f64(Sb, Pb, gam, S, K)
            f32_judged(f"cluster {R} ranks single-reduction vs its restatement {S}/{C}/{K} flat={flat}", got, lam_cg, conv)
            f32_judged(f"cluster {R} ranks single-reduction vs the reference recurrence {S}/{C}/{K} flat={flat}", got, lam_o, conv)
    else:
        assert len(set(its)) == 1 and abs(its[0] - it_o) <= (0 if f64 else 2), (its, it_o)
        if f64:
            err = np.abs(got - lam_o).max() / np.abs(lam_o).max()
            assert err < 1e-9, err
        else:
            f32_judged(f"cluster {R} ranks {S}/{C}/{K} flat={flat}", got, lam_o, converged_f64(Sb, Pb, gam, S, K))
    for x in sols:
        x.close()


def test_cluster_back_to_back_solves_and_fixed_iterations():
    """Several solves through one connected cluster (lock-step epoch ranges), exit_tol = 0 (exactly max_iters iterations)."""
    from gato_python_amd.dist import ClusterPCG
    from gato_python_amd.solver import Solver
    S, C, K, R, dt = 14, 7, 600, 3, np.float64
    Sb, Pb, gam = oracle_blocks(S, C, K, dt)
    lam_o, it_o = co.pcg(Sb, Pb, gam, S, K, 0.0, 25)
    sols = [Solver(S, C, K, dt) for _ in range(R)]
    dS, dP, dg = sols[0].to_device(Sb), sols[0].to_device(Pb), sols[0].to_device(gam)
    cl = [ClusterPCG(s_, r, R, inprocess_peers=True) for r, s_ in enumerate(sols)]
    ClusterPCG.connect_inprocess(cl)
    assert [(c.k0, c.k1) for c in cl] == knot_ranges(K, R)
    streams = lockstep_streams(R)
    torch.cuda.synchronize()
    first = None
    for rep in range(5):
        lam = torch.zeros(S * K, dtype=torch.float64, device="cuda")
        its = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(R)]
        torch.cuda.synchronize()
        for r in range(R):
            cl[r].pcg(dS, dP, dg, 0.0, 25, lam, its[r], stream=streams[r].cuda_stream)
        torch.cuda.synchronize()
        assert [int(i.cpu()[0]) for i in its] == [25] * R
        got = lam.cpu().numpy()
        assert np.abs(got - lam_o).max() / np.abs(lam_o).max() < 1e-9
        first = got if first is None else first
        assert np.array_equal(got, first)                  # fixed summation order: bitwise reproducible
    for s_ in sols:
        s_.check_status()
        s_.close()


@pytest.mark.parametrize("variant,flat", [(0, 1), (0, 0), (1, 1), (1, 0)])
def test_cluster_renews_its_epoch_space(variant, flat):
    """The hand-off epochs of a cluster only grow; ten million solves use the 32 bits up.  The counters of every rank are placed
    two launches before the end (test hook cluster_epoch): the third solve renews the space - mirrors and slots zeroed on every
    rank between two waits - and the solves go on with the same bits; the C entry alone refuses the launch that does not fit."""
    from gato_python_amd.dist import ClusterPCG
    from gato_python_amd.solver import Solver
    S, C, K, R, dt, mi = 14, 7, 300, 3, np.float64, 40
    Sb, Pb, gam = oracle_blocks(S, C, K, dt)
    lam_o = (o.pcg_single_reduction(Sb, Pb, gam, S, K, 0.0, mi) if variant else co.pcg(Sb, Pb, gam, S, K, 0.0, mi))[0]
    sols = [Solver(S, C, K, dt) for _ in range(R)]
    for x in sols:
        x.set_option("pcg_variant", variant)
        x.set_option("cluster_flat", flat)
    dS, dP, dg = sols[0].to_device(Sb), sols[0].to_device(Pb), sols[0].to_device(gam)
    cl = [ClusterPCG(s_, r, R, inprocess_peers=True) for r, s_ in enumerate(sols)]
    ClusterPCG.connect_inprocess(cl)
    need = 2 * mi + 8
    top = 0xFFFFFFFF - need - 8
    for x in sols:
        x.set_option("cluster_epoch", (top - need - 5) - (1 << 32))     # as a C int: two launches left
        x.set_option("pcg_epoch", (top - need - 5) - (1 << 32))
    assert [c.launches_left(mi) for c in cl] == [2] * R
    streams = lockstep_streams(R)
    first = None
    for rep in range(5):
        lam = torch.zeros(S * K, dtype=torch.float64, device="cuda")
        its = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(R)]
        torch.cuda.synchronize()
        for r in range(R):
            cl[r].pcg(dS, dP, dg, 0.0, mi, lam, its[r], stream=streams[r].cuda_stream)
        torch.cuda.synchronize()
        assert [int(i.cpu()[0]) for i in its] == [mi] * R, rep
        got = lam.cpu().numpy()
        assert np.abs(got - lam_o).max() / np.abs(lam_o).max() < (1e-7 if variant else 1e-9), rep
        first = got if first is None else first
        assert np.array_equal(got, first), rep
        assert [c.rewinds for c in cl] == [0 if rep < 2 else 1] * R
    assert cl[0].launches_left(mi) > 5_000_000
    # a rank's LEVEL-1 counter alone near its end (the solver also served plain multi-workgroup launches): it renews itself in
    # stream order - its slots are written by the rank's own workgroups only - and no rank has to wait for another
    for x in sols:
        x.set_option("pcg_epoch", (top - 5) - (1 << 32))
    for rep in range(3):
        lam = torch.zeros(S * K, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        for r in range(R):
            cl[r].pcg(dS, dP, dg, 0.0, mi, lam, its[r], stream=streams[r].cuda_stream)
        torch.cuda.synchronize()
        assert [int(i.cpu()[0]) for i in its] == [mi] * R and np.array_equal(lam.cpu().numpy(), first), rep
    assert [c.rewinds for c in cl] == [1] * R
    # the C entry by itself: a launch that does not fit is refused
    for x in sols:
        x.set_option("cluster_epoch", top + 1 - (1 << 32))
    assert cl[0].launches_left(mi) == 0
    lam = torch.zeros(S * K, dtype=torch.float64, device="cuda")
    it = torch.zeros(1, dtype=torch.int32, device="cuda")
    p = lambda t: __import__("ctypes").c_void_p(t.data_ptr())
    rc = _lib.lib().gato_cluster_pcg(sols[0]._h, p(dS), p(dP), p(dg), p(lam), 0.0, mi, p(it), None)
    assert rc != 0 and b"epoch space" in _lib.lib().gato_last_error()
    rc = _lib.lib().gato_cluster_pcg(sols[0]._h, p(dS), p(dP), p(dg), p(lam), 0.0, -1, p(it), None)      # -1 is what a time-out reports
    assert rc != 0 and b"max_iters must be >= 0" in _lib.lib().gato_last_error()
    for c in cl:
        c.close()
    for x in sols:
        x.check_status()
        x.close()


@pytest.mark.parametrize("S,C,K,semi", [(14, 7, 30000, 1), (14, 7, 30000, 3), (32, 16, 9000, 3)])
def test_cluster_semi_resident_shards(S, C, K, semi):
    """Shards beyond the register file against the streaming kernels on one GPU: the semi-resident launch per rank (semi = 1)
    and the LDS-DMA ring launch per rank (semi = 3: pcg_dma_kernel<..., MR>, what shards far beyond the caches get)."""
    from gato_python_amd.solver import Solver
    R, dt = 2, np.float32
    sysm = synth.make_system(S, C, K, seed=3)
    one = Solver(S, C, K, dt)
    d = one.upload_system(sysm)
    Gd, Cd = one.convert(*d[:6], sysm.rho)
    Sb, Pb, gam, _ = one.form_schur(Gd, Cd, d[6], d[7])
    one.form_ss(Sb, Pb)
    one.set_option("pcg_mode", _lib.PCG_STREAMING)
    lam_s, it_s = one.pcg(Sb, Pb, gam, 1e-4, 60)
    sols = [Solver(S, C, K, dt) for _ in range(R)]
    for x in sols:
        x.set_option("max_workgroups", 120)               # the ranks share ONE GPU here: 2 x 120 workgroups fit its 256 CUs
        x.set_option("pcg_semi", semi)
    lam, its = run_cluster_lockstep(sols, Sb, Pb, gam, 1e-4, 60)
    assert sols[0].get_option("last_semi") == semi and sols[0].get_option("last_groups") <= 120
    assert len(set(its)) == 1 and abs(its[0] - int(it_s.cpu()[0])) <= 1, (its, it_s)
    # against the C oracle on the same (GPU-assembled) fp32 matrices, measured like every fp32 result
    hS, hP, hg = Sb.cpu().numpy(), Pb.cpu().numpy(), gam.cpu().numpy()
    lam_o, it_o = co.pcg(hS, hP, hg, S, K, 1e-4, 60)
    assert abs(its[0] - it_o) <= 1, (its, it_o)
    truth = converged_f64(hS, hP, hg, S, K)
    f32_judged(f"cluster of {R} semi-resident shards {S}/{C}/{K}", lam.cpu().numpy(), lam_o, truth)
    f32_judged(f"streaming kernels {S}/{C}/{K} (the shards' comparison run)", lam_s.cpu().numpy(), lam_o, truth)
    for x in sols + [one]:
        x.close()


def test_cluster_neighbouring_ranks_on_different_launch_variants():
    """Shards differ by a knot when K is not a multiple of the ranks, and a launch-capacity boundary can fall between them: here
    rank 0 (6481 knots) takes the semi-resident launch, rank 1 (6480 = 120 workgroups x 54 knots of 768 threads) the resident one.  Both
    are pcg_resident_kernel<..., MR> with the same hand-off protocol; the solve must not care."""
    from gato_python_amd.solver import Solver
    S, C, K, R, dt = 14, 7, 12961, 2, np.float32
    Sb, Pb, gam = oracle_blocks(S, C, K, dt, seed=5)
    lam_o, it_o = co.pcg(Sb, Pb, gam, S, K, 1e-4, 60)
    sols = [Solver(S, C, K, dt) for _ in range(R)]
    for x in sols:
        x.set_option("max_workgroups", 120)               # two ranks share this GPU
    dS, dP, dg = sols[0].to_device(Sb), sols[0].to_device(Pb), sols[0].to_device(gam)
    lam, its = run_cluster_lockstep(sols, dS, dP, dg, 1e-4, 60)
    for x in sols:
        x.check_status()
    semi = [x.get_option("last_semi") for x in sols]
    assert knot_ranges(K, R) == [(0, 6481), (6481, 12961)] and semi[0] != 0 and semi[1] == 0, (knot_ranges(K, R), semi)
    assert len(set(its)) == 1 and abs(its[0] - it_o) <= 1, (its, it_o)
    f32_judged(f"cluster of a semi-resident and a resident shard {S}/{C}/{K}", lam.cpu().numpy(), lam_o, converged_f64(Sb, Pb, gam, S, K))
    for x in sols:
        x.close()


def test_cluster_launch_refuses_stream_capture():
    """a cluster launch draws its hand-off epochs per launch - captured into a graph it would be replayed with stale
    ones - so gato_cluster_pcg refuses a capturing stream like the multi-workgroup launches of one GPU do."""
    from gato_python_amd.dist import ClusterPCG
    from gato_python_amd.solver import Solver
    S, C, K, R, dt = 14, 7, 64, 2, np.float64
    Sb, Pb, gam = oracle_blocks(S, C, K, dt)
    sols = [Solver(S, C, K, dt) for _ in range(R)]
    cl = [ClusterPCG(x, r, R, inprocess_peers=True) for r, x in enumerate(sols)]
    ClusterPCG.connect_inprocess(cl)
    dS, dP, dg = sols[0].to_device(Sb), sols[0].to_device(Pb), sols[0].to_device(gam)
    lam = torch.zeros(S * K, dtype=torch.float64, device="cuda")
    it = torch.zeros(1, dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        g.capture_begin()
        try:
            with pytest.raises(_lib.GatoError, match="captured"):
                cl[0].pcg(dS, dP, dg, 1e-9, 50, lam, it, stream=st.cuda_stream)
        finally:
            g.capture_end()
    torch.cuda.synchronize()
    for c in cl:
        c.close()
    for x in sols:
        x.close()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("S,C,K,dt,world,variant", [(14, 7, 4096, "f64", 2, 0), (32, 16, 1024, "f64", 2, 0), (14, 7, 4096, "f32", 2, 0),
                                                    (14, 7, 4096, "f64", 2, 1), (32, 16, 1024, "f32", 2, 1)])
def test_cluster_one_process_per_rank_ipc(S, C, K, dt, world, variant):
    """TWO PROCESSES, mirrors shared by hipIpc handles: BASELINE configs[3] / configs[4] shapes, equal `iters` in f64 - the PCG
    launches on oracle-assembled matrices, then whole solves through linsys_solve_cluster (sharded assembly, launch and dz in
    one library call per rank; lambda_{k1} crosses the ranks inside the launch).  variant 1: the single-reduction recurrence."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), OMP_NUM_THREADS="4",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    tol, mi = ("1e-9", "150") if dt == "f64" else ("1e-4", "60")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"],
           os.path.join(ROOT, "tests", "cluster_worker.py"), str(S), str(C), str(K), dt, tol, mi, "3", str(variant)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count(" ok iters=") == world, r.stdout[-2000:]


@pytest.mark.parametrize("S,C,K,R,dt", [(14, 7, 50, 2, np.float64), (14, 7, 53, 3, np.float64), (32, 16, 40, 3, np.float32), (2, 1, 9, 3, np.float64)])
@pytest.mark.parametrize("variant", [0, 1])
def test_sharded_whole_solve_in_one_call_per_rank(S, C, K, R, dt, variant):
    """gato_cluster_linsys: every rank's stage kernels on the knots its shard reads (rows k0..k1-1 of S / Pinv and gamma on
    k0-1..k1 - one knot more on either side for the single-reduction recurrence - bit-identical to the full assembly), its
    persistent launch, and dz on its knots with lambda_{k1} delivered INSIDE the launch (cluster_lambda_ghost): no lambda
    all-reduce, nothing host-side between the three.  The assembled lambda / dz equal the one-GPU solve."""
    from gato_python_amd.dist import ClusterPCG
    from gato_python_amd.solver import Solver
    s = synth.make_system(S, C, K, seed=21)
    f64 = dt == np.float64
    tol, mi = (1e-9, 150) if f64 else (1e-4, 60)
    one = Solver(S, C, K, dt)
    one.set_option("pcg_variant", variant)
    d = one.upload_system(s)
    Gd, Cd = one.convert(*d[:6], s.rho)
    Sb, Pb, gam, Gi = one.form_schur(Gd, Cd, d[6], d[7])
    one.form_ss(Sb, Pb)
    lam1, dz1 = one.new(S * K), one.new(one.N)
    one.linsys(*d, tol, mi, s.rho, lam1, dz1)
    one.check_status()
    sols = [Solver(S, C, K, dt) for _ in range(R)]
    for x in sols:
        x.set_option("pcg_variant", variant)
    cl = [ClusterPCG(x, r, R, inprocess_peers=True) for r, x in enumerate(sols)]
    ClusterPCG.connect_inprocess(cl)
    streams = lockstep_streams(R)
    # every rank its OWN full-length output buffers (as one process per GPU has), NaN where it must not rely on anything
    lams = [torch.full((S * K,), float("nan"), dtype=one.dtype, device="cuda") for _ in range(R)]
    dzs = [torch.full((one.N,), float("nan"), dtype=one.dtype, device="cuda") for _ in range(R)]
    its = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(R)]
    torch.cuda.synchronize()
    for rep in range(2):                                   # twice through the connected cluster: lock-step epochs and tags
        for r in range(R):
            cl[r].linsys(d, tol, mi, s.rho, lams[r], dzs[r], its[r], stream=streams[r].cuda_stream)
        torch.cuda.synchronize()
    SS, n = S * S, S + C
    lam, dz = torch.zeros(S * K, dtype=one.dtype, device="cuda"), torch.zeros(one.N, dtype=one.dtype, device="cuda")
    hS, hP, hg = Sb.cpu().numpy(), Pb.cpu().numpy(), gam.cpu().numpy()
    for r in range(R):
        sols[r].check_status()
        assert sols[r].get_option("last_variant") == variant
        k0, k1 = cl[r].k0, cl[r].k1
        bS, bP, bg = sols[r].read_buffer("S"), sols[r].read_buffer("Pinv"), sols[r].read_buffer("gamma")
        h = variant
        assert np.array_equal(bS[k0 * 3 * SS:k1 * 3 * SS], hS[k0 * 3 * SS:k1 * 3 * SS]), r
        p0, p1 = max(k0 - h, 0), min(k1 + h, K)
        assert np.array_equal(bP[p0 * 3 * SS:p1 * 3 * SS], hP[p0 * 3 * SS:p1 * 3 * SS]), r
        g0, g1 = max(k0 - 1 - h, 0), min(k1 + 1 + h, K)
        assert np.array_equal(bg[g0 * S:g1 * S], hg[g0 * S:g1 * S]), r
        lam[k0 * S:k1 * S] = lams[r][k0 * S:k1 * S]
        hi = min(k1 * n, one.N)
        dz[k0 * n:hi] = dzs[r][k0 * n:hi]
        if r < R - 1:                                      # the neighbour's first block arrived inside the launch
            assert torch.equal(lams[r][k1 * S:(k1 + 1) * S], lams[r + 1][k1 * S:(k1 + 1) * S])
    assert len({int(i.cpu()[0]) for i in its}) == 1
    den = float(lam1.abs().max())
    if f64:
        bar = 1e-8 if variant else 1e-9                    # (the one-GPU launch of variant 1 groups its dots differently)
        assert float((lam - lam1).abs().max()) / den < bar
        assert float((dz - dz1).abs().max()) / float(dz1.abs().max()) < bar
    else:           # fp32: the sharded solve against the oracle's whole solve, measured (tests/f32_parity.py)
        lam_o, dz_o, _ = co.linsys_solve(*s.csr_args(), S, C, K, tol, mi, s.rho, dtype=dt)
        s64 = s.astype(np.float32).astype(np.float64)
        lam_t, dz_t, _ = co.linsys_solve(*s64.csr_args(), S, C, K, 1e-14, 600, float(np.float32(s.rho)), dtype=np.float64)
        f32_judged(f"sharded whole solve variant {variant} {S}/{C}/{K} lambda", lam.cpu().numpy(), lam_o, lam_t)
        f32_judged(f"sharded whole solve variant {variant} {S}/{C}/{K} dz", dz.cpu().numpy(), dz_o, dz_t)
    for c in cl:
        c.close()
    for x in sols + [one]:
        x.close()


@pytest.mark.parametrize("flat", [1, 0])
def test_cluster_true_warm_start_and_eta_history(flat):
    """Options of the resident kernel inside a cluster launch: lambda0 given (r0 = gamma - S lambda0: the ghost blocks of
    r0 cross the ranks through the same hand-off) and the eta history, against one launch on the whole GPU."""
    from gato_python_amd.dist import ClusterPCG
    from gato_python_amd.solver import Solver
    S, C, K, R, dt = 14, 7, 300, 3, np.float64
    Sb, Pb, gam = oracle_blocks(S, C, K, dt)
    rng = np.random.default_rng(3)
    lam0 = 0.1 * rng.standard_normal(S * K)
    one = Solver(S, C, K, dt)
    one.set_option("true_warm_start", 1)
    one.set_option("record_eta", 1)
    dS, dP, dg = one.to_device(Sb), one.to_device(Pb), one.to_device(gam)
    l1 = one.to_device(lam0)
    l1, it1 = one.pcg(dS, dP, dg, 1e-9, 150, lam=l1)
    n1 = int(it1.cpu()[0])
    h1 = one.eta_history(n1 + 1)
    sols = [Solver(S, C, K, dt) for _ in range(R)]
    for x in sols:
        x.set_option("true_warm_start", 1)
        x.set_option("record_eta", 1)
        x.set_option("cluster_flat", flat)
    cl = [ClusterPCG(x, r, R, inprocess_peers=True) for r, x in enumerate(sols)]
    ClusterPCG.connect_inprocess(cl)
    streams = lockstep_streams(R)
    lam = one.to_device(lam0)                              # every rank reads its lambda0 (and its ghosts) from the full array
    its = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(R)]
    torch.cuda.synchronize()
    for r in range(R):
        cl[r].pcg(dS, dP, dg, 1e-9, 150, lam, its[r], stream=streams[r].cuda_stream)
    torch.cuda.synchronize()
    assert [int(i.cpu()[0]) for i in its] == [n1] * R
    assert float((lam - l1).abs().max()) / float(l1.abs().max()) < 1e-9
    for x in sols:
        x.check_status()
        assert np.allclose(x.eta_history(n1 + 1), h1, rtol=1e-9, atol=0)
    for x in sols + [one]:
        x.close()


def test_clusters_created_and_destroyed_in_one_process():
    """Randomised cluster geometries one after the other in ONE process (tools/cluster_fuzz.py, fixed seed): solvers and
    clusters are destroyed and re-created between the cases, so each case runs on memory the previous ones gave back.  Cases
    27-31 of seed 1 are the sequence that read stale lines in round 5 (an arena allocated over a freed uncached mirror, see
    mirror_take in gato_capi.hip); every PCG solve and every whole sharded solve is held against the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import cluster_fuzz
    rng = np.random.default_rng(1)
    only = set(range(20, 36))
    bad = []
    for i in range(36):
        msg, ok = cluster_fuzz.case(rng, i, only)
        if not ok:
            bad.append(msg)
    assert not bad, "\n".join(bad)


def test_cluster_states_connected_and_closed_across_processes():
    """THREE PROCESSES, a fixed-seed slice of tools/cluster_fuzz_ipc.py: twenty random geometries one after the other, each with a
    fresh cluster state (mirrors re-exported and re-mapped through hipIpc handles), two whole solves per state through
    linsys_solve_cluster / linsys_solve_auto - the second with another system and sometimes the other recurrence - then
    close_state; every rank holds the gathered lambda / dz against the oracle's whole solve."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), OMP_NUM_THREADS="4",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
           "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"],
           os.path.join(ROOT, "tools", "cluster_fuzz_ipc.py"), "20", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "FUZZ ok 0" in r.stdout and r.stdout.count("\nok   case") + r.stdout.startswith("ok   case") == 20, r.stdout[-3000:]


def test_cluster_survives_a_late_rank():
    """TWO PROCESSES through linsys_solve_auto: a rank that launches later than timeout_ms makes the hand-off time out on every
    rank alike; that solve is repeated over the RCCL schedule (complete, equal to the oracle), the next solves run on the cluster
    again (its epochs only grow: the late launch leaves nothing behind that a later one could take for its own), and only
    MAX_CONSECUTIVE_TIMEOUTS in a row drop the cluster for good."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), OMP_NUM_THREADS="4",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"], os.path.join(ROOT, "tests", "cluster_retry_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count(" ok") == 2, r.stdout[-2000:]


@pytest.mark.parametrize("S,C", [(14, 7), (2, 1), (32, 16)])
@pytest.mark.parametrize("R", [2, 3, 8])
def test_sharded_whole_solve_with_one_knot_per_rank(S, C, R):
    """The smallest shards: K = R (one knot per rank: its only block row is first AND last of the launch), R + 1, 2 R - 1 (ragged),
    both recurrences asked for (the single-reduction one needs two knots per workgroup on every rank: all ranks take the default
    together).  Whole sharded solves against the oracle's; systems this small run CG to its finite termination, where iterates
    depend on the summation order at cond(S) * eps, so the bar is loose - a wrong halo or ghost block is an error of order one."""
    from gato_python_amd.dist import ClusterPCG
    from gato_python_amd.solver import Solver
    for K in (R, R + 1, 2 * R - 1):
        for variant in (0, 1):
            s = synth.make_system(S, C, K, seed=K)
            lam_w, dz_w, it_w = co.linsys_solve(*s.csr_args(), S, C, K, 1e-10, 100, s.rho, dtype=np.float64)
            sols = [Solver(S, C, K, np.float64) for _ in range(R)]
            for x in sols:
                x.set_option("pcg_variant", variant)
            cl = [ClusterPCG(x, r, R, inprocess_peers=True) for r, x in enumerate(sols)]
            ClusterPCG.connect_inprocess(cl)
            streams = lockstep_streams(R)
            d = sols[0].upload_system(s)
            lams = [torch.full((S * K,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(R)]
            dzs = [torch.full((sols[0].N,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(R)]
            its = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(R)]
            torch.cuda.synchronize()
            for r in range(R):
                cl[r].linsys(d, 1e-10, 100, s.rho, lams[r], dzs[r], its[r], stream=streams[r].cuda_stream)
            torch.cuda.synchronize()
            n = S + C
            la, da = np.empty(S * K), np.empty(sols[0].N)
            for r in range(R):
                k0, k1 = cl[r].k0, cl[r].k1
                la[k0 * S:k1 * S] = lams[r][k0 * S:k1 * S].cpu().numpy()
                hi = min(k1 * n, sols[0].N)
                da[k0 * n:hi] = dzs[r][k0 * n:hi].cpu().numpy()
            itg = [int(t.cpu()[0]) for t in its]
            assert len(set(itg)) == 1 and abs(itg[0] - it_w) <= 2, (K, variant, itg, it_w)
            assert np.abs(la - lam_w).max() / np.abs(lam_w).max() < 1e-5, (K, variant)
            assert np.abs(da - dz_w).max() / max(np.abs(dz_w).max(), 1e-300) < 1e-4, (K, variant)
            for c_ in cl:
                c_.close()
            for x in sols:
                x.check_status()
                x.close()
