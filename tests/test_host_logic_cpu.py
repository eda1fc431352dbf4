"""CPU suite: host-side logic added in round 2 that needs no GPU - the cluster's knot partition through the C ABI, the
hand-off layout mirror, bench.py's traffic provenance check and its secondary CPU baseline."""
import ctypes as ct
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gato_python_amd import _lib, csrc_layout, synth          # noqa: E402
from gato_python_amd.dist import knot_ranges                   # noqa: E402


@pytest.mark.parametrize("K,R", [(4096, 8), (10, 3), (53, 3), (9, 4), (8, 8), (262144, 8), (1, 1)])
def test_cluster_knot_range_matches_the_python_partition(K, R):
    L = _lib.lib()
    got = []
    for r in range(R):
        a, b = ct.c_int(), ct.c_int()
        assert L.gato_cluster_knot_range(K, r, R, ct.byref(a), ct.byref(b)) == 0
        got.append((a.value, b.value))
    assert got == knot_ranges(K, R)
    assert got[0][0] == 0 and got[-1][1] == K and all(x[1] == y[0] for x, y in zip(got, got[1:]))


def test_cluster_knot_range_rejects_more_ranks_than_knots():
    a, b = ct.c_int(), ct.c_int()
    assert _lib.lib().gato_cluster_knot_range(3, 0, 4, ct.byref(a), ct.byref(b)) == -1
    assert b"cannot shard" in _lib.lib().gato_last_error()


def test_handoff_slots_are_whole_cache_lines():
    """One writing workgroup per 128-byte line (DESIGN.md 3.1 dead end 2): a slot is a whole number of 16-granule lines,
    the partial has line 0 to itself, both halo blocks fit behind it; the cluster mirror gives every rank total a line."""
    for S in (2, 4, 6, 12, 14, 32):
        for esz in (4, 8):
            g = csrc_layout.slot_granules(S, esz)
            assert g % 16 == 0 and g >= 16 + 2 * S * (esz // 4)
            x = csrc_layout.xslot_granules(S, esz)
            assert x % 16 == 0 and x >= 16 * csrc_layout.MAX_RANKS + 2 * S * (esz // 4)


def test_bench_refuses_stale_traffic_entries():
    import bench
    src = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    e = src["iiwa_14_7_k50_f64"]
    ok = dict(pcg_groups=1, pcg_threads=e["grid_threads"], dtype="f64", S=14)
    t, why = bench.committed_traffic("iiwa_14_7_k50_f64", ok)
    assert t == e["hbm_bytes_per_launch"] and why["git_blob"] and why["kernel"] == e["kernel"]
    t, why = bench.committed_traffic("iiwa_14_7_k50_f64", dict(ok, pcg_threads=e["grid_threads"] + 64))   # another geometry ran
    assert t is None and "stale" in why
    t, why = bench.committed_traffic("iiwa_14_7_k50_f64", dict(ok, dtype="f32"))              # another kernel family ran
    assert t is None and "stale" in why
    assert bench.committed_traffic("no_such_workload", ok)[0] is None
    e5 = src["iiwa_14_7_k512_f32"]                                                           # one-XCD launch: 8x oversubscribed grid
    assert bench.committed_traffic("iiwa_14_7_k512_f32", dict(pcg_groups=e5["grid_threads"] // 8 // 512, pcg_threads=512, dtype="f32", S=14))[0]


def test_scipy_secondary_baseline_runs_the_same_iteration():
    """bench.py's secondary CPU baseline: scipy's cg on -S with -Pinv reproduces the oracle's iterates."""
    import bench
    import scipy.sparse.linalg as spl
    from oracle import gato_oracle as o
    S, C, K = 14, 7, 12
    s = synth.make_system(S, C, K, seed=2)
    saved = bench.MAX_ITERS
    bench.MAX_ITERS = 15
    try:
        r = bench.scipy_cg_baseline(s, np.float64)
    finally:
        bench.MAX_ITERS = saved
    assert r["value"] > 0 and "scipy" in r["kind"]
    out = o.linsys_solve(*s.csr_args(), S, C, K, 0.0, 15, s.rho, dtype=np.float64, return_all=True)
    # same Krylov iterates: 15 steps of scipy's PCG on the negated system give the oracle's lambda
    import scipy.sparse as sp

    def bd(m):
        L, M, R = o.unpack_bd(m, S, K)
        blocks = [[None] * K for _ in range(K)]
        for k in range(K):
            blocks[k][k] = -M[k]
            if k:
                blocks[k][k - 1] = -L[k]
            if k < K - 1:
                blocks[k][k + 1] = -R[k]
        return sp.bmat(blocks, format="csr")
    x, _ = spl.cg(bd(out["S"]), -out["gamma"], rtol=0.0, atol=0.0, maxiter=15, M=bd(out["Pinv"]))
    assert np.abs(x - out["lam"]).max() / np.abs(out["lam"]).max() < 1e-8


def test_transport_search_over_mirror_memory_kinds():
    """dist_bench.first_working_kind: the in-kernel transport is tried with the mirrors in each memory kind in turn; the
    first kind in which every rank could map the mirrors and finish a solve is taken, a rejection for size ends the
    search, and what was tried is kept for the bench line."""
    from gato_python_amd.dist_bench import first_working_kind
    calls = []

    def attempt(kind):
        calls.append(kind)
        return ("cluster", "") if kind == "finegrained" else (None, "mirrors unavailable: connect: hipIpcOpenMemHandle")

    c, why = first_working_kind(["uncached", "finegrained", "plain"], attempt)
    assert c == "cluster" and calls == ["uncached", "finegrained"] and why.startswith("uncached: mirrors unavailable")
    calls.clear()
    c, why = first_working_kind(["uncached", "finegrained", "plain"], lambda k: (calls.append(k), (None, "the first in-kernel exchange timed out"))[1])
    assert c is None and calls == ["uncached", "finegrained", "plain"] and why.count("timed out") == 3
    calls.clear()
    c, why = first_working_kind(["uncached", "finegrained"], lambda k: (calls.append(k), (None, "mirrors unavailable: 9 knots over 2 ranks do not fit one launch"))[1])
    assert c is None and calls == ["uncached"]
