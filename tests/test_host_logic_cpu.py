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


def _canned_result():
    lf = {"us_per_iteration": 1.42, "products_us": 0.99, "reductions_and_handoffs_us": 0.43, "diagnostic_build_full_us": 1.97,
          "loop_skeleton_us": 0.43, "production_us_per_iteration": 2.044, "frac_of_floor": float("nan"),
          "handoff_floor_us": 0.59, "us_per_iteration_with_handoff_floor": 2.17}
    return dict(workload="iiwa_14_7_k50_f64", S=14, C=7, K=50, dtype="f64", iters_per_s=453210.98765432, ms_per_step=0.22064,
                timed_steps=1140, latency_floor=lf, pcg_launch_ms=0.2044, pcg_launch_ms_min=0.2031,
                pcg_iters_per_s=489236.79, pcg_us_per_iter=2.044, pcg_mode="resident (one workgroup, mixed 2/1 rows per lane)",
                pcg_groups=1, pcg_threads=512, algorithmic_bytes_per_launch=53692800, achieved_gbs=262.68)


def test_bench_headline_is_compact_strict_json():
    """the driver parses the LAST stdout line of bench.py and keeps only a tail of stdout, so that line
    must be one strict-JSON object under 4 KB with metric / value / roofline / cpu_baseline, whatever the sweep holds."""
    import bench
    res = _canned_result()
    traffic, src = bench.committed_traffic("iiwa_14_7_k50_f64", res)
    cpu = {"value": 35123.4, "unit": "PCG iterations/s", "cores": 1, "kind": "port", "sample": "x" * 150,
           "tried": [{"cores": 1, "value": 35123.4}, {"cores": 16, "value": 21000.0}],
           "secondary": {"value": 36000.0, "unit": "PCG iterations/s", "cores": 1, "kind": "scipy", "sample": "y" * 60}}
    out = bench.headline(res, "iiwa_14_7_k50_f64", 20, 5, traffic, src, cpu, "gpurun_out/bench_sweep.json")
    # the full r02 sweep had 15 entries: give the line as many summaries
    out["sweep"] = {f"workload_number_{i}_with_a_long_name_f32": bench.sweep_summary(dict(res, roofline_frac=0.89 if i % 2 else None))
                    for i in range(16)}
    line = bench.dumps_strict(out, bench.LINE_LIMIT)
    assert len(line) < 4096 and "\n" not in line and "NaN" not in line and "Infinity" not in line
    back = json.loads(line, parse_constant=lambda c: pytest.fail(f"non-strict constant {c}"))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in back
    assert back["config"]["workload"] == "iiwa_14_7_k50_f64" and back["vs_baseline"] is None and back["dtype"] == "f64"
    ro = back["roofline"]
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and ro["peak"] == 8000.0 and abs(ro["frac"] - ro["achieved"] / 8000.0) < 1e-6
    assert abs(ro["traffic"] - traffic) < 1.0 and ro["traffic_over_algorithmic"] < 0.05 and ro["limiter"].startswith("latency")
    assert ro["latency_floor"]["frac_of_floor"] is None           # NaN -> null
    assert {"value", "unit", "cores", "kind", "sample"} <= set(back["cpu_baseline"])
    # a launch that streams its matrices every iteration is reported as HBM-limited
    hb = dict(res, algorithmic_bytes_per_launch=1000, achieved_gbs=7000.0)
    assert bench.roofline_object(hb, 900, "x")["limiter"] == "hbm"
    with pytest.raises(ValueError):
        bench.dumps_strict({"x": "y" * 5000}, bench.LINE_LIMIT)


def test_bench_multi_gpu_line_is_led_by_the_sharded_system():
    """the last line of `bench.py --gpus N` leads with BASELINE configs[3] knot-sharded over the ranks
    ("scaling": "strong", the same system on one GPU beside it as `one_gpu_value` / `sharded_speedup`); the N replicas of the
    N = 1 workload ride under `replicas`, the other sharded shapes under config.sharded; under 4 KB, strict JSON.  If the
    lead rider did not deliver, the replicas line stays the headline and says why."""
    import copy
    import bench
    from gato_python_amd import dist_bench as db
    assert "sharded_s32_k1024_f32" in db.DEFAULT_RIDERS and db.LEAD == "sharded_k4096_f32" and db.LEAD in db.DEFAULT_RIDERS
    rider = {"value": 181000.0, "unit": "iterations/s", "ms_per_step": 0.55, "scaling": "strong", "dtype": "f32",
             "config": {"workload": "sharded_k4096_f32", "transport": "xgmi", "transport_fallback_reason": "", "mirror_memory": "uncached",
                        "knots_per_gpu": 2048, "pcg_workgroups_per_gpu": 57, "parallelism": "p" * 300},
             "pcg_us_per_iter": 5.3, "out_of_loop_ms": 0.02, "roofline": {"bound": "hbm", "achieved": 4000.0, "frac": 0.25},
             "single_reduction": {"ran_variant": 1, "iters_per_s": 250000.0, "pcg_us_per_iter": 3.9},
             "parity": {"lam_rel_err_vs_single_gpu": 3e-7, "dz_abs_err_vs_single_gpu": 1e-6, "iters": 100,
                        "same_system_on_one_gpu_iters_per_s": 231000.0, "same_system_on_one_gpu_pcg_us_per_iter": 3.86,
                        "same_system_on_one_gpu_kernel": "resident"}}
    rep = {"metric": "PCG iterations/s", "value": 9e5, "unit": "iterations/s", "n_gpus": 2, "steps": 20, "warmup": 5,
           "ms_per_step": 0.22, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "iiwa_14_7_k50_f64", "parallelism": "replicas only: " + "r" * 150},
           "roofline": {"bound": "hbm", "achieved": 262.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.0328, "traffic": None}}
    out = db.attach_riders(copy.deepcopy(rep), {"sharded_k4096_f32": rider, "sharded_s32_k1024_f32": dict(rider),
                                                "sharded_k262144_f32": {"error": "rider child job did not deliver (exit 1, deadline 240 s)", "log_tail": "z" * 400}})
    line = bench.dumps_strict(out, db.LINE_LIMIT)
    back = json.loads(line)
    assert len(line) < 4096
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in back, key
    assert back["scaling"] == "strong" and back["value"] == 181000.0 and back["dtype"] == "f32" and back["n_gpus"] == 2
    assert back["config"]["workload"] == "sharded_k4096_f32" and back["config"]["transport"] == "xgmi"
    assert back["one_gpu_value"] == 231000.0 and abs(back["sharded_speedup"] - 181.0 / 231.0) < 1e-5
    assert back["replicas"]["value"] == 9e5 and back["replicas"]["scaling"] == "weak" and back["replicas"]["workload"] == "iiwa_14_7_k50_f64"
    assert back["single_reduction"]["pcg_us_per_iter"] == 3.9 and back["out_of_loop_ms"] == 0.02
    sh = back["config"]["sharded"]
    assert set(sh) == set(db.DEFAULT_RIDERS) - {db.LEAD}
    assert abs(sh["sharded_s32_k1024_f32"]["speedup_vs_one_gpu"] - 181.0 / 231.0) < 1e-5 and "error" in sh["sharded_k262144_f32"]
    # the lead rider failed: the replicas value stays the headline, the key is there and null
    out2 = db.attach_riders(copy.deepcopy(rep), {"sharded_k4096_f32": {"error": "rider child job did not deliver"}})
    assert out2["sharded_speedup"] is None and out2["scaling"] == "weak" and out2["value"] == 9e5 and "lead_error" in out2
    json.loads(bench.dumps_strict(out2, db.LINE_LIMIT))


def test_fast_list_conversion_matches_the_numpy_narrowing():
    """bindings/fastseq (include/gato_pyseq.h, also what the pybind11 module copies its arguments with): Python lists, tuples
    and numpy arrays -> float32 / float64 / int32 buffers with the values std::vector<float> / <int> casters would hold
    (double -> float by rounding, ints where floats are expected), and the outputs back as lists of Python floats."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "bindings", "fastseq"), "-s"])
    from gato_python_amd import _gato_fastseq as fs
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(5000) * 10.0 ** rng.integers(-30, 30, 5000)).tolist() + [0, -3, 7, True, np.float32(1.25), np.float64(-2.5), np.int64(9)]
    want = np.asarray([float(v) for v in x], np.float64)
    assert np.array_equal(np.frombuffer(fs.pack(x, "f"), np.float32), want.astype(np.float32))
    assert np.array_equal(np.frombuffer(fs.pack(tuple(x), "d"), np.float64), want)
    for dt in (np.float32, np.float64, np.int32, np.int64, np.uint32):
        a = (rng.standard_normal(100) * 50).astype(dt)
        assert np.array_equal(np.frombuffer(fs.pack(a, "f"), np.float32), a.astype(np.float32))
        assert np.array_equal(np.frombuffer(fs.pack(a[::2], "d"), np.float64), a[::2].astype(np.float64))     # non-contiguous view
    idx = rng.integers(0, 2 ** 31 - 1, 1000)
    assert np.array_equal(np.frombuffer(fs.pack(idx.tolist(), "i"), np.int32), idx.astype(np.int32))
    assert np.array_equal(np.frombuffer(fs.pack(idx.astype(np.int64), "i"), np.int32), idx.astype(np.int32))
    assert np.array_equal(np.frombuffer(fs.pack(range(7), "i"), np.int32), np.arange(7, dtype=np.int32))
    assert len(fs.pack([], "f")) == 0
    for bad, kind, exc in (("abc", "f", TypeError), ([1.5], "i", TypeError), ([2 ** 40], "i", OverflowError), ([[1.0]], "f", TypeError),
                           (3.0, "f", TypeError), ([None], "d", TypeError), (np.ones(3), "i", TypeError),
                           (np.asarray([1, 2 ** 40]), "i", OverflowError), (np.asarray([2 ** 31], np.uint32), "i", OverflowError),
                           (np.asarray([-1, 5], np.int64), "f", None)):
        if exc is None:
            assert np.array_equal(np.frombuffer(fs.pack(bad, kind), np.float32), bad.astype(np.float32))
            continue
        with pytest.raises(exc):
            fs.pack(bad, kind)
    out = fs.unpack(np.asarray([1.5, -2.25, 3e-40], np.float32), "f")
    assert out == [float(np.float32(v)) for v in (1.5, -2.25, 3e-40)] and all(type(v) is float for v in out)
    assert fs.unpack(np.asarray([1e300, -0.0]), "d") == [1e300, -0.0]
    # and the drop-in's own conversion goes through it
    from gato_python_amd import linsys
    assert linsys._fs is fs or linsys._fs is not None
    assert np.array_equal(linsys._from_list(x, "f", np.float32), want.astype(np.float32))
