"""GPU suite: a fixed-seed slice of tools/solve_fuzz.py - random shape, K, type, entry (reference surface, device entry, stage
entries, block entry, batch), recurrence, launch options, tolerance, iteration cap, rho - every solve against the oracle.  The
long runs (thousands of cases, other seeds) are run by hand: profiles/r05_solve_fuzz.txt."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_solves_match_the_oracle():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import solve_fuzz
    rng = np.random.default_rng(1)
    bad, ran = [], 0
    for i in range(150):
        msg, ok = solve_fuzz.case(rng, i)
        ran += not msg.endswith("not run")
        if not ok:
            bad.append(msg)
    assert ran >= 140 and not bad, "\n".join(bad)


def test_random_csr_inputs_scatter_bit_for_bit():
    """A fixed-seed slice of tools/csr_fuzz.py: CSR rows shuffled, thinned, with explicit zeros, repeated columns, columns outside
    the block structure, empty rows and empty matrices - the gather kernel, the stage path and the fused assembly launch of a
    whole solve must leave G_dense / C_dense bit for bit as the restatements of csr_to_custom_G / _C do."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import csr_fuzz
    rng = np.random.default_rng(1)
    bad = []
    for i in range(80):
        msg, ok = csr_fuzz.case(rng, i)
        if not ok:
            bad.append(msg)
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("args", [("14", "7", "f64", "700"), ("14", "7", "f32", "1500"), ("32", "16", "f32", "300", "pcg_variant=1")])
def test_both_sides_of_the_planner_boundaries(args):
    """tools/boundary_sweep.py on a short range: the K at which the planner changes the launch (workgroup size of the one-workgroup
    kernels, one -> several workgroups, row layout ...) are located and a whole solve is held against the oracle at K = b - 1, b,
    b + 1 of each.  The long ranges (to K = 140 000, the semi-resident forms, the ring, streaming) are run by hand:
    profiles/r05_boundary_sweep.txt."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "boundary_sweep.py"), *args], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "BOUNDARIES ok 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
