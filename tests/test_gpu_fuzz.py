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
