"""Worker of test_extra_shape_library (tests/test_gpu_parity.py): whole solves through a library built with ONE extra
(STATE_SIZE, CONTROL_SIZE) shape (make EXTRA_SHAPES / tools/devbuild.sh; selected by GATO_HIP_LIB before the package is
imported) against the C oracle.  ADVICE r4: the generic launch bounds of such shapes (S = 16: 256-thread one-workgroup
kernels, four waves) take code paths the six default shapes never reach."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                           # noqa: E402,F401
from gato_python_amd import _lib, synth                # noqa: E402
from gato_python_amd.solver import Solver              # noqa: E402
from oracle import c_oracle as co                      # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def main():
    S, C = int(sys.argv[1]), int(sys.argv[2])
    assert (S, C) in _lib.shapes(), (_lib.SO_PATH, _lib.shapes())
    seen = set()
    for K in (1, 2, 7, 31, 32, 33, 50, 64, 300):
        for dt in (np.float32, np.float64):
            for batch in ((1, 3) if K <= 64 else (1,)):
                f64 = dt == np.float64
                tol, mi = (1e-9, 200) if f64 else (1e-4, 80)
                systems = [synth.make_system(S, C, K, seed=40 + b) if K > 1 else synth.blocks_to_csr(*synth.make_blocks(S, C, 1, 40 + b, False))
                           for b in range(batch)]
                sol = Solver(S, C, K, dt, batch=batch)
                lam, dz = sol.new(batch * S * K), sol.new(batch * sol.N)
                if batch == 1:
                    sol.linsys(*sol.upload_system(systems[0]), tol, mi, systems[0].rho, lam, dz)
                else:
                    sol.linsys_batched(*sol.upload_batch(systems), tol, mi, systems[0].rho, lam, dz)
                sol.check_status()
                seen.add((np.dtype(dt).name, sol.get_option("last_pair"), sol.get_option("last_groups"), sol.get_option("last_threads")))
                hl, hz = lam.cpu().numpy().reshape(batch, -1), dz.cpu().numpy().reshape(batch, -1)
                for b, s in enumerate(systems):
                    lam_o, dz_o, it_o = co.linsys_solve(*s.csr_args(), S, C, K, tol, mi, s.rho, dtype=dt)
                    assert np.isfinite(hl[b]).all() and np.isfinite(hz[b]).all(), (K, batch, b)
                    if os.environ.get("GATO_EXTRA_DIAG") == "1":
                        print(K, np.dtype(dt).name, batch, b, "pair/groups/threads/dpp", sol.get_option("last_pair"), sol.get_option("last_groups"),
                              sol.get_option("last_threads"), sol.get_option("last_dpp"), "rel lam", rel(hl[b], lam_o), "rel dz", rel(hz[b], dz_o),
                              "max|dz_o|", np.abs(dz_o).max(), "it_o", it_o, flush=True)
                        continue
                    if K <= 2:          # (one- and two-knot systems: dz is rounding noise around zero where c_0 = 0 and the fp32 iteration
                        if f64:         #  runs past convergence on noise - order-chaotic, tools/past_convergence.py: fp64 only, to solver tolerance)
                            assert rel(hl[b], lam_o) < 1e-5 and np.abs(hz[b] - dz_o).max() < 1e-3 * max(np.abs(dz_o).max(), 1e-3), (K, batch, b)
                    elif f64:
                        assert rel(hl[b], lam_o) < 1e-8 and rel(hz[b], dz_o) < 1e-8, (K, batch, b, rel(hl[b], lam_o), rel(hz[b], dz_o))
                    else:
                        s64 = s.astype(np.float32).astype(np.float64)
                        lam_t, dz_t, _ = co.linsys_solve(*s64.csr_args(), S, C, K, 1e-14, 600, float(np.float32(s.rho)), dtype=np.float64)
                        for got, orc, tr, what in ((hl[b], lam_o, lam_t, "lambda"), (hz[b], dz_o, dz_t, "dz")):
                            eg, eo = rel(got, tr), rel(orc, tr)
                            assert eg <= 2.0 * eo + 5e-6, (what, K, batch, b, eg, eo)
                sol.close()
    print("extra shape ok", S, C, sorted(seen))



if __name__ == "__main__":
    main()
