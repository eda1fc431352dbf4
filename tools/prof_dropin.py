"""cProfile of the Python side of gpu_library.linsys_solve at 14/7/50."""
import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["GATO_VERBOSE"] = "0"
import gpu_library
from gato_python_amd import synth
s = synth.make_system(14, 7, 50, seed=0)
args = (s.G_row, s.G_col, s.G_val, s.C_row, s.C_col, s.C_val, s.g, s.c, np.zeros(700), 1, 1e-6, 100, False, s.rho)
for _ in range(10): gpu_library.linsys_solve(*args)
pr = cProfile.Profile(); pr.enable()
for _ in range(300): gpu_library.linsys_solve(*args)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
