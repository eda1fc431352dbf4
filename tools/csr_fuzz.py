"""Random CSR inputs for the scatter (A1: gather kernel, stage path of a whole solve, fused assembly launch) against the C and numpy
restatements of csr_to_custom_G / csr_to_custom_C (gato_schur.cuh:674-756), BIT FOR BIT: rows shuffled, entries dropped, explicit
zeros, the same column several times in a row (the last entry in storage order wins), columns OUTSIDE the block structure (anywhere in
[0, N): the reference folds them with col % n and drops what lies right of the row's block), empty rows, empty matrices.
      python tools/csr_fuzz.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GATO_NO_TUNE", "1")
import torch
from gato_python_amd import synth
from gato_python_amd.solver import Solver
from oracle import c_oracle as co
from oracle import gato_oracle as o


def mutate(rng, indptr, indices, data, ncols, p, g_blocks=None):
    """g_blocks = (S, C): the matrix is G - a column off the structure stays in the row's KIND of block (state rows: col % n < S,
    control rows: col % n >= S; any knot): the reference's index arithmetic for the other kind (gato_schur.cuh:697-700 with
    in_set_row - STATE_SIZE < 0) leaves the array, such input is not a G."""
    idx, dat, ptr = [], [], [0]
    for r in range(len(indptr) - 1):
        cols = list(indices[indptr[r]:indptr[r + 1]]); vals = list(data[indptr[r]:indptr[r + 1]])
        u = rng.random()
        if u < p["empty"]:
            cols, vals = [], []
        else:
            keep = rng.random(len(cols)) >= p["drop"]
            cols = [c for c, k in zip(cols, keep) if k]; vals = [v for v, k in zip(vals, keep) if k]
            for _ in range(int(rng.poisson(p["dup"]))):
                if cols:
                    j = int(rng.integers(0, len(cols))); at = int(rng.integers(0, len(cols) + 1))
                    cols.insert(at, cols[j]); vals.insert(at, float(rng.standard_normal()))
            for _ in range(int(rng.poisson(p["wild"]))):
                at = int(rng.integers(0, len(cols) + 1))
                cw = int(rng.integers(0, ncols))
                if g_blocks is not None:
                    S_, C_ = g_blocks
                    n_ = S_ + C_
                    kn = cw // n_
                    cw = kn * n_ + (int(rng.integers(0, S_)) if r % n_ < S_ else S_ + int(rng.integers(0, C_)))
                    if cw >= ncols:
                        cw = r
                cols.insert(at, cw); vals.insert(at, float(rng.standard_normal()))
            for _ in range(int(rng.poisson(p["zero"]))):
                if cols:
                    vals[int(rng.integers(0, len(cols)))] = 0.0
            if rng.random() < p["shuffle"]:
                perm = rng.permutation(len(cols))
                cols = [cols[i] for i in perm]; vals = [vals[i] for i in perm]
        idx += cols; dat += vals
        ptr.append(len(idx))
    return np.asarray(ptr, np.int32), np.asarray(idx, np.int32).reshape(-1), np.asarray(dat, np.float64).reshape(-1)


def case(rng, i):
    S, C = [(14, 7), (2, 1), (12, 6), (32, 16), (4, 2), (6, 3)][int(rng.integers(0, 6))]
    K = int(rng.integers(1, 40)) if rng.integers(0, 3) else int(rng.integers(40, 700))
    dt = np.float64 if rng.integers(0, 2) else np.float32
    s = synth.make_system(S, C, K, seed=7000 + i, dense_q=bool(rng.integers(0, 2)))
    lvl = rng.random()
    p = dict(empty=0.05 * lvl, drop=0.3 * lvl * rng.random(), dup=1.0 * lvl * rng.random(), wild=0.8 * lvl * rng.random(), zero=0.5 * lvl, shuffle=rng.random())
    if rng.integers(0, 25) == 0:
        p["empty"] = 1.0                                         # an empty matrix
    G = mutate(rng, s.G_row, s.G_col, s.G_val, s.N, p, g_blocks=(S, C))
    Cm = mutate(rng, s.C_row, s.C_col, s.C_val, s.N, p)
    rho = float(rng.choice([1e-3, 0.0, 0.5]))
    tag = f"case {i}: {S}/{C}/{K} {np.dtype(dt).name} nnz G {len(s.G_val)} -> {len(G[2])}, C {len(s.C_val)} -> {len(Cm[2])} rho {rho:g}"
    Gd_c, Cd_c = co.convert(*G, *Cm, S, C, K, rho, dt)
    Gd_n, Cd_n = o.convert(*G, *Cm, S, C, K, rho, dt)
    ok = np.array_equal(Gd_c, Gd_n) and np.array_equal(Cd_c, Cd_n)
    notes = [] if ok else ["the two CPU restatements differ"]
    sol = Solver(S, C, K, dt)
    i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.int32)).cuda() if len(a) else torch.zeros(1, dtype=torch.int32, device="cuda")[:0]
    val = lambda a: sol.to_device(a) if len(a) else sol.new(1)[:0]
    dev = (i32(G[0]), i32(G[1]), val(G[2]), i32(Cm[0]), i32(Cm[1]), val(Cm[2]), sol.to_device(s.g), sol.to_device(s.c))
    Gd, Cd = sol.convert(*dev[:6], rho)
    torch.cuda.synchronize()
    g1 = np.array_equal(Gd.cpu().numpy(), Gd_c) and np.array_equal(Cd.cpu().numpy(), Cd_c)
    if not g1:
        notes.append("gather kernel differs")
    for mode in (1, 2):                                          # stage path / fused launch of a whole solve (the solve itself may be singular: only the blocks)
        sol.set_option("asm_mode", mode)
        lam, dz = sol.new(S * K), sol.new(sol.N)
        try:
            sol.linsys(*dev, 1e-6, 2, rho, lam, dz)
            torch.cuda.synchronize()
            gm = np.array_equal(sol.read_buffer("G_dense"), Gd_c) and np.array_equal(sol.read_buffer("C_dense"), Cd_c)
        except Exception as e:                                   # noqa: BLE001
            gm = False
            notes.append(f"asm_mode {mode} raised {str(e)[:80]}")
        if not gm:
            notes.append(f"asm_mode {mode} blocks differ")
        ok = ok and gm
    sol.close()
    return tag + (" | " + "; ".join(notes) if notes else ""), ok and g1


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for i in range(n):
        msg, ok = case(rng, i)
        if not ok or i % 50 == 0:
            print(("ok   " if ok else "FAIL ") + msg, flush=True)
        bad += not ok
    print("CSR FUZZ", "FAILED" if bad else "ok", bad, "of", n)
    sys.exit(1 if bad else 0)
