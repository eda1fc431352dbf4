"""One-off check at a size whose block-row arrays hold more than 2^31 ELEMENTS (32/16 with K = 700 000: S and Pinv 2.15 G floats =
8.6 GB each; CSR C 1.1 G entries): every index computation of the stage kernels, the streaming / ring PCG launches and dz in 64 bits.
A whole fp32 solve with a fixed iteration count against the C oracle.  Needs ~45 GB of device memory and ~60 GB of host memory.
      python tools/max_size_check.py [S C K iters [f32|f64 [option=value ...]]]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gato_python_amd import synth
from gato_python_amd.solver import Solver
from oracle import c_oracle as co

S, C, K, mi = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (32, 16, 700000, 12)
dt = np.float64 if len(sys.argv) > 5 and sys.argv[5] == "f64" else np.float32
opts = dict(kv.split("=") for kv in sys.argv[6:])


def big_system(S, C, K, seed, rho=1e-3):
    """The generator of synth.make_blocks / blocks_to_csr (diagonal Q and R, A = -(I + 0.01 N), B = -0.1 N, identity blocks) written
    for whole arrays: CSR rows of C hold [A_k row | B_k row | 1], sorted columns; values drawn in fp32, knot chunks of 20 000."""
    rng = np.random.default_rng(seed)
    n = S + C
    N = n * K - C
    G_val = np.empty(N, np.float32)
    gq = rng.uniform(0.1, 10.0, (K, S)).astype(np.float32)
    gq[K - 1] *= 100.0
    gr = rng.uniform(0.01, 1.0, (K - 1, C)).astype(np.float32)
    Gv = np.zeros((K, n), np.float32)
    Gv[:, :S] = gq
    Gv[:K - 1, S:] = gr
    G_val[:] = Gv.reshape(-1)[:N]
    G_row = np.arange(N + 1, dtype=np.int32)
    G_col = np.arange(N, dtype=np.int32)
    per = n + 1                                         # entries of a row of row-block k >= 1
    nnzC = S + (K - 1) * S * per
    C_val = np.empty(nnzC, np.float32)
    C_col = np.empty(nnzC, np.int32)
    C_val[:S] = 1.0
    C_col[:S] = np.arange(S)
    C_row = np.concatenate([np.arange(S + 1, dtype=np.int64), S + per * np.arange(1, (K - 1) * S + 1, dtype=np.int64)]).astype(np.int32)
    eye = np.eye(S, dtype=np.float32)
    for k0 in range(0, K - 1, 20000):
        k1 = min(K - 1, k0 + 20000)
        m = k1 - k0
        blk = np.empty((m, S, per), np.float32)
        blk[:, :, :S] = -(eye[None] + 0.01 * rng.standard_normal((m, S, S), dtype=np.float32))
        blk[:, :, S:n] = -0.1 * rng.standard_normal((m, S, C), dtype=np.float32)
        blk[:, :, n] = 1.0
        cols = np.empty((m, S, per), np.int32)
        base = (np.arange(k0, k1, dtype=np.int64) * n)[:, None, None]
        cols[:, :, :n] = base + np.arange(n)[None, None, :]
        cols[:, :, n] = (base[:, :, 0] + n) + np.arange(S)[None, :]
        lo = S + k0 * S * per
        C_val[lo:lo + m * S * per] = blk.reshape(-1)
        C_col[lo:lo + m * S * per] = cols.reshape(-1)
    g = np.zeros((K, n), np.float32)
    g[:, :S] = rng.standard_normal((K, S), dtype=np.float32)
    g[:K - 1, S:] = rng.standard_normal((K - 1, C), dtype=np.float32)
    c = (0.1 * rng.standard_normal((K, S), dtype=np.float32))
    c[0] = 0.0
    return synth.KKTSystem(S, C, K, G_row, G_col, G_val, C_row, C_col, C_val, g.reshape(-1)[:N].copy(), c.reshape(-1), rho)


t0 = time.time()
s = big_system(S, C, K, seed=5)
print(f"system {S}/{C}/{K}: N = {s.N}, nnz(G) = {len(s.G_val)}, nnz(C) = {len(s.C_val)}, bd elements = {3 * S * S * K} (2^31 = {1 << 31})  [{time.time() - t0:.0f} s]", flush=True)
sol = Solver(S, C, K, dt)
for k_, v_ in opts.items():
    sol.set_option(k_, int(v_))
d = sol.upload_system(s)
lam = torch.full((S * K,), float("nan"), dtype=sol.dtype, device="cuda")
dz = torch.full((sol.N,), float("nan"), dtype=sol.dtype, device="cuda")
sol.set_option("time_stages", 1)
sol.linsys(*d, 0.0, mi, s.rho, lam=lam, dz=dz)
torch.cuda.synchronize()
sol.check_status()
print(f"GPU solve done: mode {sol.get_option('last_mode')} groups {sol.get_option('last_groups')} semi {sol.get_option('last_semi')} stage ms {sol.last_stage_ms()}  "
      f"[{time.time() - t0:.0f} s]", flush=True)
lam_g, dz_g = lam.cpu().numpy(), dz.cpu().numpy()
del d, lam, dz
sol.close()
torch.cuda.empty_cache()
lam_o, dz_o, it_o = co.linsys_solve(*s.csr_args(), S, C, K, 0.0, mi, s.rho, dtype=dt)
print(f"oracle done [{time.time() - t0:.0f} s]", flush=True)
rel = lambda a, b: float(np.abs(a.astype(np.float64) - b).max() / np.abs(b).max())
# per-segment maxima: an index that wraps would hit the far end of the arrays
seg = np.linspace(0, K, 9).astype(int)
errs = [rel(lam_g[a * S:b * S], lam_o[a * S:b * S]) for a, b in zip(seg[:-1], seg[1:])]
el, ed = rel(lam_g, lam_o), rel(dz_g, dz_o)
bar = 5e-3 if dt == np.float32 else 1e-9
ok = np.isfinite(lam_g).all() and np.isfinite(dz_g).all() and el < bar and ed < bar
print(f"{np.dtype(dt).name} {opts}: lambda rel {el:.2e} (by eighths of the knots: {' '.join(f'{e:.1e}' for e in errs)}), dz rel {ed:.2e}, iterations {mi}: {'ok' if ok else 'FAILED'}", flush=True)
sys.exit(0 if ok else 1)
