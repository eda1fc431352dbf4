"""Generates the committed golden fixtures under tests/golden/.

The reference (CUDA-only) cannot be run anywhere in this pipeline, so these vectors come from
the fp64 numpy restatement (oracle/gato_oracle.py) and are pinned independently by the dense
fp64 KKT solve - the reference test's own oracle construction (test_pendulum_5.py:28-37) with
rho added as the solver adds it.  Inputs of pendulum.json are the reference-owned literals of
test_pendulum_5.py:9-24.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gato_python_amd import synth            # noqa: E402
from oracle import gato_oracle as o          # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    # ---- pendulum: inputs verbatim + every intermediate in reference memory order ------------
    p = synth.pendulum_system()
    P = synth.PENDULUM
    out = o.linsys_solve(*p.csr_args(), p.S, p.C, p.K, P["exit_tol"], P["max_iters"], p.rho,
                         dtype=np.float64, return_all=True)
    dz_d, lam_d = synth.dense_kkt_solve(p, with_rho=True)
    dz_0, lam_0 = synth.dense_kkt_solve(p, with_rho=False)
    out32 = o.linsys_solve(*p.csr_args(), p.S, p.C, p.K, P["exit_tol"], P["max_iters"], p.rho,
                           dtype=np.float32, return_all=True)
    doc = dict(
        source="inputs: reference test_pendulum_5.py:9-24; expected: oracle/gato_oracle.py fp64 + dense KKT solve",
        inputs={k: P[k] for k in ("S", "C", "K", "G_row", "G_col", "G_val", "C_row", "C_col", "C_val", "g_val",
                                  "c_val", "input_lambda", "testiters", "exit_tol", "max_iters", "warm_start", "rho")},
        expected=dict(
            G_dense=out["G_dense"].tolist(), C_dense=out["C_dense"].tolist(), S=out["S"].tolist(),
            Pinv=out["Pinv"].tolist(), gamma=out["gamma"].tolist(), Ginv=out["Ginv"].tolist(),
            eta=out["eta"], iters_f64=int(out["iters"]), iters_f32=int(out32["iters"]),
            lam=out["lam"].tolist(), dz=out["dz"].tolist(),
            dense_kkt_lam=lam_d.tolist(), dense_kkt_dz=dz_d.tolist(),
            dense_kkt_norho_lam=lam_0.tolist(), dense_kkt_norho_dz=dz_0.tolist()),
    )
    with open(os.path.join(HERE, "pendulum.json"), "w") as f:
        json.dump(doc, f, indent=1)

    # ---- seeded synthetic IIWA-shaped system 14/7/50 (BASELINE config 2) ------------------------
    s = synth.make_system(14, 7, 50, seed=0)
    out = o.linsys_solve(*s.csr_args(), 14, 7, 50, 1e-6, 100, s.rho, dtype=np.float64, return_all=True)
    tight = o.linsys_solve(*s.csr_args(), 14, 7, 50, 1e-15, 1000, s.rho, dtype=np.float64, return_all=True)
    dz_d, lam_d = synth.dense_kkt_solve(s)
    np.savez_compressed(
        os.path.join(HERE, "iiwa_14_7_50_seed0.npz"),
        gamma=out["gamma"], lam=out["lam"], dz=out["dz"], iters=np.int32(out["iters"]),
        eta=np.asarray(out["eta"]), S_block_k1=out["S"][588:1176], Pinv_block_k1=out["Pinv"][588:1176],
        S_sum=np.float64(out["S"].sum()), Pinv_sum=np.float64(out["Pinv"].sum()),
        Ginv_sum=np.float64(out["Ginv"].sum()),
        lam_tight=tight["lam"], dz_tight=tight["dz"], iters_tight=np.int32(tight["iters"]),
        dense_lam=lam_d, dense_dz=dz_d)
    # dense-Q variant at 32/16/12 (BASELINE config 5 shape, small K)
    s = synth.make_system(32, 16, 12, seed=5, dense_q=True)
    out = o.linsys_solve(*s.csr_args(), 32, 16, 12, 1e-12, 500, s.rho, dtype=np.float64, return_all=True)
    dz_d, lam_d = synth.dense_kkt_solve(s)
    np.savez_compressed(os.path.join(HERE, "s32_c16_k12_seed5_denseq.npz"), gamma=out["gamma"], lam=out["lam"],
                        dz=out["dz"], iters=np.int32(out["iters"]), eta=np.asarray(out["eta"]),
                        dense_lam=lam_d, dense_dz=dz_d)
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
