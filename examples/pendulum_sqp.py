"""The callers either side of the hot path: a Gauss-Newton / SQP loop for the pendulum swing-up whose
every step is one `gpu_library.linsys_solve` call - the KKT producer (gato_python_amd/kkt.py) in front, the line search
and trajectory update behind.      python examples/pendulum_sqp.py [K]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GATO_VERBOSE", "0")
import gpu_library                                     # noqa: E402
from gato_python_amd import kkt                        # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
plant, dt = kkt.PendulumPlant(), 0.05
xs, xg = np.array([0.0, 0.0]), np.array([np.pi, 0.0])
Q, R, QF = np.diag([0.1, 0.01]), np.array([[0.01]]), np.diag([100.0, 10.0])
gpu_library.set_precision("f64")


def cost(x, u):
    e = x - xg
    return 0.5 * np.einsum("ki,ij,kj", e[:-1], Q, e[:-1]) + 0.5 * e[-1] @ QF @ e[-1] + 0.5 * R[0, 0] * float((u ** 2).sum())


u = np.zeros((K - 1, 1))
x = kkt.rollout(plant, xs, u, dt)
for it in range(30):
    p = kkt.get_kkt(plant, x, u, xs, xg, dt, Q, R, QF, rho=1e-3)
    lam, dz = gpu_library.linsys_solve(p.G_row, p.G_col, p.G_val, p.C_row, p.C_col, p.C_val, p.g, p.c,
                                       [0.0] * (2 * K), 1, 1e-12, 500, False, p.rho)
    dz = np.asarray(dz)
    du = np.array([dz[k * 3 + 2] for k in range(K - 1)])[:, None]
    # the KKT system [G C'; C 0][dz; lam] = [g; c] gives the NEGATIVE of the Newton step; backtrack on the rollout cost
    c0, step = cost(x, u), 1.0
    while step > 1e-4:
        un = u - step * du
        xn = kkt.rollout(plant, xs, un, dt)
        if cost(xn, un) < c0:
            break
        step *= 0.5
    if step <= 1e-4:
        break
    x, u = xn, un
    print(f"iteration {it:2d}: cost {cost(x, u):10.4f}  step {step:.3f}  PCG iterations {gpu_library.last_stats()['iters']}")
print(f"final angle {x[-1, 0]:.4f} rad (target {np.pi:.4f}), velocity {x[-1, 1]:.4f}")
