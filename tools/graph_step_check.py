"""Does replaying the whole step (assembly + PCG + dz) as a captured graph shorten it?  14/7/50 f64, device-resident inputs."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gato_python_amd import synth
from gato_python_amd.solver import Solver


def main():
    S, C, K = 14, 7, 50
    s = synth.make_system(S, C, K, seed=0)
    sol = Solver(S, C, K, np.float64)
    dev = sol.upload_system(s)
    lam, dz = sol.new(S * K), sol.new(sol.N)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(5):
            sol.linsys(*dev, 0.0, 100, s.rho, lam, dz)
        st.synchronize()
        n = 500
        t0 = time.perf_counter()
        for _ in range(n):
            sol.linsys(*dev, 0.0, 100, s.rho, lam, dz)
        st.synchronize()
        plain = (time.perf_counter() - t0) / n * 1e6
        ref = lam.clone()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=st):
                sol.linsys(*dev, 0.0, 100, s.rho, lam, dz)
            for _ in range(5):
                g.replay()
            st.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                g.replay()
            st.synchronize()
            graph = (time.perf_counter() - t0) / n * 1e6
            print(f"plain stream: {plain:.1f} us per step; graph replay: {graph:.1f} us per step; same result: {torch.equal(lam, ref)}")
        except Exception as e:      # noqa: BLE001
            print(f"plain stream: {plain:.1f} us per step; capture failed: {type(e).__name__}: {e}")


if __name__ == "__main__":
    main()
