"""One-GPU inputs of the multi-GPU expectation table: us per PCG iteration of the launch ONE rank of an
N-way knot split would run (K / N knots on one GPU) for the BASELINE shapes, and of the whole system on one GPU."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tune_pcg import run


def main():
    out = {}
    variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0          # 1: the single-reduction recurrence (one exchange per iteration)
    cases = [(14, 7, K, np.float32) for K in ((512, 1024, 2048, 4096, 8192, 12288, 16384, 32768, 65536, 131072, 262144) if not variant else
                                              (512, 1024, 2048, 4096, 8192))]
    cases += [(32, 16, K, np.float32) for K in (128, 256, 512, 1024)]
    for (S, C, K, dt) in cases:
        it = 100 if K <= 16384 else 20
        r = run(S, C, K, dt, iters=it, reps=5, opts={"pcg_variant": variant})
        key = f"{S}/{C}/{K}/{np.dtype(dt).name}"
        out[key] = r
        print(key, json.dumps(r), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/r5_scaling_inputs" + ("_single_reduction" if variant else "") + ".json", "w"), indent=1)


if __name__ == "__main__":
    main()
