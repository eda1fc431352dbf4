"""What hipLaunchCooperativeKernel costs a multi-workgroup persistent launch (option coop_launch): launches of ONE iteration
(launch overhead) and of 100 (us per iteration), plain against cooperative, 14/7/512 (15 workgroups on one XCD) and 14/7/4096
(114 workgroups)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tune_pcg import run


def main():
    for (S, C, K) in ((14, 7, 512), (14, 7, 4096)):
        for coop in (0, 1, 0, 1):
            one = run(S, C, K, np.float32, iters=1, reps=40, opts={"coop_launch": coop})
            hun = run(S, C, K, np.float32, iters=100, reps=20, opts={"coop_launch": coop})
            print(f"{S}/{C}/{K} f32 coop_launch={coop}: 1-iteration launch {one['us_per_iter']:.2f} us, 100 iterations {hun['us_per_iter'] * 100:.1f} us "
                  f"= {hun['us_per_iter']:.3f} us per iteration ({hun['groups']} x {hun['threads']})", flush=True)


if __name__ == "__main__":
    main()
