"""Streaming kernels between register residency (K <= ~13 k at 14/7 f32) and the HBM-bound regime."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stream_bench import run
for K in (16384, 32768, 49152, 65536, 131072):
    print(json.dumps(run(14, 7, K, np.float32, mode=2)), flush=True)
