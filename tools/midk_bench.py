import sys, os, json
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from stream_bench import run
for K in (8192, 12288, 16384, 24576, 32768, 65536):
    for mode in (0, 2):
        try:
            print(json.dumps(run(14, 7, K, np.float32, mode=mode)), flush=True)
        except Exception as e:
            print(K, mode, "ERR", str(e)[:100], flush=True)
