cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipelined or single_reduction" > gpurun_out/r4_gputests4.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4_gputests4.log
tail -4 gpurun_out/r4_gputests4.log
python - > gpurun_out/r4_pipe_time.log 2>&1 <<'PY'
import sys, numpy as np
sys.path.insert(0, 'tools')
from tune_pcg import run
for (S,C,K,dt) in ((14,7,512,np.float32),(14,7,1024,np.float32),(14,7,2048,np.float32),(14,7,4096,np.float32),(14,7,4096,np.float64),(32,16,1024,np.float32)):
    out=[]
    for sl in (0, 4, 8, 12, 16, 20):
        r = run(S,C,K,dt,reps=20,opts={"pcg_variant":2, "ablate": (sl+1)<<8})
        out.append((sl, round(r['us_per_iter'],3)))
    print(S,C,K,np.dtype(dt).name,"pipelined, by sleep units:",out, r['groups'], r['threads'], flush=True)
PY
