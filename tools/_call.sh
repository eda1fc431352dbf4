cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "foreign_kernel or captured or pybind11 or dropin or pendulum_golden" > gpurun_out/r4_gputests3.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4_gputests3.log
tail -6 gpurun_out/r4_gputests3.log
python tools/coop_cost.py > gpurun_out/r4_coop_cost.log 2>&1
python tools/dropin_latency.py > gpurun_out/r4_dropin1.log 2>&1
python tools/dropin_latency.py pybind11 >> gpurun_out/r4_dropin1.log 2>&1
