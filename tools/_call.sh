cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r4_gputests2.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4_gputests2.log
tail -4 gpurun_out/r4_gputests2.log
python bench.py > gpurun_out/r4_bench1.jsonl 2> gpurun_out/r4_bench1.err
tail -c 1200 gpurun_out/r4_bench1.jsonl
python tools/dropin_latency.py > gpurun_out/r4_dropin0.log 2>&1
python tools/dropin_latency.py pybind11 >> gpurun_out/r4_dropin0.log 2>&1
python tools/scaling_inputs.py > gpurun_out/r4_scaling_inputs.log 2>&1
