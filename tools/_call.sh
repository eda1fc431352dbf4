cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/ab_libs.py reps=2 prod build/ab/libgato_prio1.so build/ab/libgato_prio3.so > gpurun_out/r4_ab6.log 2>&1
tail -5 gpurun_out/r4_ab6.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4_gputests8.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4_gputests8.log
tail -4 gpurun_out/r4_gputests8.log
