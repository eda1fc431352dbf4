set -e
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/ab_libs.py reps=2 build/ab/libgato_drA.so build/ab/libgato_drB.so build/ab/libgato_drC.so > gpurun_out/r4_ab4.log 2>&1
echo "== dense layout of the same library" >> gpurun_out/r4_ab4.log
python tools/ab_libs.py reps=1 mixed_dense=1 build/ab/libgato_drA.so >> gpurun_out/r4_ab4.log 2>&1
