cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export GATO_BENCH_ONE_GPU=1
timeout -k 10 700 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r4_rehearse2.out 2> gpurun_out/r4_rehearse2.err
echo "rc=$?" >> gpurun_out/r4_rehearse2.err
tail -c 3500 gpurun_out/r4_rehearse2.out
tail -5 gpurun_out/r4_rehearse2.err
