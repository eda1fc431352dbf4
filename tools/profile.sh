#!/bin/bash
# Run ON THE GPU BOX (through gpurun):  bash tools/profile.sh [tag]
# Produces gpurun_out/prof_<tag>_{stats,fetch,write}/ ; summarise with tools/summarize_profile.py
# and copy the summaries into profiles/.  Counters are collected in their own passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950), never together with API tracing.
set -o pipefail
TAG=${1:-r01}
PART=${2:-all}          # all | main (bench passes + issue counters) | cache (per-counter passes of the HBM-bound launches)
R=$PWD
ARGS=${BENCH_ARGS:---steps 50 --warmup 5 --no-cpu}
cd /tmp && export TMPDIR=/tmp
if [ "$PART" != "cache" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_bench.json 2> $R/gpurun_out/prof_${TAG}_stats.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -- python3 $R/bench.py $ARGS > /dev/null 2> $R/gpurun_out/prof_${TAG}_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_write -- python3 $R/bench.py $ARGS > /dev/null 2> $R/gpurun_out/prof_${TAG}_write.err || exit 1
# matrix-core evidence for the assembly stages (the only dense contractions of the path)
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_mfma -- python3 $R/bench.py $ARGS > /dev/null 2> $R/gpurun_out/prof_${TAG}_mfma.err || exit 1
# what bounds the register-resident loop of the default workload: VALU / LDS activity and waits (two passes, default workload only)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_issue1 -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-sweep > /dev/null 2> $R/gpurun_out/prof_${TAG}_issue1.err || exit 1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_issue2 -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-sweep > /dev/null 2> $R/gpurun_out/prof_${TAG}_issue2.err || exit 1
# the same counters for the multi-workgroup resident launch: W = 15 on one XCD (14/7/512 f32) and W = 114 across the chip (14/7/4096 f32)
for WLD in iiwa_14_7_k512_f32 iiwa_14_7_k4096_f32; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_issue1_$WLD -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-sweep --workload $WLD > /dev/null 2> $R/gpurun_out/prof_${TAG}_issue1_$WLD.err || exit 1
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_issue2_$WLD -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-sweep --workload $WLD > /dev/null 2> $R/gpurun_out/prof_${TAG}_issue2_$WLD.err || exit 1
done
echo profile $TAG done
fi
[ "$PART" = "main" ] && exit 0
# cache behaviour of the HBM-bound persistent launches (K = 131072 f32, 10 iterations per launch: the LDS-DMA ring = auto,
# and the semi-resident launch): every counter in a pass of its own (four TCP/TCC counters in one pass make rocprofv3 abort
# on gfx950: "exceeds the capabilities of the hardware to collect"); summarised into profiles/<tag>_cache_counters.csv
for GRP in "TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_sum" "TCC_HIT_sum" "TCC_MISS_sum" "TCC_EA0_RDREQ_sum" "TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $GRP --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_cache_$GRP -- python3 $R/tools/semi_one.py 131072 > /dev/null 2> $R/gpurun_out/prof_${TAG}_cache_$GRP.err || echo "pass $GRP failed"
  rocprofv3 --pmc $GRP --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_cache_semi_$GRP -- python3 $R/tools/semi_one.py 131072 1 > /dev/null 2> $R/gpurun_out/prof_${TAG}_cache_semi_$GRP.err || echo "pass semi $GRP failed"
done
echo cache passes $TAG done
