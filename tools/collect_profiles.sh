#!/bin/bash
# Collects, on the GPU box, what the summaries under profiles/ are made from: the benchmark line, a rocprofv3 kernel trace with
# statistics of the same command, and three counter passes (each in a run of its own) of a short eager run.
# usage: bash tools/collect_profiles.sh <tag>      (run from the repository root; writes under gpurun_out/<tag>_*)
set -e
TAG=${1:-r3}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
export TMPDIR=/tmp
timeout -k 10 400 python bench.py --steps 50 --warmup 10 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o ${TAG} -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-extra > $OUT/${TAG}_stats.log 2>&1
# the same trace with the critic passes on one stream: per-kernel durations free of overlap with other streams' kernels
# (what the benchmark's event-timed roofline entry measures)
VEGANS_HIP_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats1 -o ${TAG}s -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-extra > $OUT/${TAG}_stats1.log 2>&1
export VEGANS_HIP_GRAPH=0
export VEGANS_HIP_STREAMS=0
PMC_CMD="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-bf16-extra --allow-eager"
timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/${TAG}_pmc1 -o p1 -- $PMC_CMD > $OUT/${TAG}_pmc1.log 2>&1
timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/${TAG}_pmc2 -o p2 -- $PMC_CMD > $OUT/${TAG}_pmc2.log 2>&1
timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/${TAG}_pmc3 -o p3 -- $PMC_CMD > $OUT/${TAG}_pmc3.log 2>&1
# the bf16 data path: kernel trace of replayed iterations, then HBM traffic of an eager run
unset VEGANS_HIP_GRAPH
unset VEGANS_HIP_STREAMS
BF_CMD="python $ROOT/bench.py --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-bf16-extra"
VEGANS_HIP_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_bf16_stats -o ${TAG}b -- $BF_CMD > $OUT/${TAG}_bf16_stats.log 2>&1
export VEGANS_HIP_GRAPH=0
export VEGANS_HIP_STREAMS=0
BF_PMC="python $ROOT/bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-bf16-extra --allow-eager"
timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/${TAG}_bf16_pmc1 -o p1 -- $BF_PMC > $OUT/${TAG}_bf16_pmc1.log 2>&1
timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/${TAG}_bf16_pmc2 -o p2 -- $BF_PMC > $OUT/${TAG}_bf16_pmc2.log 2>&1
timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/${TAG}_bf16_pmc3 -o p3 -- $BF_PMC > $OUT/${TAG}_bf16_pmc3.log 2>&1
echo collected
