#!/bin/bash
# Counter passes over the conv micro-benchmark: where the waves of the igemm kernels wait (vector-memory latency, LDS).
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_latency
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/p1 -o p1 -- python $ROOT/tools/micro/igemm_ablate.py > $OUT/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d $OUT/p2 -o p2 -- python $ROOT/tools/micro/igemm_ablate.py > $OUT/p2.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv --pmc TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum -d $OUT/p3 -o p3 -- python $ROOT/tools/micro/igemm_ablate.py > $OUT/p3.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum -d $OUT/p4 -o p4 -- python $ROOT/tools/micro/igemm_ablate.py > $OUT/p4.log 2>&1
for p in p1 p2 p3 p4; do python $ROOT/tools/pmc_quick.py $OUT/$p igemm > $OUT/$p.txt; done
find $OUT -name "*.csv" -size +2M -delete
echo ok
