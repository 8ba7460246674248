# SQ counters of one bf16 contraction (tools/micro/pgemm_ablate.py), two passes.  usage: bash tools/probes/pmc_pgemm.sh fwd:128
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PG_ONLY=$1 PG_REPS=30
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmcp_a -- python3 $R/tools/micro/pgemm_ablate.py > $R/gpurun_out/pmcp_a.log 2>&1 || exit 1
python3 $R/tools/pmc_quick.py $R/gpurun_out/pmcp_a pgemm_kernel
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES --output-format csv -d $R/gpurun_out/pmcp_b -- python3 $R/tools/micro/pgemm_ablate.py > $R/gpurun_out/pmcp_b.log 2>&1 || exit 1
python3 $R/tools/pmc_quick.py $R/gpurun_out/pmcp_b pgemm_kernel
