# SQ counters of one fat contraction (tools/conv_micro.py), two passes.  usage: bash tools/probes/pmc_igemm.sh MODE N H W Ci Co k s p
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmci_a -- python3 $R/tools/conv_micro.py "$@" 50 > $R/gpurun_out/pmci_a.log 2>&1 || exit 1
python3 $R/tools/pmc_quick.py $R/gpurun_out/pmci_a gemm_kernel
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES --output-format csv -d $R/gpurun_out/pmci_b -- python3 $R/tools/conv_micro.py "$@" 50 > $R/gpurun_out/pmci_b.log 2>&1 || exit 1
python3 $R/tools/pmc_quick.py $R/gpurun_out/pmci_b gemm_kernel
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES --output-format csv -d $R/gpurun_out/pmci_c -- python3 $R/tools/conv_micro.py "$@" 50 > $R/gpurun_out/pmci_c.log 2>&1 || exit 1
python3 $R/tools/pmc_quick.py $R/gpurun_out/pmci_c gemm_kernel
