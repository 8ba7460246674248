# SQ counters of the three image-side contractions of DCGAN-64 at batch 256 (tools/micro/thin_ablate.py), two passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmcs_a -- python3 $R/tools/micro/thin_ablate.py > $R/gpurun_out/pmcs_a.log 2>&1 || exit 1
python3 $R/tools/pmc_quick.py $R/gpurun_out/pmcs_a ${1:-kernel}
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES --output-format csv -d $R/gpurun_out/pmcs_b -- python3 $R/tools/micro/thin_ablate.py > $R/gpurun_out/pmcs_b.log 2>&1 || exit 1
python3 $R/tools/pmc_quick.py $R/gpurun_out/pmcs_b ${1:-kernel}
