cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in fwd dgrad wgrad; do
  timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmcl_$m -- python3 $R/tools/conv_micro.py $m 256 16 16 128 256 4 2 1 10 > $R/gpurun_out/pmcl_$m.log 2>&1 || exit 1
  python3 $R/tools/pmc_quick.py $R/gpurun_out/pmcl_$m igemm
done
