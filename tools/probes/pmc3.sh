cd /tmp && export TMPDIR=/tmp && export VEGANS_HIP_GRAPH=0
R=$GRAFT_REPO_ROOT
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_p1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-bf16-extra > $R/gpurun_out/pmc_p1.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_p2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-bf16-extra > $R/gpurun_out/pmc_p2.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_p3 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-bf16-extra > $R/gpurun_out/pmc_p3.log 2>&1 || exit 1
echo passes done
