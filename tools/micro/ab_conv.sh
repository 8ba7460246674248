# Operator tests, the benchmark line and a single-stream kernel trace of the benchmark (per-kernel durations of the
# input-gradient kernels printed): what a change to the contraction kernels is compared on.  usage: bash tools/micro/ab_conv.sh
set -e
export TMPDIR=/tmp
R=$(pwd)
timeout -k 10 400 python -u -m pytest tests/test_ops_gpu.py -x -q --timeout 120 --timeout-method thread > gpurun_out/ab_t1.log 2>&1 || { tail -30 gpurun_out/ab_t1.log; exit 1; }
tail -2 gpurun_out/ab_t1.log
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/ab_bench.json 2> gpurun_out/ab_bench.err
python -c "
import json;d=json.load(open('gpurun_out/ab_bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d.get('conv_kernels',{}).get('aggregate_tflops'),d.get('bf16_mode',{}).get('ms_per_step'))"
cd /tmp
VEGANS_HIP_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_stats1 -o abs -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-extra > $R/gpurun_out/ab_stats1.log 2>&1
cd $R
python tools/rocprof_summary.py gpurun_out/ab_stats1 gpurun_out/ab_stats1_summary 'single stream bench'
grep 'igemm_kernel<1\|Total' gpurun_out/ab_stats1_summary.md
