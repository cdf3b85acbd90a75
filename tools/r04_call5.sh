#!/bin/bash
# round-4 call 5: bias gradient shared over the waves of a row (v2): correctness + same-box A/B, FourCastNet3 line, then the whole GPU suite
O=gpurun_out/r04e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv1x1 or conv_gelu" > $O/kernels.log 2>&1; tail -3 $O/kernels.log
timeout 900 python -m pytest tests/test_gpu_headline.py -q -k "conv1x1_wgrad" > $O/headline.log 2>&1; tail -3 $O/headline.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sht-metric"
step() { "$@" 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['hip_kernels']; print(round(d['ms_per_step'],3), 'ms/step; nn', round(sum(v['ms_per_step'] for n,v in k.items() if n.startswith('conv1x1_nn')),3), 'wgrad', round(sum(v['ms_per_step'] for n,v in k.items() if n.startswith('conv1x1_wgrad')),3), 'loss', d['final_loss'], 'peak GB', d['peak_hbm_GB'])"; }
{
echo "== fused bias gradient (shared over the waves of a row)"; step $B
echo "== separate plane sums (MAKANI_AMD_WGRAD_BIAS=0)"; MAKANI_AMD_WGRAD_BIAS=0 step $B
echo "== fused again"; step $B
echo "== separate again"; MAKANI_AMD_WGRAD_BIAS=0 step $B
} > $O/step_ab_bias.txt 2>&1; cat $O/step_ab_bias.txt
{
echo "== fcn3 fused"; step python bench.py --config fcn3_sc2_edim45_layers10 --steps 5 --warmup 2 --no-cpu-baseline
echo "== fcn3 separate"; MAKANI_AMD_WGRAD_BIAS=0 step python bench.py --config fcn3_sc2_edim45_layers10 --steps 5 --warmup 2 --no-cpu-baseline
} > $O/step_ab_bias_fcn3.txt 2>&1; cat $O/step_ab_bias_fcn3.txt
timeout 1500 python -m pytest tests -q -m gpu -x > $O/gpu_suite.log 2>&1; tail -5 $O/gpu_suite.log
