#!/bin/bash
# round-4 call 4: fused bias gradient + counted waits of the epilogue-operand GEMMs: correctness, then same-box A/B
O=gpurun_out/r04d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv1x1 or conv_gelu" > $O/kernels.log 2>&1; tail -3 $O/kernels.log
timeout 900 python -m pytest tests/test_gpu_headline.py -q -k "conv1x1 or block_240" > $O/headline.log 2>&1; tail -3 $O/headline.log
timeout 900 python -m pytest tests/test_gpu_model.py -q -x > $O/model.log 2>&1; tail -3 $O/model.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sht-metric"
step() { "$@" 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['hip_kernels']; print(round(d['ms_per_step'],3), 'ms/step; nn', round(sum(v['ms_per_step'] for n,v in k.items() if n.startswith('conv1x1_nn')),3), 'wgrad', round(sum(v['ms_per_step'] for n,v in k.items() if n.startswith('conv1x1_wgrad')),3), 'plane_sums', round(sum(v['ms_per_step'] for n,v in k.items() if 'plane_sum' in n),3), 'loss', d['final_loss'])"; }
{
echo "== base (counted waits, fused bias gradient)"; step $B
echo "== round-3 drain at every tile start (libmakani_amd_drain.so)"; MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd_drain.so step $B
echo "== base, bias gradient by separate plane sums (MAKANI_AMD_WGRAD_BIAS=0)"; MAKANI_AMD_WGRAD_BIAS=0 step $B
echo "== base again"; step $B
echo "== drain again"; MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd_drain.so step $B
} > $O/step_ab.txt 2>&1; cat $O/step_ab.txt
python tools/ab.py run drain base -- python tools/microbench.py conv 2>&1 | grep -E "K=384 721|K=384 240" > $O/ab_conv_waits.txt; cat $O/ab_conv_waits.txt
