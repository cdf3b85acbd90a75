#!/bin/bash
O=gpurun_out/r04k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv1x1 or conv_gelu" > $O/kernels.log 2>&1; tail -3 $O/kernels.log
timeout 900 python -m pytest tests/test_gpu_headline.py -q > $O/headline.log 2>&1; tail -3 $O/headline.log
timeout 600 python -m pytest tests/test_gpu_model.py -q -x > $O/model.log 2>&1; tail -3 $O/model.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sht-metric"
step() { "$@" 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['hip_kernels']; print(round(d['ms_per_step'],3), 'ms/step; nn', round(sum(v['ms_per_step'] for n,v in k.items() if n.startswith('conv1x1_nn')),3), 'loss', d['final_loss'])"; }
{ echo "== default (two-group kernel where faster)"; step $B; echo "== MAKANI_AMD_ASTAT2=0"; MAKANI_AMD_ASTAT2=0 step $B; echo "== default"; step $B; echo "== MAKANI_AMD_ASTAT2=0"; MAKANI_AMD_ASTAT2=0 step $B; } > $O/step_ab.txt 2>&1; cat $O/step_ab.txt
