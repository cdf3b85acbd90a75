#!/bin/bash
O=gpurun_out/r04h; mkdir -p $O
MAKANI_AMD_ASTAT2=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv1x1_nn_and_wgrad or conv_gelu" > $O/kernels.log 2>&1; tail -3 $O/kernels.log
{ echo "== astat (one group)"; MAKANI_AMD_ASTAT2=0 timeout 300 python tools/microbench.py conv 2>&1 | grep -E "K=384";
  echo "== astat2, 4 slots"; MAKANI_AMD_ASTAT2=1 timeout 300 python tools/microbench.py conv 2>&1 | grep -E "K=384";
  echo "== astat2, 6 slots without epilogue operand"; MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd_s6.so MAKANI_AMD_ASTAT2=1 timeout 300 python tools/microbench.py conv 2>&1 | grep -E "K=384"; } > $O/ab_astat2.txt 2>&1; cat $O/ab_astat2.txt
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sht-metric"
step() { "$@" 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['hip_kernels']; print(round(d['ms_per_step'],3), 'ms/step; nn', round(sum(v['ms_per_step'] for n,v in k.items() if n.startswith('conv1x1_nn')),3), 'loss', d['final_loss'])"; }
{ echo "== astat"; MAKANI_AMD_ASTAT2=0 step $B; echo "== astat2"; MAKANI_AMD_ASTAT2=1 step $B; echo "== astat"; MAKANI_AMD_ASTAT2=0 step $B; echo "== astat2"; MAKANI_AMD_ASTAT2=1 step $B; } > $O/step_ab_astat2.txt 2>&1; cat $O/step_ab_astat2.txt
