#!/bin/bash
O=gpurun_out/r04m; mkdir -p $O
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sht-metric"
step() { "$@" 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['hip_kernels']; print(round(d['ms_per_step'],3), 'ms/step; nn', round(sum(v['ms_per_step'] for n,v in k.items() if n.startswith('conv1x1_nn')),3), 'loss', d['final_loss'])"; }
{ for i in 1 2; do echo "== default (two-group kernel except bias+GELU+pre)"; step $B; echo "== MAKANI_AMD_ASTAT2=1 (two-group kernel everywhere)"; MAKANI_AMD_ASTAT2=1 step $B; echo "== MAKANI_AMD_ASTAT2=0"; MAKANI_AMD_ASTAT2=0 step $B; done; } > $O/step_ab.txt 2>&1; cat $O/step_ab.txt
