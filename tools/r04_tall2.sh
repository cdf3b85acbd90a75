#!/bin/bash
O=gpurun_out/r04t; mkdir -p $O
for v in "" _nostld "" _nostld; do
  echo "== lib$v (tall)" | tee -a $O/ab_stld.txt
  MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd$v.so timeout 300 python tools/microbench.py dhconv 2>&1 | grep "^gen2 dhconv\|rel-L2" | tee -a $O/ab_stld.txt
done
for t in 1 0 1 0; do
  echo "== bench MAKANI_AMD_X2_TALL=$t" | tee -a $O/step_ab_tall.txt
  MAKANI_AMD_X2_TALL=$t timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" | tee -a $O/step_ab_tall.txt
done
