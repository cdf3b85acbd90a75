#!/bin/bash
O=gpurun_out/r07p; mkdir -p $O
for v in 2 -1 0 2 -1 0; do
  if [ $v = -1 ]; then unset MAKANI_AMD_CONV_NT; else export MAKANI_AMD_CONV_NT=$v; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-exact --no-sht-metric 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nt=$v', round(d['ms_per_step'],3), d['final_loss'])" >> $O/bench.txt 2>&1; done
cat $O/bench.txt
