#!/bin/bash
O=gpurun_out/r07t; mkdir -p $O
for v in 0 128 0 128 0 128; do
  MAKANI_AMD_CONV_NT_MIN=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-exact --no-sht-metric 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ntmin=$v', round(d['ms_per_step'],3), d['final_loss'])" >> $O/bench.txt 2>&1; done
cat $O/bench.txt
