#!/bin/bash
O=gpurun_out/r07s; mkdir -p $O
for v in 2 -1 1 2 -1 1 2 -1 1; do
  if [ $v = -1 ]; then unset MAKANI_AMD_CONV_NT; else export MAKANI_AMD_CONV_NT=$v; fi
  timeout 900 python bench.py --config fcn3_sc2_edim45_layers10 --steps 8 --warmup 2 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fcn3 nt=$v', round(d['ms_per_step'],2), d['value'])" >> $O/bench.txt 2>&1; done
cat $O/bench.txt
