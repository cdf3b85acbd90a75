#!/bin/bash
O=gpurun_out/r07i; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_optim.py -x -q -m gpu > $O/tests.log 2>&1
tail -n 3 $O/tests.log
for v in 0 1 0 1; do
  MAKANI_AMD_GRAD_SSQ=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-exact --no-sht-metric 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ssq=$v', round(d['ms_per_step'],3), d['final_loss'])" >> $O/bench.txt 2>&1; done
cat $O/bench.txt
