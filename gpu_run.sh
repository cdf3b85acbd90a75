#!/bin/bash
O=gpurun_out/r07o; mkdir -p $O
python tools/ab.py run w0 w1 w3 -- python tools/microbench.py wgrad > $O/micro.txt 2>&1
grep -E "wgrad" $O/micro.txt | cut -c1-170 | head -45
for t in w0 w1 w3 w0 w1 w3; do
  MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd_$t.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-exact --no-sht-metric 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', round(d['ms_per_step'],3), d['final_loss'])" >> $O/bench.txt 2>&1; done
cat $O/bench.txt
