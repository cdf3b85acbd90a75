#!/bin/bash
O=gpurun_out/r07j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gelu or conv1x1 or instance_norm" > $O/tests.log 2>&1
tail -n 3 $O/tests.log
python tools/ab.py run gg0 gg1 -- python tools/microbench.py cold conv > $O/micro.txt 2>&1
grep -E "instnorm bwd|conv M" $O/micro.txt | cut -c1-230
for t in gg0 gg1 gg0 gg1; do
  MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd_$t.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-exact --no-sht-metric 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', round(d['ms_per_step'],3), d['final_loss'])" >> $O/bench.txt 2>&1; done
cat $O/bench.txt
