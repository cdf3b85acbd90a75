#!/bin/bash
O=gpurun_out/r04t; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "dhconv or gemm or split" > $O/tests.log 2>&1; tail -3 $O/tests.log
for t in 0 1; do
  echo "== MAKANI_AMD_X2_TALL=$t"; MAKANI_AMD_X2_TALL=$t timeout 300 python tools/microbench.py dhconv 2>&1 | grep -v "^gen1" | tee -a $O/ab_tall.txt
done
timeout 600 python -m pytest tests/test_gpu_headline.py -q -x -k "fwd_bwd or config2" > $O/headline.log 2>&1; tail -3 $O/headline.log
