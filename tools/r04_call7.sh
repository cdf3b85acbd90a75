#!/bin/bash
# round-4 call 7: the two-group weight-stationary GEMM (MAKANI_AMD_ASTAT2=1): correctness, then same-box A/B
O=gpurun_out/r04g; mkdir -p $O
MAKANI_AMD_ASTAT2=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv1x1_nn_and_wgrad or conv_gelu" > $O/kernels.log 2>&1; tail -15 $O/kernels.log
MAKANI_AMD_ASTAT2=1 timeout 300 python -m pytest tests/test_gpu_headline.py -q -k "conv1x1_nn_fullres" > $O/headline.log 2>&1; tail -3 $O/headline.log
for v in 0 1; do echo "== MAKANI_AMD_ASTAT2=$v"; MAKANI_AMD_ASTAT2=$v timeout 300 python tools/microbench.py conv 2>&1 | grep -E "K=384"; done > $O/ab_astat2.txt 2>&1; cat $O/ab_astat2.txt
