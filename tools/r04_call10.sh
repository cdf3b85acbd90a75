#!/bin/bash
O=gpurun_out/r04j; mkdir -p $O
MAKANI_AMD_ASTAT2=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv1x1_nn_and_wgrad or conv_gelu" > $O/kernels.log 2>&1; tail -3 $O/kernels.log
MAKANI_AMD_ASTAT2=1 timeout 300 python -m pytest tests/test_gpu_headline.py -q -k "conv1x1_nn_fullres" > $O/headline.log 2>&1; tail -2 $O/headline.log
{ echo "== astat (one group)"; MAKANI_AMD_ASTAT2=0 timeout 300 python tools/microbench.py conv 2>&1 | grep -E "K=384";
  echo "== astat2"; MAKANI_AMD_ASTAT2=1 timeout 300 python tools/microbench.py conv 2>&1 | grep -E "K=384"; } > $O/ab_astat2.txt 2>&1; cat $O/ab_astat2.txt
MAKANI_AMD_ASTAT2=1 MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd_adiag.so timeout 300 python tools/astat_diag.py 2>&1 | head -9 > $O/astat2_diag.txt; cat $O/astat2_diag.txt
