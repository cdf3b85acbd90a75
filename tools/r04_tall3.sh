#!/bin/bash
O=gpurun_out/r04t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "high_degrees or square_tile" > $O/tests3.log 2>&1; tail -3 $O/tests3.log
for t in 1 0; do MAKANI_AMD_X2_TALL=$t timeout 300 python tools/dhconv_accuracy.py 2>&1 | grep MAKANI | tee -a $O/accuracy.txt; done
