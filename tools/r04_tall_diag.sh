#!/bin/bash
O=gpurun_out/r04t; mkdir -p $O
for v in "" _t1 _t2 _t4 _t5; do
  for t in 1 0; do
    echo "== lib$v TALL=$t" | tee -a $O/diag_tall.txt
    MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd$v.so MAKANI_AMD_X2_TALL=$t timeout 300 python tools/microbench.py dhconv 2>&1 | grep "^gen2 dhconv" | tee -a $O/diag_tall.txt
  done
done
for t in 1 0; do
  echo "== stamps TALL=$t" | tee -a $O/diag_tall.txt
  MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd_t32.so MAKANI_AMD_X2_TALL=$t timeout 300 python tools/x2_diag.py 2>&1 | grep -v amdgpu.ids | tee -a $O/diag_tall.txt
done
