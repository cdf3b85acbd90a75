#!/bin/bash
O=gpurun_out/r04f; mkdir -p $O
MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd_oldlds.so timeout 300 python tools/fft_plan_check.py /tmp/fft_old.pt 2>&1 | tail -1 | tee -a $O/fft_plan.txt
timeout 300 python tools/fft_plan_check.py /tmp/fft_new.pt 2>&1 | tail -1 | tee -a $O/fft_plan.txt
python tools/fft_plan_check.py /tmp/fft_old.pt /tmp/fft_new.pt 2>&1 | tail -1 | tee -a $O/fft_plan.txt
for v in _oldlds "" _oldlds ""; do
  echo "== lib$v" | tee -a $O/fft_plan.txt
  MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd$v.so timeout 300 python tools/microbench.py fft 2>&1 | grep "fft " | tee -a $O/fft_plan.txt
done
