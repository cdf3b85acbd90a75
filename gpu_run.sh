#!/bin/bash
O=gpurun_out/r07y; mkdir -p $O
for r in 1 2; do
for v in d 1; do
  if [ $v = d ]; then unset MAKANI_AMD_ASTAT2; else export MAKANI_AMD_ASTAT2=$v; fi
  python tools/microbench.py conv 2>&1 | grep -E "conv M=(768|384) K=384" | sed "s/^/astat2=$v | /" | cut -c1-175 >> $O/micro.txt
done; done
sort -k4,4 -k6,6 -s $O/micro.txt
