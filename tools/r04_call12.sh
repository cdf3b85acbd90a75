#!/bin/bash
O=gpurun_out/r04l; mkdir -p $O
{ echo "== default"; MAKANI_AMD_ASTAT2=1 timeout 300 python tools/microbench.py conv 2>&1 | grep -E "K=384";
  echo "== s_setprio 1 around the multiplication phases"; MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd_prio.so MAKANI_AMD_ASTAT2=1 timeout 300 python tools/microbench.py conv 2>&1 | grep -E "K=384"; } > $O/ab_astat2_prio.txt 2>&1; cat $O/ab_astat2_prio.txt
