#!/bin/bash
O=gpurun_out/r04i; mkdir -p $O
MAKANI_AMD_ASTAT2=1 MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd_adiag.so timeout 300 python tools/astat_diag.py > $O/astat2_diag.txt 2>&1; cat $O/astat2_diag.txt
