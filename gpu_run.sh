#!/bin/bash
O=gpurun_out/r07e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "forward_gelu" > $O/tests.log 2>&1
grep -n "AssertionError" $O/tests.log | head
