#!/bin/bash
# round-4 call 2: two-half FFT kernels (correctness + same-box A/B of the skew), the tests fixed after call 1
O=gpurun_out/r04b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "rfft or irfft or fft_adjoint or segmented" > $O/fft_tests.log 2>&1; tail -4 $O/fft_tests.log
timeout 900 python tools/ab.py run hv1 s1 s2 s3 -- python tools/microbench.py fft > $O/ab_fft_halves.txt 2>&1; cat $O/ab_fft_halves.txt
timeout 300 python -m pytest tests/test_crps.py -q > $O/crps.log 2>&1; tail -3 $O/crps.log
timeout 600 python -m pytest tests/test_gpu_distributed.py -q -k "multistep4" > $O/dist.log 2>&1; tail -5 $O/dist.log
timeout 900 python -m pytest tests/test_gpu_headline.py -q -s -k "config2" > $O/headline.log 2>&1; grep -n "config 2\|^    [a-z]\|passed\|failed" $O/headline.log | head -30
