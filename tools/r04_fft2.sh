#!/bin/bash
O=gpurun_out/r04f; mkdir -p $O
timeout 1200 python -m pytest tests -q -x -m gpu -k "fft or seg or sht or transform" > $O/tests_fft.log 2>&1; tail -3 $O/tests_fft.log
timeout 600 python -m pytest tests/test_gpu_headline.py -q -x -k "fwd_bwd" > $O/tests_headline.log 2>&1; tail -2 $O/tests_headline.log
