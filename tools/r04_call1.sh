#!/bin/bash
# round-4 call 1: the new parity / de-risk tests
O=gpurun_out/r04a; mkdir -p $O
python -m pytest tests/test_gpu_headline.py -q -s -k "config2" > $O/headline.log 2>&1; tail -15 $O/headline.log
python -m pytest tests/test_gpu_kernels.py -q -k "plane_sums or conv1x1_nn_and_wgrad" > $O/kernels.log 2>&1; tail -5 $O/kernels.log
python -m pytest tests/test_gpu_distributed.py -q -k "multistep4 or rccl" > $O/dist.log 2>&1; tail -25 $O/dist.log
python -m pytest tests/test_crps.py -q > $O/crps.log 2>&1; tail -5 $O/crps.log
python -m pytest tests/test_bench_contract.py -q -k "eight" > $O/bench8.log 2>&1; tail -25 $O/bench8.log
