#!/bin/bash
# round-4 call 3: 73-channel weight-stationary GEMM + narrow Legendre kernel: correctness, then same-box A/B through the env switches
O=gpurun_out/r04c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv1x1_nn_and_wgrad or sgemm or cgemm or rfft or irfft" > $O/kernels.log 2>&1; tail -4 $O/kernels.log
timeout 600 python -m pytest tests/test_gpu_headline.py -q -k "conv1x1_nn_fullres" > $O/conv_full.log 2>&1; tail -3 $O/conv_full.log
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_distributed.py -q -x > $O/model.log 2>&1; tail -4 $O/model.log
for v in 0 1; do echo "== MAKANI_AMD_X2_NARROW=$v"; MAKANI_AMD_X2_NARROW=$v timeout 300 python tools/microbench.py sht; done > $O/ab_narrow.txt 2>&1; cat $O/ab_narrow.txt
for v in 0 1; do echo "== MAKANI_AMD_ASTAT_SMALLK=$v"; MAKANI_AMD_ASTAT_SMALLK=$v timeout 600 python tools/microbench.py conv 2>&1 | grep -E "K= 73|M=384 K=384 721"; done > $O/ab_smallk.txt 2>&1; cat $O/ab_smallk.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>$O/bench.err | grep '^{' | tail -1 > $O/bench.json
python -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['final_loss'], d['fwd_sht'])"
