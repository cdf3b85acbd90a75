#!/bin/bash
# after the last kernel change of the round (FFT LDS layouts): smoke, the default bench line, the whole-network GPU tests
O=gpurun_out/r04z2; mkdir -p $O
timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['hip_kernels']['rfft_1440']['ms_avg'], d['hip_kernels']['irfft_1440']['ms_avg'], d['hip_kernels']['rfft_480']['ms_avg'], d['hip_kernels']['irfft_480']['ms_avg'])"
timeout 170 python -m pytest tests/test_gpu_model.py -q -x > $O/tests_model.log 2>&1; tail -2 $O/tests_model.log
