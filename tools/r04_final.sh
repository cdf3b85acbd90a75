#!/bin/bash
# round-4 final: the whole GPU suite, smoke, the default bench line with its rocprofv3 / PMC passes, and the FourCastNet3 line
O=gpurun_out/r04z; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1; tail -4 $O/gpu_suite.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
bash tools/profile_round.sh r04z/prof > /dev/null 2>&1
python -c "import json; d=json.load(open('$O/prof/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['fwd_sht'])"
timeout 600 python bench.py --config fcn3_sc2_edim45_layers10 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_fcn3.json 2> $O/bench_fcn3.err
python -c "import json; d=json.load(open('$O/bench_fcn3.json')); print('fcn3', d['value'], d['ms_per_step'])"
