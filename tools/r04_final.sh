#!/bin/bash
# round-4 final: the whole GPU suite, smoke, and the default bench line on the final library
O=gpurun_out/r04z; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1; tail -4 $O/gpu_suite.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['fwd_sht'])"
