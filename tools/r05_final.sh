#!/bin/bash
# round 5: how the records under profiles/r05_* were produced (each block was one gpurun call)
# 1. the whole GPU suite with durations, then smoke                      -> profiles/r05_gpu_suite.txt
#      python -m pytest tests/ -x -q -m gpu --durations=40 ; python __graft_entry__.py smoke
# 2. the bench line (live counter passes, CPU baseline with bf16 yardsticks), kernel trace, SQ counters, FourCastNet3 line
#      bash tools/profile_round.sh r05z fcn3
#      python tools/kernel_stats_md.py gpurun_out/r05z/kernel_stats.csv 18 "<title>" sfno > profiles/r05_bench_kernel_stats.md
#                                                                         -> profiles/r05_bench.json, r05_pmc_*, r05_bench_fcn3.json
# 3. one rank of every split alone on the GPU, phantom collectives       -> profiles/r05_shadow_*.json, r05_shard_shapes.md
#      for s in "1 1" "2 1" "4 1" "4 2"; do set -- $s; python tools/shadow_rank.py --h $1 --w $2 --steps 4 --json gpurun_out/x/shadow_h$1w$2.json; done
#      python tools/shard_table.py gpurun_out/x/shadow_h1w1.json ... > profiles/r05_shard_shapes.md
# 4. full-size distributed tests (their log lines)                        -> profiles/r05_dist_fullsize*.txt
#      MAKANI_AMD_DIST_LOG=... python -m pytest tests/test_gpu_dist_fullsize.py -q -s
# 5. the 8-rank line on ONE GPU over gloo at full size (functional)       -> profiles/r05_bench_8ranks_one_gpu_gloo.json
#      MAKANI_AMD_BENCH_BACKEND=gloo python bench.py --gpus 8 --steps 2 --warmup 1 --no-secondary
# 6. the hunts: tools/race_hunt.py (docs/LAB_NOTEBOOK.md 5.1), tools/probes/rccl_graph_probe.py (5.2), tools/glue_trace.py
set -u
O=gpurun_out/r05_final; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --durations=40 > $O/gpu_suite.log 2>&1; tail -4 $O/gpu_suite.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
bash tools/profile_round.sh r05_final/prof fcn3 > /dev/null 2>&1
python -c "import json; d=json.load(open('$O/prof/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['fwd_sht'])"
