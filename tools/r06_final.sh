#!/bin/bash
# round 6: how the records under profiles/r06_* were produced (each block was one gpurun call; see also docs/LAB_NOTEBOOK.md round 6)
# 1. the whole GPU suite with durations, then smoke                      -> profiles/r06_gpu_suite.txt
# 2. the bench line (live counter passes, CPU baseline, both settings of the fp32-matmul switch), kernel trace, SQ counters,
#    FourCastNet3 line with its CPU baseline                              -> profiles/r06_bench.json, r06_bench_kernel_stats.md, r06_pmc_*, r06_bench_fcn3.json
# 3. one rank of every split alone on the GPU, phantom collectives       -> profiles/r06_shadow_*.json, r06_shard_shapes.md
# 4. the interference hunt (tools/two_stream_hunt.py, two_stream_micro.py, pk_hazard_probe.py) -> profiles/r06_interference_root_cause.md
# 5. full-size distributed tests without compute-unit masks              -> profiles/r06_dist_fullsize_no_cu_mask.txt, r06_fcn3_fullsize_h2w2.txt
set -u
O=gpurun_out/${1:-r06_final}; mkdir -p $O
export TMPDIR=/tmp
if [ "${2:-all}" != "noprof" ]; then
  bash tools/profile_round.sh ${1:-r06_final}/prof fcn3 > /dev/null 2>&1
  python -c "import json; d=json.loads(open('$O/prof/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['exact_fp32_spectral'], d['fwd_sht'])"
  for s in "1 1" "2 1" "4 1" "4 2"; do set -- $s; timeout 600 python tools/shadow_rank.py --h $1 --w $2 --steps 4 --json $O/shadow_h$1w$2.json > $O/shadow_h$1w$2.log 2>&1; done
  python tools/shard_table.py $O/shadow_h1w1.json $O/shadow_h2w1.json $O/shadow_h4w1.json $O/shadow_h4w2.json > $O/shard_shapes.md 2>/dev/null; head -30 $O/shard_shapes.md
fi
if [ "${2:-all}" != "nosuite" ]; then
  timeout 1800 python -m pytest tests -q -m gpu --durations=40 > $O/gpu_suite.log 2>&1; tail -4 $O/gpu_suite.log
  timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
fi
