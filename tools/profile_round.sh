#!/bin/bash
# One gpurun call: the default bench line — it collects its own roofline.traffic with two rocprofv3 counter passes (FETCH_SIZE,
# WRITE_SIZE; their per-kernel tables are kept under <out>/pmc_live) — , the rocprofv3 kernel-trace summary of the same command
# and an SQ-activity counter pass.  Usage: bash tools/profile_round.sh <out-dir-under-gpurun_out> [fcn3]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-prof}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MAKANI_AMD_PMC_KEEP=$O/pmc_live python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sht-metric --no-pmc --no-exact > $O/kt.log 2>&1
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/kt
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sht-metric --graph off --no-pmc --no-exact > $O/pmc_sq.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_sq.md $(find $O/pmc_sq -name "*counter_collection.csv") > /dev/null 2>&1
rm -rf $O/pmc_sq
if [ "${2:-}" = "fcn3" ]; then
  # FourCastNet3 (BASELINE configs[3]): the line with its own counter passes
  MAKANI_AMD_PMC_KEEP=$O/pmc_live_fcn3 timeout 1500 python $R/bench.py --config fcn3_sc2_edim45_layers10 --steps 5 --warmup 2 > $O/bench_fcn3.json 2> $O/bench_fcn3.err
fi
ls -la $O $O/pmc_live 2>/dev/null
