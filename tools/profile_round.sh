#!/bin/bash
# One gpurun call: the default bench line, its rocprofv3 kernel-trace summary, and the PMC passes (HBM fetch / write,
# SQ activity) of the same command.  Usage: bash tools/profile_round.sh <out-dir-under-gpurun_out>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-prof}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sht-metric > $O/kt.log 2>&1
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/kt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sht-metric --graph off > $O/pmc_$c.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc_$c.md $(find $O/pmc_$c -name "*counter_collection.csv") > /dev/null 2>&1
  rm -rf $O/pmc_$c
done
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sht-metric --graph off > $O/pmc_sq.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_sq.md $(find $O/pmc_sq -name "*counter_collection.csv") > /dev/null 2>&1
rm -rf $O/pmc_sq
ls -la $O
