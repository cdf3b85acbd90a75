# round-3 call 16: kernel trace of an fp32 (no autocast) run: no library GEMM kernel may appear
mkdir -p gpurun_out/r03k
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03k/kt -- python $R/bench.py --fp32 --steps 2 --warmup 1 --graph off --no-cpu-baseline --no-sht-metric > $R/gpurun_out/r03k/kt.log 2>&1
find $R/gpurun_out/r03k/kt -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r03k/kernel_stats_fp32.csv \;
rm -rf $R/gpurun_out/r03k/kt
grep -c "Cijk" $R/gpurun_out/r03k/kernel_stats_fp32.csv
grep '^{' $R/gpurun_out/r03k/kt.log | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['dtype'])"
head -8 $R/gpurun_out/r03k/kernel_stats_fp32.csv | cut -c1-150
