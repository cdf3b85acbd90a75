# round-3 call 8: SEG irfft after the store fix, the oracle's full CPU train step on this host, BASELINE configs[4] (multistep 4),
# FourCastNet3 profile set (kernel stats + HBM traffic of the contraction kernels)
mkdir -p gpurun_out/r03g
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03g
python tools/segfft_bench.py > $O/segfft.txt 2>&1; cat $O/segfft.txt
( time MAKANI_AMD_CPU_BASELINE=step timeout 400 python bench.py --cpu-worker sfno_sc3_layers8_edim384 --cpu-mode step ) > $O/cpu_step.log 2>&1 &
CPUPID=$!
timeout 600 python bench.py --multistep-count 4 --steps 5 --warmup 2 --no-cpu-baseline --no-sht-metric > $O/bench_multistep4.json 2> $O/bench_multistep4.err
timeout 600 python bench.py --multistep-count 4 --multistep-checkpoint --steps 5 --warmup 2 --no-cpu-baseline --no-sht-metric > $O/bench_multistep4_ckpt.json 2> $O/bench_multistep4_ckpt.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --config fcn3_sc2_edim45_layers10 --steps 5 --warmup 2 > $O/kt_fcn3.log 2>&1
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_fcn3.csv \;
rm -rf $O/kt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --config fcn3_sc2_edim45_layers10 --steps 1 --warmup 1 --graph off > $O/pmc_fcn3_$c.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc_fcn3_$c.md $(find $O/pmc_$c -name "*counter_collection.csv") > /dev/null 2>&1
  rm -rf $O/pmc_$c
done
wait $CPUPID
cd $R
tail -4 $O/cpu_step.log
python - <<'PY'
import json
for f in ("bench_multistep4", "bench_multistep4_ckpt"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r03g/{f}.json") if l.startswith("{")][-1])
        print(f, {k: d[k] for k in ("value", "ms_per_step", "peak_hbm_GB")}, d["config"]["launch"])
    except Exception as e:
        print(f, "failed", e)
PY
ls -la $O | head -30
