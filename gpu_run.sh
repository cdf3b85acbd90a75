R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc8
timeout 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sht-metric > $R/gpurun_out/pmc8/bench.json 2> $R/gpurun_out/pmc8/bench.err; tail -c 300 $R/gpurun_out/pmc8/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/pmc8/bench.json")); print(d["ms_per_step"], d["roofline"]); print([ (o["kernel"], o["bound"], o["frac"]) for o in d["roofline_runners_up"] if o])
PY
cd /tmp && export TMPDIR=/tmp
timeout 60 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc8/fetch -- python $R/tools/wgrad_step_mix.py > $R/gpurun_out/pmc8/fetch.log 2>&1
timeout 60 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc8/write -- python $R/tools/wgrad_step_mix.py > $R/gpurun_out/pmc8/write.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc8/summary.md $(find gpurun_out/pmc8 -name "*counter_collection.csv") | grep -i "wgrad\|reduce_splits\|kernel"
find gpurun_out/pmc8 -name "*.csv" ! -name "*counter_collection.csv" -delete
