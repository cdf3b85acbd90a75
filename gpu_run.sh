set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final5
timeout 300 python -m pytest tests -m gpu -q --tb=short -k "checkpointing" 2>&1 | tail -3
timeout 200 python tools/microbench.py spectral 2>&1 | grep -v amdgpu > $R/gpurun_out/final5/microbench_spectral.txt; cat $R/gpurun_out/final5/microbench_spectral.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final5 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sht-metric > $R/gpurun_out/final5/bench_traced.log 2>&1
cd $R
find gpurun_out/final5 -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} gpurun_out/final5/kernel_stats.md > /dev/null
tail -1 gpurun_out/final5/bench_traced.log | cut -c1-300
timeout 400 python bench.py --steps 10 --warmup 3 > gpurun_out/final5/bench.json 2> gpurun_out/final5/bench.err; cut -c1-400 gpurun_out/final5/bench.json
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sht-metric --multistep-count 4 2>/dev/null | tail -1 > gpurun_out/final5/bench_multistep4.json
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sht-metric --multistep-count 4 --multistep-checkpoint 2>/dev/null | tail -1 > gpurun_out/final5/bench_multistep4_ckpt.json
python - <<'PY'
import json
for f in ("bench_multistep4","bench_multistep4_ckpt"):
    d=json.load(open(f"gpurun_out/final5/{f}.json")); print(f, d["ms_per_step"], d["peak_hbm_GB"], d["final_loss"])
PY
find gpurun_out/final5 -name "*.db" -delete
