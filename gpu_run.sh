mkdir -p gpurun_out/r05v
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_headline.py tests/test_gpu_model.py -x -q -k "wgrad or conv1x1 or bias or mlp or block_240" > gpurun_out/r05v/pytest.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r05v/pytest.log
cd /tmp; timeout 300 python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sht-metric --no-pmc > $GRAFT_REPO_ROOT/gpurun_out/r05v/bench.json 2>/dev/null
python - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05v/bench.json"))
print(d['value'], d['ms_per_step'], d['hip_kernel_ms_per_step'])
k=d['hip_kernels']
print(sum(v['ms_per_step'] for n,v in k.items() if 'wgrad' in n))
PY
