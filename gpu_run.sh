mkdir -p gpurun_out/r05x
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_headline.py tests/test_gpu_model.py tests/test_gpu_fcn3_distributed.py -x -q -k "wgrad or conv1x1 or bias or mlp or block_240 or fcn3" > gpurun_out/r05x/pytest.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r05x/pytest.log
timeout 300 python tools/shadow_rank.py --h 4 --w 2 --steps 4 --json gpurun_out/r05x/shadow_h4w2.json > gpurun_out/r05x/shadow_h4w2.log 2>&1; echo "shadow rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05x/shadow_h4w2.json'))
print(d['hip_kernel_ms_per_step'], d['graph_ms_per_step_phantom'], d['families']['conv1x1_wgrad'])
PY
