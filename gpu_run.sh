mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -4
timeout 300 python bench.py --steps 10 --warmup 3 2>&1 | tail -1 > gpurun_out/bench_v6.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_v6.json').read())
print(d['ms_per_step'], d['value'], d['hip_kernel_ms_per_step'], d['fwd_sht'])
for k,v in d['hip_kernels'].items():
    if 'fft' in k or 'legendre' in k or 'dhconv' in k: print(k, v['ms_avg'], v['ms_per_step'])
PY
