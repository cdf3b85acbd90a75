timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q --tb=short -x -k "adamw or sfno or train" 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sht-metric 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['hip_kernel_ms_per_step'], d['roofline']['kernel'], d['roofline']['achieved'], len(d['hip_kernels']))"; done
