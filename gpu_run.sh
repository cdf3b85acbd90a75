MAKANI_AMD_CONV=hip timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q --tb=short -x -k "conv or sfno" 2>&1 | tail -2
timeout 900 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -2
MAKANI_AMD_CONV=hip timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sht-metric 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('hip conv path', d['ms_per_step'], d['final_loss'])"
