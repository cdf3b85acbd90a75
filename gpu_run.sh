timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -4
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sht-metric 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['hip_kernel_ms_per_step'], d['final_loss'])"
