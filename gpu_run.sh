for v in 0 1 0 1; do MAKANI_AMD_DEFER_BIAS=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sht-metric 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('defer $v', d['ms_per_step'], d['hip_kernel_ms_per_step'], d['final_loss'])"; done
