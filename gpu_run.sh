for v in 64 128 64 128; do MAKANI_AMD_WGRAD_BK=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sht-metric 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bk $v', d['ms_per_step'], d['hip_kernel_ms_per_step'], sum(v['ms_per_step'] for k,v in d['hip_kernels'].items() if 'wgrad_m' in k))"; done
