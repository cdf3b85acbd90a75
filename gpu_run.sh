timeout 600 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-sht-metric 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('40 steps', d['ms_per_step'], d['final_loss'])"
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sht-metric --fp32 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fp32', d['ms_per_step'], d['final_loss'], d['dtype'])"
python - <<'PY'
import torch
print('max mem GB', torch.cuda.max_memory_allocated()/1e9)
PY
