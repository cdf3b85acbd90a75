for w in 1 2 3; do timeout 300 python bench.py --steps 5 --warmup $w --no-cpu-baseline --no-sht-metric 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('warmup $w', d['ms_per_step'])"; done
