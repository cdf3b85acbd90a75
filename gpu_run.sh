timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sht-metric 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('N=1', d['ms_per_step'], d['final_loss'], d['roofline']['kernel'], d['roofline']['frac'])"
MAKANI_AMD_BENCH_BACKEND=gloo timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --config sfno_debug 2>&1 | tail -1 | cut -c1-330
