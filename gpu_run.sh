export MAKANI_AMD_BENCH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-sht-metric --config sfno_debug 2>&1 | grep -E '^\{|Error|error' | cut -c1-300
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 4 --steps 2 --warmup 1 --no-cpu-baseline --no-sht-metric --config sfno_debug --parallelism h2w2 2>&1 | grep -E '^\{|Error|error' | cut -c1-300
unset MAKANI_AMD_BENCH_BACKEND
timeout 600 python -m pytest tests/test_gpu_distributed.py -m gpu -q -x 2>&1 | tail -2
