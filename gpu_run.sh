export MAKANI_AMD_BENCH_BACKEND=gloo
for par in dp h2w1 h1w2 h2w2; do
  n=2; [ $par = h2w2 ] && n=4
  echo "== $par n=$n"
  timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $n --config sfno_debug --parallelism $par --steps 3 --warmup 1 --no-cpu-baseline --no-sht-metric 2>&1 | grep -E "^\{|Error|error|Traceback" | cut -c1-330
done
