mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q --tb=short --timeout=300 -x > gpurun_out/t_all.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/t_all.log
timeout 300 python tools/microbench.py > gpurun_out/micro.log 2>&1; cat gpurun_out/micro.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sht-metric > gpurun_out/bench_full.log 2>&1
echo "bench rc=$?"; tail -1 gpurun_out/bench_full.log | cut -c1-400
