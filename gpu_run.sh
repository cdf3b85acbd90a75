mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q --tb=short --timeout=300 > gpurun_out/t_all.log 2>&1
echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/t_all.log | tail -15
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sht-metric > gpurun_out/bench_full.log 2>&1
echo "bench rc=$?"; tail -1 gpurun_out/bench_full.log | cut -c1-200
MAKANI_AMD_CONV=hip timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short --timeout=300 2>&1 | tail -3
