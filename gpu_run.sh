mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q --tb=short --timeout=300 > gpurun_out/t_all.log 2>&1
echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/t_all.log | tail -8
MAKANI_AMD_GEMM=fp32 timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short --timeout=300 2>&1 | tail -2
MAKANI_AMD_GEMM=x3 timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=line --timeout=300 2>&1 | tail -6
timeout 250 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_full.log 2>gpurun_out/bench_full.err
echo "bench rc=$?"; tail -1 gpurun_out/bench_full.log | cut -c1-200
timeout 100 python __graft_entry__.py smoke 2>&1 | tail -1
