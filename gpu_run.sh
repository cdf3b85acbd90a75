mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -8
timeout 120 python tools/microbench.py fft legendre 2>&1 | grep -v amdgpu | cut -c1-135
timeout 300 python bench.py --steps 10 --warmup 3 2>&1 | tail -2 | tee gpurun_out/bench_ft.json
