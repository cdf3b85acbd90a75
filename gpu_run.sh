timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "conv" 2>&1 | tail -3
python tools/microbench.py conv 2>&1 | grep -v amdgpu | cut -c1-135
