MAKANI_AMD_CONV=hip timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q --tb=short -x -k "conv or sfno" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "conv" 2>&1 | tail -2
