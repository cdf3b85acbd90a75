timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=line --timeout=300 -k "sgemm" 2>&1 | tail -3
python tools/gemmprobe.py 2>&1 | grep -v amdgpu
