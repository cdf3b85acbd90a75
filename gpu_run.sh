python tools/fftprobe.py 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short --timeout=300 -k "fft or sht" 2>&1 | tail -3
