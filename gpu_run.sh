timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "fft" 2>&1 | tail -2
MK_FFT480=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "fft" 2>&1 | tail -2
for v in 0 1; do echo "== fft480 variant $v"; MK_FFT480=$v timeout 120 python tools/microbench.py fft 2>&1 | grep -v amdgpu | grep 240x480 | cut -c1-110; done
