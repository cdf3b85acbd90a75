timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "fft" 2>&1 | tail -3
for v in 0 1; do echo "== variant $v"; MAKANI_AMD_FFT_VARIANT=$v timeout 120 python tools/microbench.py fft 2>&1 | grep -v amdgpu | cut -c1-135; done
