timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "instance_norm" 2>&1 | tail -2
for v in 0 1; do echo "== plane path $v"; MAKANI_AMD_NORM_PLANE=$v timeout 120 python tools/microbench.py pointwise 2>&1 | grep -v amdgpu | grep "instnorm" | cut -c1-100; done
