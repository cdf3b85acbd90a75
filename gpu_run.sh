# nine piecewise linear basis functions (kernel_shape (5, 4)): the fused K = 9 kernels on a basis other than Morlet
mkdir -p gpurun_out/r05g
SECONDS=0
timeout 60 python -m pytest tests/test_gpu_disco.py -q -m gpu -k "other_bases and kshape6" > gpurun_out/r05g/pl9.log 2>&1; echo "rc $? in $SECONDS s"
grep -v "amdgpu.ids" gpurun_out/r05g/pl9.log | tail -12 | cut -c1-400
