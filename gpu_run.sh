# round-3 call 29: complex split kernel with one loop per wave group (exact wait counts, two k-steps of prefetch distance)
mkdir -p gpurun_out/r03p
MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd_tl.so timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_headline.py -q -x -m gpu -k "cgemm or dhconv or contract or spectral" 2>&1 | tail -3
timeout 300 python tools/ab.py run cur tl -- python tools/microbench.py dhconv 2>&1 | grep -v gen1 | tee gpurun_out/r03p/ab_twoloops.txt
