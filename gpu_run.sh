# round-3 call 25: two-pass FFTs (480 = 2 x (16 x 15), 1440 = 2 x (30 x 24)): tests on the variant, then same-box A/B
mkdir -p gpurun_out/r03m
MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd_p2.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "fft or sht" 2>&1 | tail -3
timeout 600 python tools/ab.py run cur p2 -- python tools/microbench.py fft 2>&1 | tee gpurun_out/r03m/ab_fft_2pass.txt
