# round-3 call 43: forward 480-point bf16 FFT with two workgroups per CU, same box
mkdir -p gpurun_out/r03s
MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd_occ.so timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "fft" 2>&1 | tail -2
timeout 300 python tools/ab.py run cur occ -- python tools/microbench.py fft cold 2>&1 | grep "rfft" | grep -v irfft | tee gpurun_out/r03s/ab_fft_occ.txt
