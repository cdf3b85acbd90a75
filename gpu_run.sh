# round-3 call 17: packed-fp32 FFT butterflies: FFT / SHT tests, then same-box A/B against the previous library
mkdir -p gpurun_out/r03l
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "fft or sht" 2>&1 | tail -5
timeout 600 python tools/ab.py run base pk -- python tools/microbench.py fft 2>&1 | tee gpurun_out/r03l/ab_fft_packed.txt
