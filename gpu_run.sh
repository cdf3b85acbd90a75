# round-3 call 26: 480-point FFT with two workgroups per CU actually requested (launch bounds = waves per SIMD)
mkdir -p gpurun_out/r03m
timeout 600 python tools/ab.py run cur lb -- python tools/microbench.py fft 2>&1 | tee gpurun_out/r03m/ab_fft_lb.txt
