# round-3 call 39: FFT kernels with non-temporal input loads, cold buffers, same box
mkdir -p gpurun_out/r03p
timeout 300 python tools/ab.py run cur fnt -- python tools/microbench.py fft cold 2>&1 | grep "fft" | tee gpurun_out/r03p/ab_fft_nt.txt
